// Does VALU work of one wave run in the shadow of another wave's MFMAs on the same SIMD (gfx950)?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run a chain of v_mfma_f32_32x32x2_f32, waves 4-7
// (second wave of each SIMD) run dependent-free v_fma_f32.  Times: MFMA waves alone, VALU waves alone, both.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap tools/microbench/mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k(float *out, int n_mfma, int n_valu, int mode, int interleave) {
  const int wave = threadIdx.x >> 6;
  if (interleave) {  // every wave: n_mfma groups of (2 MFMA + `interleave` FMAs)
    f32x16 a = {0}, b = {0};
    float x = threadIdx.x, y0 = 1.f, y1 = 2.f, y2 = 3.f, y3 = 4.f;
    for (int i = 0; i < n_mfma; i++) {
      a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 1.f, a, 0, 0, 0);
      b = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 2.f, b, 0, 0, 0);
      for (int j = 0; j < interleave; j += 4) {
        y0 = fmaf(y0, 1.0001f, 0.5f);
        y1 = fmaf(y1, 1.0001f, 0.5f);
        y2 = fmaf(y2, 1.0001f, 0.5f);
        y3 = fmaf(y3, 1.0001f, 0.5f);
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a[0] + b[3] + y0 + y1 + y2 + y3;
    return;
  }
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 a = {0}, b = {0};
    float x = threadIdx.x;
    for (int i = 0; i < n_mfma; i++) {
      a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 1.f, a, 0, 0, 0);
      b = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 2.f, b, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = a[0] + b[3];
  } else {
    if (!(mode & 2)) return;
    float y0 = threadIdx.x, y1 = 2.f, y2 = 3.f, y3 = 4.f;
    for (int i = 0; i < n_valu; i++) {
      y0 = fmaf(y0, 1.0001f, 0.5f);
      y1 = fmaf(y1, 1.0001f, 0.5f);
      y2 = fmaf(y2, 1.0001f, 0.5f);
      y3 = fmaf(y3, 1.0001f, 0.5f);
    }
    out[blockIdx.x * 512 + threadIdx.x] = y0 + y1 + y2 + y3;
  }
}

// the same experiment with the bf16 matrix instruction (v_mfma_f32_32x32x16_bf16, 8 passes)
__global__ __launch_bounds__(512) void kb(float *out, int n_mfma, int n_valu, int mode) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 a = {0}, b = {0};
    bf16x8 x, y;
    for (int i = 0; i < 8; i++) x[i] = (__bf16)(float)(threadIdx.x + i), y[i] = (__bf16)1.0f;
    for (int i = 0; i < n_mfma; i++) {
      a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a, 0, 0, 0);
      b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, b, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = a[0] + b[3];
  } else {
    if (!(mode & 2)) return;
    float y0 = threadIdx.x, y1 = 2.f, y2 = 3.f, y3 = 4.f;
    for (int i = 0; i < n_valu; i++) {
      y0 = fmaf(y0, 1.0001f, 0.5f);
      y1 = fmaf(y1, 1.0001f, 0.5f);
      y2 = fmaf(y2, 1.0001f, 0.5f);
      y3 = fmaf(y3, 1.0001f, 0.5f);
    }
    out[blockIdx.x * 512 + threadIdx.x] = y0 + y1 + y2 + y3;
  }
}

static float runb(float *d, int nm, int nv, int mode) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(kb, dim3(256), dim3(512), 0, 0, d, nm, nv, mode);
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kb, dim3(256), dim3(512), 0, 0, d, nm, nv, mode);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

static float run(float *d, int nm, int nv, int mode, int il) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, nm, nv, mode, il);
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, nm, nv, mode, il);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float *d;
  hipMalloc(&d, 256 * 512 * 4);
  const int nm = 20000, nv = 80000;  // 40000 MFMAs x 64 cycles = 2.56 M cycles; 320000 FMAs x 4 cycles = 1.28 M cycles
  printf("mfma waves alone  %.3f ms\n", run(d, nm, nv, 1, 0));
  printf("valu waves alone  %.3f ms\n", run(d, nm, nv, 2, 0));
  printf("both              %.3f ms\n", run(d, nm, nv, 3, 0));
  for (int il : {4, 8, 16, 24, 32})
    printf("8 waves, 2 MFMA + %2d FMA per group: %.3f ms (MFMA-only lower bound %.3f ms)\n", il, run(d, 5000, 0, 0, il),
           5000.0 * 2 * 2 * 64 / 2.4e6);
  printf("bf16 32x32x16: mfma waves alone %.3f ms, valu waves alone %.3f ms, both %.3f ms\n", runb(d, 2 * nm, nv, 1),
         runb(d, 2 * nm, nv, 2), runb(d, 2 * nm, nv, 3));
  return 0;
}
