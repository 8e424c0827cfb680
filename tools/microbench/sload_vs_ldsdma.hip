// sload_vs_ldsdma.hip -- does a kernel that reads memory through scalar loads (s_load_dwordx8 from the constant address space,
// the way voxelize_tiles fetches the record of a hit) see wrong data while a kernel full of LDS-DMA (buffer_load_dwordx4 ... lds,
// the way the split-fp16 conv kernels stage their tiles) runs on a second hardware queue of the same process?
//
// LAB.md §3.10: two scorers on two host threads do not reproduce their single-thread bits when the LDS-DMA kernels are in
// the mix; this is the smallest program that asks the hardware the same question.
//   victim   : single-wave workgroups, each checks `iters` pseudo-random records of a read-only table fetched (a) by scalar
//              loads, (b) by vector loads; every dword is a function of its index, mismatches are counted
//   aggressor: 256-thread workgroups with 47 KB of LDS that DMA 1 KB per wave-instruction into it in a loop (mode 1), or do
//              the same traffic through registers and ds_write (mode 0, the control)
// hipcc --offload-arch=gfx950 -O3 -o sload_vs_ldsdma sload_vs_ldsdma.hip ; ./sload_vs_ldsdma
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) {                                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                    \
      exit(1);                                                                                     \
    }                                                                                              \
  } while (0)

typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) u32x8 *ConstRec;
typedef __attribute__((address_space(3))) void *LdsPtr;

__host__ __device__ inline unsigned pattern(unsigned i, unsigned k) { return (i * 2654435761u) ^ (k * 40503u + 0x9e3779b9u); }

__global__ __launch_bounds__(64) void victim(const u32x8 *table, unsigned n, int iters, int use_scalar, unsigned long long *bad) {
  const unsigned wg = blockIdx.x;
  const ConstRec tc = (ConstRec)(const void *)table;
  unsigned long long errs = 0;
  unsigned idx = (wg * 7919u) % n;
  float acc = (float)threadIdx.x;  // some VALU work between the loads, like the density arithmetic
  for (int it = 0; it < iters; it++) {
    idx = (idx * 1664525u + 1013904223u) % n;
    u32x8 r;
    if (use_scalar) {
      r = tc[idx];  // uniform address, constant address space: s_load_dwordx8
    } else {
      const volatile unsigned *q = reinterpret_cast<const volatile unsigned *>(table + idx);
#pragma unroll
      for (int k = 0; k < 8; k++) r[k] = q[k];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) errs += (r[k] != pattern(idx, k)) ? 1ull : 0ull;
#pragma unroll
    for (int k = 0; k < 16; k++) acc = acc * 1.0001f + (float)r[k & 7] * 1e-9f;
  }
  if (acc == 12345.678f) errs += 1000000;  // (keeps acc alive)
  if (errs && threadIdx.x == 0) atomicAdd(bad, errs);
}

__global__ __launch_bounds__(256) void aggressor(const float *src, size_t src_floats, int rounds, int use_dma, float *sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int pieces = 47;  // KB per round, like a (tile + weights) phase of conv3d_h2_kernel
  float s = 0.f;
  const size_t base = ((size_t)blockIdx.x * 9973u * 256u) % (src_floats - 64 * 1024);
  for (int r = 0; r < rounds; r++) {
    if (use_dma) {
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src + base), 0, 64 * 1024 * 4, 0x00020000);
      for (int q = wave; q < pieces; q += 4)
        // (use_dma == 2: every eighth lane asks for an out-of-range offset, the way the conv kernels obtain their zero padding)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LdsPtr)(lds + q * 1024), 16,
                                                 (use_dma == 2 && ((lane + q) & 7) == 0) ? 0x80000000u : (unsigned)((lane * 37 + q * 11 + r) % 4096) * 32u, 0, 0, 0);
      __builtin_amdgcn_s_waitcnt(0x0F70);
    } else {
      for (int q = wave; q < pieces; q += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(src + base + ((size_t)((lane * 37 + q * 11 + r) % 4096)) * 8);
        *reinterpret_cast<float4 *>(lds + q * 1024 + lane * 16) = v;
      }
    }
    __syncthreads();
    s += *reinterpret_cast<float *>(lds + ((tid * 52 + r * 4) % (pieces * 1024 - 4) & ~3));
    __syncthreads();
  }
  if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char **argv) {
  const int seconds_x10 = argc > 1 ? atoi(argv[1]) : 20;  // tenths of a second per configuration
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  const unsigned n = 1 << 16;
  std::vector<unsigned> h((size_t)n * 8);
  for (unsigned i = 0; i < n; i++)
    for (unsigned k = 0; k < 8; k++) h[(size_t)i * 8 + k] = pattern(i, k);
  u32x8 *table;
  CHECK(hipMalloc((void **)&table, h.size() * 4));
  CHECK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const size_t src_floats = (size_t)64 << 20;
  float *src, *sink;
  CHECK(hipMalloc((void **)&src, src_floats * 4));
  CHECK(hipMemset(src, 0x3c, src_floats * 4));
  CHECK(hipMalloc((void **)&sink, 64));
  unsigned long long *bad;
  CHECK(hipMalloc((void **)&bad, 8));
  hipStream_t sv, sa;
  CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(aggressor), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int use_scalar = 1; use_scalar >= 0; use_scalar--)
    for (int mode = 3; mode >= 0; mode--) {  // 3: LDS-DMA with out-of-range lanes, 2: LDS-DMA, 1: through registers, 0: no aggressor
      CHECK(hipMemset(bad, 0, 8));
      hipEvent_t t0, t1;
      CHECK(hipEventCreate(&t0));
      CHECK(hipEventCreate(&t1));
      CHECK(hipEventRecord(t0, sv));
      unsigned long long checked = 0;
      float ms = 0.f;
      int launches = 0;
      do {
        if (mode) hipLaunchKernelGGL(aggressor, dim3(768), dim3(256), 48 * 1024, sa, src, src_floats, 40, mode == 3 ? 2 : mode == 2 ? 1 : 0, sink);
        hipLaunchKernelGGL(victim, dim3(20000), dim3(64), 0, sv, table, n, 64, use_scalar, bad);
        checked += 20000ull * 64 * 8;
        launches++;
        if (launches % 8 == 0) {
          CHECK(hipEventRecord(t1, sv));
          CHECK(hipEventSynchronize(t1));
          CHECK(hipEventElapsedTime(&ms, t0, t1));
        }
      } while (ms < 100.f * seconds_x10);
      CHECK(hipDeviceSynchronize());
      unsigned long long hb = 0;
      CHECK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
      printf("victim reads by %s loads, aggressor %s: %llu dwords checked in %d launches, %llu wrong\n", use_scalar ? "SCALAR" : "vector",
             mode == 3 ? "with LDS-DMA, out-of-range lanes" : mode == 2 ? "with LDS-DMA" : mode == 1 ? "through registers" : "absent", checked, launches, hb);
    }
  return 0;
}
