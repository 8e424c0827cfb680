// exact_sqrt_div_check.hip -- exhaustive check of the voxelizer's short correctly-rounded sqrt and divide
// (voxelize.hip: density(), the thin-shell path) against the compiler's sqrtf / IEEE divide on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/esd tools/microbench/exact_sqrt_div_check.hip && /tmp/esd
// Every float x in [0.25, 64) (the squared distances the shell path can see are 1..12 A^2) x a list of atom radii.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

__device__ __forceinline__ float sqrt_rn(float x) {  // x normal: v_sqrt_f32 (1 ulp) + one-ulp fix-up
  const float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
  const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
  float r = r_dn <= 0.f ? s_dn : s;
  return r_up > 0.f ? s_up : r;
}
__device__ __forceinline__ float div_rn(float a, float b, float y /* RN(1/b) */) {
  const float q0 = a * y;
  const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, b, a), y, q0);
  return __builtin_fmaf(__builtin_fmaf(-q1, b, a), y, q1);
}

__global__ void check(unsigned lo, unsigned hi, const float *radii, int nr, unsigned long long *bad) {
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned u = lo + blockIdx.x * blockDim.x + threadIdx.x; u < hi; u += stride) {
    const float x = __uint_as_float(u);
    const float s = sqrtf(x);
    if (__float_as_uint(sqrt_rn(x)) != __float_as_uint(s)) atomicAdd(bad, 1ull);
    for (int i = 0; i < nr; i++) {
      const float ar = radii[i];
      const float ref = s / ar;
      if (__float_as_uint(div_rn(s, ar, 1.0f / ar)) != __float_as_uint(ref)) atomicAdd(bad + 1, 1ull);
    }
  }
}

int main() {
  std::vector<float> radii;
  for (int i = 0; i < 400; i++) radii.push_back(0.5f + 0.01f * (float)i);  // 0.5 .. 4.5 A in 0.01 steps (incl. scaled radii)
  const float odd[] = {1.9f, 1.8f, 1.7f, 2.0f, 1.5f, 1.2f, 2.2f, 1.4f, 1.75f, 2.3f, 0.37f, 1.9999999f, 1.0000001f, 1.99999988f};
  for (float r : odd) radii.push_back(r);
  float *d_r;
  unsigned long long *d_bad, bad[2] = {0, 0};
  hipMalloc(&d_r, radii.size() * 4);
  hipMemcpy(d_r, radii.data(), radii.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&d_bad, 16);
  hipMemset(d_bad, 0, 16);
  float lo = 0.25f, hi = 64.f;
  unsigned ulo, uhi;
  memcpy(&ulo, &lo, 4);
  memcpy(&uhi, &hi, 4);
  hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, ulo, uhi, d_r, (int)radii.size(), d_bad);
  hipDeviceSynchronize();
  hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost);
  printf("floats checked: %u x %zu radii; sqrt mismatches: %llu; divide mismatches: %llu\n", uhi - ulo, radii.size(), bad[0], bad[1]);
  return bad[0] || bad[1];
}
