// shader_clock_under_load.hip -- at which shader clock does the chip run a stream of SMALL dependent launches (a per-pose
// DLScorer::score call: tens of workgroups per launch on a 256-CU chip), and what does a dependent launch cost by itself?
//
// Round 6 (LAB.md "Per-pose calls"): the kernels of a B = 1 call take ~1.5x what their instruction counts predict at
// 2.4 GHz and an (almost) empty kernel ~5 us.  Every launch below runs W workgroups of 256 threads that spin for ~20 us of
// the constant 100 MHz clock (s_memrealtime) and report how many shader cycles (s_memtime) went by: the ratio is the shader clock
// DURING the launch.  W = 4, 54, 256, 2048; then the time per launch of an empty kernel in a stream of 2,000.
// hipcc --offload-arch=gfx950 -O3 -o shader_clock_under_load shader_clock_under_load.hip ; ./shader_clock_under_load
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ void spin_kernel(unsigned long long *out, int ticks) {
  const unsigned long long r0 = wall_clock64(), c0 = clock64();
  unsigned long long r1 = r0;
  float x = (float)threadIdx.x;
  while (r1 - r0 < (unsigned long long)ticks) {
    for (int i = 0; i < 64; i++) x = x * 1.0001f + 0.5f;
    r1 = wall_clock64();
  }
  const unsigned long long c1 = clock64();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = r1 - r0;
    out[2 * blockIdx.x + 1] = c1 - c0 + (x == 0.12345f ? 1 : 0);
  }
}
__global__ void empty_kernel(int *p) {
  if (p && threadIdx.x == 1000) *p = 1;
}

int main() {
  CHECK(hipSetDevice(0));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  unsigned long long *d;
  CHECK(hipMalloc(&d, 2 * 4096 * sizeof(unsigned long long)));
  std::vector<unsigned long long> h(2 * 4096);
  for (int W : {4, 54, 256, 2048}) {
    for (int rep = 0; rep < 3; rep++) {
      const int n = 3000;  // ~60 ms of back-to-back launches
      for (int i = 0; i < n; i++) hipLaunchKernelGGL(spin_kernel, dim3(W), dim3(256), 0, s, d, 2000);
      CHECK(hipStreamSynchronize(s));
      CHECK(hipMemcpy(h.data(), d, 2 * W * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      double mhz_min = 1e9, mhz_max = 0;
      for (int w = 0; w < W; w++) {
        const double mhz = (double)h[2 * w + 1] / (double)h[2 * w] * 100.0;
        mhz_min = mhz < mhz_min ? mhz : mhz_min, mhz_max = mhz > mhz_max ? mhz : mhz_max;
      }
      printf("W = %4d workgroups per launch, %d launches: shader clock in the last launch %.0f .. %.0f MHz\n", W, n, mhz_min, mhz_max);
    }
  }
  for (int rep = 0; rep < 3; rep++) {
    const int n = 2000;
    CHECK(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(empty_kernel, dim3(54), dim3(256), 0, s, (int *)nullptr);
    CHECK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("empty kernel, 54 workgroups, %d dependent launches in one stream: %.2f us per launch\n", n, us / n);
  }
  return 0;
}
