/* Host check of the sinf / cosf restatement that gnina_amd/csrc/vina.hip (sincos_ref) runs on the device: the same
 * fp64 operations, compared with this host's libm for every float of [-100, 100].
 *   gcc -O2 -fopenmp -ffp-contract=off -mfma tools/microbench/glibc_sincosf_check.c -o /tmp/sc_check -lm && /tmp/sc_check
 * glibc 2.35 (x86-64, FMA multiarch variant): 0 mismatches in 2,240,806,914 arguments.  Without fused mul-adds
 * (-DNO_FMA: what a CPU without FMA would select) 28 arguments differ in the last bit. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#ifdef NO_FMA
#define FMA(a, b, c) ((a) * (b) + (c))
#else
#define FMA(a, b, c) fma((a), (b), (c))
#endif

static void sincos_ref(float y, float *sn, float *cs) {
  const double x = (double)y;
  const double r = x * 0x1.45F306DC9C883p+23;
  const int n = ((int32_t)r + 0x800000) >> 24;
  const double xr = FMA(-(double)n, 0x1.921FB54442D18p0, x);
  const double xs = ((n & 1) ^ ((n >> 1) & 1)) ? -xr : xr;
  const double x2 = xr * xr;
  const double x3 = xs * x2;
  const double s1 = FMA(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
  const double x7 = x3 * x2;
  const double s = FMA(x3, -0x1.555545995a603p-3, xs);
  const float sp = (float)FMA(x7, s1, s);
  const double x4 = x2 * x2;
  const double c2 = FMA(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
  const double c1 = FMA(x2, -0x1.ffffffd0c621cp-2, 1.0);
  const double x6 = x4 * x2;
  const double c = FMA(x4, 0x1.55553e1068f19p-5, c1);
  float cp = (float)FMA(x6, c2, c);
  if (n & 2) cp = -cp;
  *sn = (n & 1) ? cp : sp;
  *cs = (n & 1) ? sp : cp;
  uint32_t u;
  memcpy(&u, &y, 4);
  if (((u >> 20) & 0x7ff) < 0x398u) {
    *sn = y;
    *cs = 1.0f;
  }
}

int main(void) {
  const float hi = 100.f;
  uint32_t hb;
  memcpy(&hb, &hi, 4);
  long ms = 0, mc = 0, n = 0;
#pragma omp parallel for reduction(+ : ms, mc, n)
  for (uint32_t b = 0; b <= hb; b++)
    for (int sg = 0; sg < 2; sg++) {
      const uint32_t bb = b | ((uint32_t)sg << 31);
      float x, s, c;
      memcpy(&x, &bb, 4);
      sincos_ref(x, &s, &c);
      const float s0 = sinf(x), c0 = cosf(x);
      if (memcmp(&s, &s0, 4)) ms++;
      if (memcmp(&c, &c0, 4)) mc++;
      n++;
    }
  printf("n=%ld sinf mismatches %ld cosf mismatches %ld\n", n, ms, mc);
  return ms || mc;
}
