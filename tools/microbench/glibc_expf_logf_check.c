/* Host check of the expf / logf restatements that gnina_amd/csrc/vina.hip (expf_ref, logf_ref) runs on the device: the
 * same fp64 operations, compared with this host's libm -- expf for every float of [log 2^-150, log 2^128], logf for every positive
 * normal float.
 *   gcc -O2 -fopenmp -ffp-contract=off -mfma tools/microbench/glibc_expf_logf_check.c -o /tmp/el_check -lm && /tmp/el_check
 * glibc 2.35 (x86-64, FMA multiarch variant): 0 mismatches.  (The fused
 * r = fma(N/ln2, x, -k) matters: with the product rounded first two expf arguments differ in the last bit.) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static const uint64_t EXP_TAB[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
static float expf_ref(float x) {
  const double xd = (double)x;
  const double z = 0x1.71547652b82fep+0 * 32 * xd;
  double kd = z + 0x1.8p+52;
  uint64_t ki;
  memcpy(&ki, &kd, 8);
  kd -= 0x1.8p+52;
  const double r = fma(0x1.71547652b82fep+0 * 32, xd, -kd);
  const uint64_t t = EXP_TAB[ki & 31] + (ki << 47);
  double s;
  memcpy(&s, &t, 8);
  const double zz = fma(0x1.c6af84b912394p-5 / 32 / 32 / 32, r, 0x1.ebfce50fac4f3p-3 / 32 / 32);
  const double r2 = r * r;
  double y = fma(0x1.62e42ff0c52d6p-1 / 32, r, 1.0);
  y = fma(zz, r2, y);
  return (float)(y * s);
}
static const double LOG_TAB[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
static float logf_ref(float x) {
  uint32_t ix;
  memcpy(&ix, &x, 4);
  if (ix == 0x3f800000u) return 0.f;
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (tmp >> 19) & 15;
  const int k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & (0x1ffu << 23));
  float zf;
  memcpy(&zf, &iz, 4);
  const double z = zf;
  const double r = fma(z, LOG_TAB[i][0], -1.0);
  const double y0 = fma((double)k, 0x1.62e42fefa39efp-1, LOG_TAB[i][1]);
  const double r2 = r * r;
  double y = fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
  y = fma(-0x1.00ea348b88334p-2, r2, y);
  y = fma(y, r2, y0 + r);
  return (float)y;
}

int main(void) {
  long me = 0, ne = 0, ml = 0, nl = 0;
#pragma omp parallel for reduction(+ : me, ne)
  for (uint32_t b = 0; b < 0x42d00000u; b++)  /* |x| < 104 */
    for (int sg = 0; sg < 2; sg++) {
      const uint32_t bb = b | ((uint32_t)sg << 31);
      float x;
      memcpy(&x, &bb, 4);
      if (x < -0x1.9fe368p6f || x > 0x1.62e42ep6f) continue; /* glibc's underflow / overflow exits */
      const float a = expf(x), c = expf_ref(x);
      if (memcmp(&a, &c, 4)) me++;
      ne++;
    }
#pragma omp parallel for reduction(+ : ml, nl)
  for (uint32_t b = 0x00800000u; b < 0x7f800000u; b++) {
    float x;
    memcpy(&x, &b, 4);
    const float a = logf(x), c = logf_ref(x);
    if (memcmp(&a, &c, 4)) ml++;
    nl++;
  }
  printf("expf: n=%ld mismatches %ld; logf: n=%ld mismatches %ld\n", ne, me, nl, ml);
  return me || ml;
}
