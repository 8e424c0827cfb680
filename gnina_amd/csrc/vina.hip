// vina.hip -- smina/Vina scoring and BFGS local optimisation for gfx950 (MI355X).
//
// Replaces, behind the igrid / quasi_newton seam, the CPU loop nest
//   quasi_newton::operator() (quasi_newton.cpp:49-83) -> bfgs<> (bfgs.h:357-502)
//     -> model::eval_deriv (model.cu:202-225): set(conf) (tree.h:361-373), cache::eval_deriv
//        (cache.cpp:65-83 -> grid.cpp:96-186), eval_interacting_pairs_deriv (model.cu:38-60),
//        ligands.derivative (tree.h:374-382)
// and cache::populate (cache.cpp:104-184).
//
// MI355X design (not the reference's experimental --gpu_docking layout, which runs one BFGS per
// kernel launch in a single block with dynamic parallelism, bfgs.cu:229-337):
//  * ONE 64-lane wavefront owns ONE conformation for its whole life -- evaluation, line search and the
//    quasi-Newton update (and, in the Monte-Carlo kernel, the whole chain) stay inside one kernel; nothing
//    returns to the host between function evaluations.  What a dependent chain touches lives in registers
//    (node constants, BFGS vectors, rows of H), the rest in a private LDS workspace of ~12 KB.
//  * Lanes map to tree nodes for the frames (one level per step on branched trees), to atoms for the
//    receptor-grid term (8-point trilinear gather per atom), to pairs for the intramolecular term, to atoms
//    again for the per-atom lists of pair forces (contribution slots, pair order), to nodes for the
//    force/torque sums: no atomics, deterministic.  The energy is a fixed-order butterfly (DPP).
//  * Few chains are latency bound: W waves per chain evaluate the trials of a line search at once
//    (WaveTeam) and take the first accepted one in trial order.  Thousands of chains are throughput bound:
//    the same code with fewer loads in flight and H in LDS fits two to three waves per SIMD
//    (vina_mc_tp_kernel, vina_bfgs_kernel<true>).  A screen's ligands share one launch (VinaMcArgs::ligs,
//    VinaEnv::ligs).  All of these give the same bits: only batching and residence differ.
//  * Ordering points between the lanes of one wave are compiler fences (wave_sync): LDS executes a wave's
//    instructions in order; __syncthreads() only where waves of a team exchange data.
//  * All arithmetic is fp32 in the reference's operation order (compiled with -ffp-contract=off);
//    only sinf/cosf and the reduction order differ from the CPU oracle.
#include <cstdlib>
#include <mutex>
#include <set>

#include "common.h"
#include "vina.h"
#include "options.h"

namespace mig {

#define VPI 3.14159265358979323846f
#define VEPS 1.1920928955078125e-07f
#define VMAXFL 3.402823466e+38f

__device__ __forceinline__ int tri_idx(int t1, int t2) {
  int a = t1 < t2 ? t1 : t2, b = t1 < t2 ? t2 : t1;
  return a + b * (b + 1) / 2;
}

// curl.h:29-42
__device__ __forceinline__ void curl3(float &e, float &dx, float &dy, float &dz, float v) {
  if (e > 0 && v < 0.1f * VMAXFL) {
    float tmp = (v < VEPS) ? 0.f : (v / (v + e));
    e *= tmp;
    float t2 = tmp * tmp;
    dx *= t2;
    dy *= t2;
    dz *= t2;
  }
}
__device__ __forceinline__ void curl1(float &e, float v) {
  if (e > 0 && v < 0.1f * VMAXFL) {
    float tmp = (v < VEPS) ? 0.f : (v / (v + e));
    e *= tmp;
  }
}

// atom_constants.h:101-133 (xs_radius, xs_hydrophobe / xs_donor / xs_acceptor)
__constant__ float c_xs_radius[kVinaTypes] = {0.37f, 0.37f, 1.9f, 1.9f, 1.9f, 1.9f, 1.8f, 1.8f, 1.8f, 1.8f,
                                              1.7f,  1.7f,  1.7f, 1.7f, 2.0f, 2.0f, 2.1f, 1.5f, 1.8f, 2.0f,
                                              2.2f,  1.2f,  1.2f, 1.2f, 1.2f, 1.2f, 1.2f, 1.92f};
__constant__ unsigned char c_hyd[kVinaTypes] = {0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                                0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1};
__constant__ unsigned char c_don[kVinaTypes] = {0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 1, 1, 0,
                                                0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 0};
__constant__ unsigned char c_acc[kVinaTypes] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 1, 1,
                                                0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ float slope_step_d(float x_bad, float x_good, float x) {  // everything.h:207-216
  if (x_bad < x_good) {
    if (x <= x_bad) return 0.f;
    if (x >= x_good) return 1.f;
  } else {
    if (x >= x_bad) return 0.f;
    if (x <= x_good) return 1.f;
  }
  return (x - x_bad) / (x_good - x_bad);
}

__device__ __forceinline__ float expf_ref(float x);  // glibc's expf restated, below
// weighted_terms::eval_fast for the default Vina term set (weighted_terms.cpp:54-68; everything.h)
__device__ float pair_energy_exact(const float *w, int t1, int t2, float r) {
  const float opt = c_xs_radius[t1] + c_xs_radius[t2];
  float acc = 0.f;
  float q = (r - (opt + 0.0f)) / 0.5f;
  acc += w[0] * expf_ref(-(q * q));
  q = (r - (opt + 3.0f)) / 2.0f;
  acc += w[1] * expf_ref(-(q * q));
  const float d = r - (opt + 0.0f);
  acc += w[2] * (d > 0 ? 0.0f : d * d);
  acc += w[3] * ((c_hyd[t1] && c_hyd[t2]) ? slope_step_d(1.5f, 0.5f, r - opt) : 0.0f);
  const bool hb = (c_don[t1] && c_acc[t2]) || (c_don[t2] && c_acc[t1]);
  acc += w[4] * (hb ? slope_step_d(0.0f, -0.7f, r - opt) : 0.0f);
  return acc;
}

// Spline::eval_deriv at r (splines.h:100-118) of the pair's spline: value and dE/dr
__device__ __forceinline__ void spline_eval(const float4 *sp, int sp_n, float fraction, float cutoff, int t1, int t2, float r,
                                            float &val, float &dx) {
  val = dx = 0.f;
  if (r >= cutoff) return;
  int idx = (int)(r / fraction);
  if (idx > sp_n - 1) idx = sp_n - 1;  // (r a rounding below the cutoff: the reference indexes one past its array here)
  const float4 k = sp[(long)tri_idx(t1, t2) * sp_n + idx];
  const float lx = r - (float)idx * fraction;
  val = ((k.x * lx + k.y) * lx + k.z) * lx + k.w;
  dx = (3 * k.x * lx + 2 * k.y) * lx + k.z;
}

// precalculate::eval_deriv: interpolated table (precalculate.h:97-133), numeric exact (:467-490) or splines (:413-442)
__device__ __forceinline__ void prec_eval_deriv(const VinaEnv &env, int t1, int t2, float r2, float &e, float &dor) {
  if (env.spline && !env.exact) {
    const float r = sqrtf(r2);
    float dx;
    spline_eval(env.spline, env.sp_n, env.sp_fraction, env.cutoff, t1, t2, r, e, dx);
    dor = dx / r;
    return;
  }
  if (env.exact) {
    const float delta = 0.000005f;
    const float r = sqrtf(r2);
    const float X = pair_energy_exact(env.w5, t1, t2, r);
    const float rhi = r + delta;
    float rlo = r - delta;
    if (rlo < 0) rlo = 0;
    const float W = pair_energy_exact(env.w5, t1, t2, rlo), Y = pair_energy_exact(env.w5, t1, t2, rhi);
    e = X;
    dor = ((Y - W) / (rhi - rlo)) / r;
  } else {
    const long base = (long)tri_idx(t1, t2) * env.n;
    const float r2f = env.factor * r2;
    const int i1 = (int)r2f;
    const float rem = r2f - (float)i1;
    const float2 s1 = env.smooth[base + i1], s2 = env.smooth[base + i1 + 1];
    e = s1.x + rem * (s2.x - s1.x);
    dor = s1.y + rem * (s2.y - s1.y);
  }
}

// precalculate::eval = eval_fast: midpoint table (precalculate.h:90-95) or exact E(r)
__device__ __forceinline__ float prec_eval(const VinaEnv &env, int t1, int t2, float r2) {
  if (env.exact) return pair_energy_exact(env.w5, t1, t2, sqrtf(r2));
  if (env.spline) {  // precalculate_splines::eval_fast, precalculate.h:407-411
    float e, dx;
    spline_eval(env.spline, env.sp_n, env.sp_fraction, env.cutoff, t1, t2, sqrtf(r2), e, dx);
    return e;
  }
  return env.fast[(long)tri_idx(t1, t2) * env.n + (int)(env.factor * r2)];
}

// grid::evaluate_aux, grid.cpp:96-186, in two halves so that the eight corner loads (L2 latency) of an atom
// can be in flight while the intramolecular pair stage runs: grid_fetch locates the cell and issues the loads,
// grid_finish interpolates.
struct GridTap {
  float f[8];  // f000 f100 f010 f110 f001 f101 f011 f111
  float s[3];
  int region[3];
  float penalty;
};

__device__ __forceinline__ void grid_fetch(const VinaGridGeom &g, const float *data, float lx, float ly, float lz,
                                           float slope, GridTap &t) {
  float s[3] = {(lx - g.init[0]) * g.factor[0], (ly - g.init[1]) * g.factor[1], (lz - g.init[2]) * g.factor[2]};
  float miss[3] = {0.f, 0.f, 0.f};
  int a[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    if (s[i] < 0) {
      miss[i] = -s[i];
      t.region[i] = -1;
      a[i] = 0;
      s[i] = 0;
    } else if (s[i] >= g.dim_m1[i]) {
      miss[i] = s[i] - g.dim_m1[i];
      t.region[i] = 1;
      a[i] = g.dim[i] - 2;
      s[i] = 1;
    } else {
      t.region[i] = 0;
      a[i] = (int)s[i];
      s[i] -= (float)a[i];
    }
    t.s[i] = s[i];
  }
  t.penalty = slope * (miss[0] * g.factor_inv[0] + miss[1] * g.factor_inv[1] + miss[2] * g.factor_inv[2]);
  const long sx = 1, sy = g.dim[0], sz = (long)g.dim[0] * g.dim[1];
  const float *p = data + a[0] * sx + a[1] * sy + a[2] * sz;
  t.f[0] = p[0], t.f[1] = p[sx], t.f[2] = p[sy], t.f[3] = p[sx + sy];
  t.f[4] = p[sz], t.f[5] = p[sz + sx], t.f[6] = p[sz + sy], t.f[7] = p[sz + sx + sy];
}

template <bool DERIV>
__device__ __forceinline__ float grid_finish(const VinaGridGeom &g, const GridTap &t, float slope, float v, float &ox,
                                             float &oy, float &oz) {
  const float f000 = t.f[0], f100 = t.f[1], f010 = t.f[2], f110 = t.f[3];
  const float f001 = t.f[4], f101 = t.f[5], f011 = t.f[6], f111 = t.f[7];
  const float x = t.s[0], y = t.s[1], z = t.s[2], mx = 1 - x, my = 1 - y, mz = 1 - z;
  float f = f000 * mx * my * mz + f100 * x * my * mz + f010 * mx * y * mz + f110 * x * y * mz + f001 * mx * my * z +
            f101 * x * my * z + f011 * mx * y * z + f111 * x * y * z;
  if (DERIV) {
    float gx = f000 * (-1) * my * mz + f100 * 1 * my * mz + f010 * (-1) * y * mz + f110 * 1 * y * mz +
               f001 * (-1) * my * z + f101 * 1 * my * z + f011 * (-1) * y * z + f111 * 1 * y * z;
    float gy = f000 * mx * (-1) * mz + f100 * x * (-1) * mz + f010 * mx * 1 * mz + f110 * x * 1 * mz +
               f001 * mx * (-1) * z + f101 * x * (-1) * z + f011 * mx * 1 * z + f111 * x * 1 * z;
    float gz = f000 * mx * my * (-1) + f100 * x * my * (-1) + f010 * mx * y * (-1) + f110 * x * y * (-1) +
               f001 * mx * my * 1 + f101 * x * my * 1 + f011 * mx * y * 1 + f111 * x * y * 1;
    curl3(f, gx, gy, gz, v);
    ox = g.factor[0] * (t.region[0] == 0 ? gx : 0.f) + slope * (float)t.region[0];
    oy = g.factor[1] * (t.region[1] == 0 ? gy : 0.f) + slope * (float)t.region[1];
    oz = g.factor[2] * (t.region[2] == 0 ? gz : 0.f) + slope * (float)t.region[2];
    return f + t.penalty;
  }
  curl1(f, v);
  return f + t.penalty;
}

template <bool DERIV>
__device__ __forceinline__ float grid_evaluate(const VinaGridGeom &g, const float *data, float lx, float ly, float lz,
                                               float slope, float v, float &ox, float &oy, float &oz) {
  GridTap t;
  grid_fetch(g, data, lx, ly, lz, slope, t);
  return grid_finish<DERIV>(g, t, slope, v, ox, oy, oz);
}

// ---- quaternion helpers (quaternion.h:243-303,327-364) ------------------------------------------
__device__ __forceinline__ float norm_angle(float x) {  // g_normalize_angle, quaternion.h:259-282
  while (x > 3 * VPI) {
    float n = (x - VPI) / (2 * VPI);
    x -= 2 * VPI * ceilf(n);
  }
  while (x < -3 * VPI) {
    float n = (-x - VPI) / (2 * VPI);
    x += 2 * VPI * ceilf(n);
  }
  if (x > VPI)
    x -= 2 * VPI;
  else if (x < -VPI)
    x += 2 * VPI;
  return x;
}

// sinf / cosf exactly as the reference's host computes them.  gnina's tree.h / quaternion.h call std::sin / std::cos
// on fl = float, i.e. glibc's sinf / cosf (2.35: sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h,
// s_sincosf_data.c -- the ARM optimized-routines algorithm; on x86-64 with FMA the multiarch build, whose mul-adds are
// fused).  Those are NOT correctly rounded (worst case 0.56 ulp): rounding an accurate fp64 sine differs from them on
// 0.1 % of the arguments, ocml's sincosf on more -- and one differing last bit sends a BFGS trajectory elsewhere.  So
// the algorithm is restated here operation by operation in fp64: one reduction step x - n * (pi/2) with
// n = round(x * 2/pi) taken from a 2^24-scaled integer conversion, then the degree-7 sine or degree-8 cosine
// polynomial of the quadrant.  Checked against the host's sinf / cosf for all 2.24e9 floats of [-100, 100]
// (tools/microbench/glibc_sincosf_check.c: 0 mismatches) and on the device by tests/test_gpu_vina_ref.py.
// Domain: |y| < 120 (glibc switches to a 192-bit reduction above; every caller passes half of a normalised angle).
__device__ __forceinline__ void sincos_ref(float y, float &sn, float &cs) {
  const double x = (double)y;
  const double r = x * 0x1.45F306DC9C883p+23;                         // 2/pi * 2^24
  const int n = ((int)r + 0x800000) >> 24;                             // quadrant, rounded to nearest
  const double xr = __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);  // x - n * pi/2, |xr| <= pi/4
  const double xs = (n & 1) ^ ((n >> 1) & 1) ? -xr : xr;               // sign[n & 3] = {1, -1, -1, 1}
  const double x2 = xr * xr;
  // sine polynomial (sinf_poly, n even)
  const double x3 = xs * x2;
  const double s1 = __builtin_fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
  const double x7 = x3 * x2;
  const double s = __builtin_fma(x3, -0x1.555545995a603p-3, xs);
  const float sp = (float)__builtin_fma(x7, s1, s);
  // cosine polynomial (sinf_poly, n odd); the table's second entry (quadrants 2, 3) negates every coefficient
  const double x4 = x2 * x2;
  const double c2 = __builtin_fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
  const double c1 = __builtin_fma(x2, -0x1.ffffffd0c621cp-2, 1.0);
  const double x6 = x4 * x2;
  const double c = __builtin_fma(x4, 0x1.55553e1068f19p-5, c1);
  float cp = (float)__builtin_fma(x6, c2, c);
  if (n & 2) cp = -cp;
  sn = (n & 1) ? cp : sp;
  cs = (n & 1) ? sp : cp;
  const unsigned top = (__float_as_uint(y) >> 20) & 0x7ff;
  if (top < 0x398u) {  // |y| < 2^-12: sinf returns y, cosf returns 1
    sn = y;
    cs = 1.0f;
  }
}

// expf / logf exactly as the reference's host computes them (std::exp / std::log on fl = float: glibc 2.35's
// e_expf.c / e_logf.c, again the ARM optimized-routines algorithms with their mul-adds fused as the x86-64 FMA build
// has them): restated in fp64 like sincos_ref and checked the same way -- tools/microbench/glibc_expf_logf_check.c
// compares these operations with the host's libm for every float of expf's [-87, 88) and every positive normal float
// of logf: 0 mismatches.  Users: the Metropolis criterion (monte_carlo.cpp:38-42), random_normal's Box-Muller
// (conf::randomize, through oracle/ref_shims' boost::normal_distribution) and precalculate_exact's Gaussians.
__constant__ unsigned long long c_exp2f_tab[32] = {  // asuint64(2^(i/32)) - (i << 47), 2^(i/32) correctly rounded
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};
__device__ __forceinline__ float expf_ref(float x) {
  if (x < -0x1.9fe368p6f) return 0.f;        // below log(2^-150): glibc's __math_uflowf
  if (x > 0x1.62e42ep6f) return __builtin_inff();  // above log(2^128)
  const double xd = (double)x;
  const double z = 0x1.71547652b82fep+0 * 32 * xd;             // x * N / ln 2, N = 32
  double kd = z + 0x1.8p+52;                                    // round to nearest integer: low mantissa bits = k
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd -= 0x1.8p+52;
  const double r = __builtin_fma(0x1.71547652b82fep+0 * 32, xd, -kd);
  const double s = __longlong_as_double((long long)(c_exp2f_tab[ki & 31] + (ki << 47)));
  const double zz = __builtin_fma(0x1.c6af84b912394p-5 / 32 / 32 / 32, r, 0x1.ebfce50fac4f3p-3 / 32 / 32);
  const double r2 = r * r;
  double y = __builtin_fma(0x1.62e42ff0c52d6p-1 / 32, r, 1.0);
  y = __builtin_fma(zz, r2, y);
  return (float)(y * s);
}

__constant__ double c_logf_tab[16][2] = {  // (1/c, log c) of the sixteen sub-intervals of [0x1.66p-1, 0x1.66p0)
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
__device__ __forceinline__ float logf_ref(float x) {  // x positive and normal (the callers' arguments are in (2^-24, 1])
  const unsigned ix = __float_as_uint(x);
  if (ix == 0x3f800000u) return 0.f;
  const unsigned tmp = ix - 0x3f330000u;
  const int i = (tmp >> 19) & 15;
  const int k = (int)tmp >> 23;
  const double z = (double)__uint_as_float(ix - (tmp & (0x1ffu << 23)));
  const double r = __builtin_fma(z, c_logf_tab[i][0], -1.0);
  const double y0 = __builtin_fma((double)k, 0x1.62e42fefa39efp-1, c_logf_tab[i][1]);
  const double r2 = r * r;
  double y = __builtin_fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
  y = __builtin_fma(-0x1.00ea348b88334p-2, r2, y);
  y = __builtin_fma(y, r2, y0 + r);
  return (float)y;
}

// acosf as the host computes it (glibc 2.35 sysdeps/ieee754/flt-32/e_acosf.c = fdlibm's rational approximation in
// plain fp32 arithmetic, no fused operations): quaternion_to_angle (quaternion.cu:46-62) inside compute_lambdamin.
// Checked against the host for every float of [-1, 1] (tools/microbench/glibc_acosf_check.c: 0 mismatches).
__device__ __forceinline__ float acosf_ref(float x) {
  const float pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
  const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
              pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f,
              qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
  const int hx = __float_as_int(x), ix = hx & 0x7fffffff;
  if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
  if (ix > 0x3f800000) return (x - x) / (x - x);
  if (ix < 0x3f000000) {  // |x| < 0.5
    if (ix <= 0x23000000) return pio2_hi + pio2_lo;
    const float z = x * x;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (hx < 0) {  // x < -0.5
    const float z = (1.0f + x) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float s = sqrtf(z);
    const float r = p / q;
    const float w = r * s - pio2_lo;
    return pi - 2.0f * (s + w);
  }
  const float z = (1.0f - x) * 0.5f;  // x > 0.5
  const float s = sqrtf(z);
  const float df = __int_as_float(__float_as_int(s) & 0xfffff000);
  const float c = (z - df * df) / (s + df);
  const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const float r = p / q;
  const float w = r * s + c;
  return 2.0f * (df + w);
}

__device__ __forceinline__ void angle_to_quat(float ax, float ay, float az, float angle, float *q) {
  angle = norm_angle(angle);
  float c, s;
#ifdef MIG_FAST_SINCOS
  s = __sinf(angle / 2);
  c = __cosf(angle / 2);
#else
  sincos_ref(angle / 2, s, c);
#endif
  q[0] = c;
  q[1] = s * ax;
  q[2] = s * ay;
  q[3] = s * az;
}

__device__ __forceinline__ void quat_mul(const float *l, const float *r, float *o) {
  const float a = l[0], b = l[1], c = l[2], d = l[3], ar = r[0], br = r[1], cr = r[2], dr = r[3];
  o[0] = +a * ar - b * br - c * cr - d * dr;
  o[1] = +a * br + b * ar + c * dr - d * cr;
  o[2] = +a * cr - b * dr + c * ar + d * br;
  o[3] = +a * dr + b * cr - c * br + d * ar;
}

__device__ __forceinline__ void quat_norm_approx(float *q) {
  const float s = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (fabsf(s - 1) < 1e-6f) return;
  const float inv = 1 / sqrtf(s);
  q[0] *= inv;
  q[1] *= inv;
  q[2] *= inv;
  q[3] *= inv;
}

__device__ __forceinline__ void quat_to_r3(const float *q, float *m) {  // m[i + 3 j] = M(i, j)
  const float a = q[0], b = q[1], c = q[2], d = q[3];
  const float aa = a * a, ab = a * b, ac = a * c, ad = a * d, bb = b * b, bc = b * c, bd = b * d, cc = c * c,
              cd = c * d, dd = d * d;
  m[0] = (aa + bb - cc - dd);
  m[3] = 2 * (-ad + bc);
  m[6] = 2 * (ac + bd);
  m[1] = 2 * (ad + bc);
  m[4] = (aa - bb + cc - dd);
  m[7] = 2 * (-ab + cd);
  m[2] = 2 * (-ac + bd);
  m[5] = 2 * (ab + cd);
  m[8] = (aa - bb - cc + dd);
}

__device__ __forceinline__ void mat_vec(const float *m, float vx, float vy, float vz, float &ox, float &oy,
                                        float &oz) {
  ox = m[0] * vx + m[3] * vy + m[6] * vz;
  oy = m[1] * vx + m[4] * vy + m[7] * vz;
  oz = m[2] * vx + m[5] * vy + m[8] * vz;
}

// ---- per-wave LDS workspace -----------------------------------------------------------------------
struct WaveWork {
  float *origin, *axis, *M, *q;  // node frames
  float *coords, *forces;        // [3 n_atoms]
  float *cx, *cy, *cz;           // [n_slots] each: per-atom lists of pair-force contributions, one array per
                                 // component so that four consecutive slots are one 16-byte read (eval_conf stage 4/5)
  float4 *ft;                    // [2 n_atoms] per-atom force and torque about its node origin (fold_forces)
};

// Two tunings of the same arithmetic.  Latency (few chains, one wave per SIMD anyway): six pair look-ups per lane in
// flight, H rows in registers.  Throughput (thousands of chains): three look-ups, H in LDS -- half the registers,
// twice the resident waves.  Results are bit-identical: only batching and residence change, not the operations.
constexpr int kPairGroup = 6, kPairGroupTp = 3;  // intramolecular pairs a lane looks up together (eval_conf stage 4)
constexpr int kHReg = 24, kHRegTp = 4;           // variables up to which lane i keeps row i of H in registers

// Ordering point between the lanes of ONE wave (private LDS workspace): LDS instructions of a wave execute in
// issue order, so no s_barrier and no counter drain is needed -- only the compiler must not move LDS accesses
// across it.  Data shared between the waves of a workgroup still goes through __syncthreads().
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float *carve(float *&p, int n) {
  float *r = p;
  p += (n + 3) & ~3;
  return r;
}

// init = false only measures the footprint (multi-wave kernels place one workspace per wave)
__device__ __forceinline__ WaveWork carve_work(float *&p, const VinaLigand &L, bool init = true) {
  WaveWork w;
  const int n_slots = (2 * L.n_pairs + 3 * L.n_atoms + 3) & ~3;  // upper bound of the ligand's slot count
  w.cx = carve(p, n_slots);
  w.cy = carve(p, n_slots);
  w.cz = carve(p, n_slots);
  w.ft = reinterpret_cast<float4 *>(carve(p, 8 * L.n_atoms));
  w.origin = carve(p, 3 * L.n_nodes);
  w.axis = carve(p, 3 * L.n_nodes);
  w.M = carve(p, 9 * L.n_nodes);
  w.q = carve(p, 4 * L.n_nodes);
  w.coords = carve(p, 3 * L.n_atoms);
  w.forces = carve(p, 3 * L.n_atoms);
  if (init) {
    for (int i = threadIdx.x & 63; i < 3 * n_slots; i += 64) w.cx[i] = 0.f;  // cx, cy, cz are contiguous
    wave_sync();
  }
  return w;
}

// Copy the (small, read-many) ligand description from global memory into LDS so that the sequential
// tree walk and the per-pair / per-atom index look-ups of every evaluation hit LDS (~64 cycles) instead of
// L2 (~200-500 cycles).  Returns a VinaLigand whose pointers address the LDS copies.
__device__ __forceinline__ VinaLigand stage_ligand(const VinaLigand &G, float *&p) {
  const int tid = threadIdx.x, nthr = blockDim.x;  // one copy per workgroup, shared by its waves
  VinaLigand L = G;
  auto cp_i = [&](const int *src, int n) -> const int * {
    int *dst = reinterpret_cast<int *>(carve(p, n));
    for (int i = tid; i < n; i += nthr) dst[i] = src[i];
    return dst;
  };
  auto cp_f = [&](const float *src, int n) -> const float * {
    float *dst = carve(p, n);
    for (int i = tid; i < n; i += nthr) dst[i] = src[i];
    return dst;
  };
  L.smt = cp_i(G.smt, G.n_atoms);
  L.node_of_atom = cp_i(G.node_of_atom, G.n_atoms);
  L.parent = cp_i(G.parent, G.n_nodes);
  L.abeg = cp_i(G.abeg, G.n_nodes);
  L.aend = cp_i(G.aend, G.n_nodes);
  L.child_start = cp_i(G.child_start, G.n_nodes + 1);
  L.child_list = cp_i(G.child_list, G.n_nodes);  // a tree has n_nodes - 1 edges
  L.pairs = reinterpret_cast<const int2 *>(cp_i(reinterpret_cast<const int *>(G.pairs), 2 * G.n_pairs));
  L.slot_start = cp_i(G.slot_start, G.n_atoms + 1);
  L.pair_slots = reinterpret_cast<const int2 *>(cp_i(reinterpret_cast<const int *>(G.pair_slots), 2 * G.n_pairs));
  L.heavy_list = cp_i(G.heavy_list, G.n_heavy);
  L.depth = cp_i(G.depth, G.n_nodes);
  if (G.pair_cap) L.pair_cap = cp_i(G.pair_cap, G.n_pairs);
  L.local_xyz = cp_f(G.local_xyz, 3 * G.n_atoms);
  L.rel_origin = cp_f(G.rel_origin, 3 * G.n_nodes);
  L.rel_axis = cp_f(G.rel_axis, 3 * G.n_nodes);
  __syncthreads();
  return L;
}

// Workspaces above the default 64 KB of dynamic LDS (large ligands; several waves per chain) need the kernel's
// limit raised once; gfx950 has 160 KB per workgroup.
constexpr size_t kVinaMaxLds = 160 * 1024;
template <class K>
static void big_lds(K kernel) {
  ensure_max_lds(reinterpret_cast<const void *>(kernel), (int)kVinaMaxLds);  // once per device and kernel
}

static size_t pad4(size_t n) { return (n + 3) & ~(size_t)3; }

// Few chains (a single docking job) are latency bound: keep the ligand description in LDS.  Many chains
// (screening) are throughput bound: spend the LDS on occupancy instead.
static bool want_stage(int B) { return B <= 2048; }

static size_t ligand_lds_floats(int na, int nn, int np, int nh) {
  return 2 * pad4(na) + 4 * pad4(nn) + pad4(nn + 1) + pad4(nn) + 2 * pad4(2 * (size_t)np) + pad4(na + 1) + pad4(nh) +
         pad4(3 * (size_t)na) + 2 * pad4(3 * (size_t)nn) + pad4(np);  // (+ pair_cap, only present with flexible residues)
}

size_t vina_wave_lds_bytes(int n_atoms, int n_nodes, int n_pairs, bool bfgs, bool stage) {
  size_t f = 3 * pad4(2 * (size_t)n_pairs + 3 * (size_t)n_atoms) + pad4(8 * (size_t)n_atoms) + 2 * pad4(3 * (size_t)n_nodes) + pad4(9 * (size_t)n_nodes) +
             pad4(4 * (size_t)n_nodes) + 2 * pad4(3 * (size_t)n_atoms);
  const size_t nt = n_nodes - 1, n = 6 + nt, nc = 7 + nt;
  f += pad4(nc);  // conf being evaluated
  f += pad4(n);   // change
  if (bfgs) f += pad4(nc) + pad4(n) + pad4(n * (n + 1) / 2);
  if (stage) f += ligand_lds_floats(n_atoms, n_nodes, n_pairs, n_atoms);  // LDS copy of the ligand description
  return f * sizeof(float);
}

// Sum over the 64 lanes, returned in every lane.  The association is the xor butterfly's (1, 2, 4, 8, 16, 32):
// quads, octets and rows are combined with DPP (quad_perm / row_half_mirror / row_mirror: after each step the
// lanes of a group hold the same partial sum, so mirroring and xor pick the same partner value), rows with
// row_bcast 15 / 31, and the total is read from lane 63 -- register to register, no ds_bpermute round trips.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, t);
}

__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float rl_any(float v, int lane) {  // lane: any wave-uniform value
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), __builtin_amdgcn_readfirstlane(lane)));
}

// Strict-order sums (VinaEnv::strict, mi_vina_set_strict_order): the reference adds energies one after the other --
// atoms in index order (cache.cpp:65-83), pairs in list order (model.cu:38-60) -- and fp32 addition does not
// associate, so the butterfly above gives the same energy only to the last bits.  With the flag set every energy sum
// of eval_conf runs in the reference's order instead: `v` holds one term per lane, `mask` (wave-uniform) the lanes
// whose term the reference would add; skipped lanes would add +0 to a sum that can never be -0, i.e. nothing.
// Together with sincos_ref this makes energies, forces and with them whole BFGS / Monte-Carlo trajectories
// bit-identical to the reference's (tests/test_gpu_vina_ref.py); it costs ~1 us per evaluation, hence a mode.
__device__ __forceinline__ void seq_add(float &acc, float v, unsigned long long mask) {
  while (mask) {
    const int l = __builtin_ctzll(mask);
    acc += rl_any(v, l);
    mask &= mask - 1;
  }
}

// ligands.derivative(coords, minus_forces, g) (model.cu:223; tree.h:133-140,293-401): per-atom forces in
// w.forces + the node frames of the conformation just set -> change[6 + T].
__device__ __forceinline__ float rl(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

__device__ __forceinline__ void fold_forces(const VinaLigand &L, const WaveWork &w, float *change) {
  const int lane = threadIdx.x & 63;
  // 6. per-node force / torque about the node origin, atoms in index order (tree.h:133-140).  The cross
  // products are formed one atom per lane first; lane k then owns node k and only adds, four atoms per LDS
  // latency, keeping the six sums and the node origin in registers.
  for (int i = lane; i < L.n_atoms; i += 64) {
    const int k = L.node_of_atom[i];
    if (k < 0) continue;  // inflex atom: its forces are ignored (model.cu:221)
    const float rx = w.coords[3 * i] - w.origin[3 * k], ry = w.coords[3 * i + 1] - w.origin[3 * k + 1],
                rz = w.coords[3 * i + 2] - w.origin[3 * k + 2];
    const float gx = w.forces[3 * i], gy = w.forces[3 * i + 1], gz = w.forces[3 * i + 2];
    w.ft[2 * i] = make_float4(gx, gy, gz, 0.f);
    w.ft[2 * i + 1] = make_float4(ry * gz - rz * gy, rz * gx - rx * gz, rx * gy - ry * gx, 0.f);
  }
  wave_sync();
  float f0 = 0.f, f1 = 0.f, f2 = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f, ox = 0.f, oy = 0.f, oz = 0.f;
  int cs = 0, ce = 0, cl = 0;
  if (lane < L.n_nodes) {
    const int k = lane;
    ox = w.origin[3 * k], oy = w.origin[3 * k + 1], oz = w.origin[3 * k + 2];
    const int a_end = L.aend[k];
    for (int i0 = L.abeg[k]; i0 < a_end; i0 += 4) {
      float4 g[4], t[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u < a_end ? i0 + u : i0;
        g[u] = w.ft[2 * i], t[u] = w.ft[2 * i + 1];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool on = i0 + u < a_end;
        f0 += on ? g[u].x : 0.f;
        f1 += on ? g[u].y : 0.f;
        f2 += on ? g[u].z : 0.f;
        t0 += on ? t[u].x : 0.f;
        t1 += on ? t[u].y : 0.f;
        t2 += on ? t[u].z : 0.f;
      }
    }
    cs = L.child_start[k];
    ce = L.child_start[k + 1];
    if (k < L.n_nodes - 1) cl = L.child_list[k];  // entry k of the child list (a tree has n_nodes - 1 edges)
  }
  // 7. fold children into parents, children in increasing order (branches_derivative, tree.h:301-311): the parent
  // adds (f_c, (o_c - o_parent) x f_c + t_c).  Every lane forms that contribution for its own node from its own
  // registers (it is only read once the node is complete: children have larger indices and k runs downwards), so
  // the dependent chain per edge is cross product -> six v_readlane -> six additions in the parent's lane.
  float rx = 0.f, ry = 0.f, rz = 0.f;
  if (lane > 0 && lane < L.n_nodes) {
    const int pr = L.parent[lane];
    if (pr >= 0)  // a residue's first segment (parent -2) hands nothing up (flex.derivative, tree.h:374-393)
      rx = ox - w.origin[3 * pr], ry = oy - w.origin[3 * pr + 1], rz = oz - w.origin[3 * pr + 2];
  }
  for (int k = L.n_nodes - 1; k >= 0; k--) {
    const int s = __builtin_amdgcn_readlane(cs, k), e_end = __builtin_amdgcn_readlane(ce, k);
    for (int e = s; e < e_end; e++) {
      const int c = __builtin_amdgcn_readlane(cl, e);
      const float u0 = (ry * f2 - rz * f1) + t0, u1 = (rz * f0 - rx * f2) + t1, u2 = (rx * f1 - ry * f0) + t2;
      const float c0 = rl(f0, c), c1 = rl(f1, c), c2 = rl(f2, c), c3 = rl(u0, c), c4 = rl(u1, c), c5 = rl(u2, c);
      if (lane == k) {
        f0 += c0;
        f1 += c1;
        f2 += c2;
        t0 += c3;
        t1 += c4;
        t2 += c5;
      }
    }
  }
  if (lane == 0) {
    change[0] = f0, change[1] = f1, change[2] = f2, change[3] = t0, change[4] = t1, change[5] = t2;
  } else if (lane < L.n_nodes) {
    change[6 + (lane - 1)] = t0 * w.axis[3 * lane] + t1 * w.axis[3 * lane + 1] + t2 * w.axis[3 * lane + 2];
  }
}

// MODE 0: model::eval_deriv (model.cu:202-225); MODE 1: model::eval (energy only, midpoint pair table);
// MODE 2: cache::eval (cache.cpp:52-63: receptor-grid term only -- the energy gnina's Metropolis step
// uses, monte_carlo.cpp:44-47); MODE 3: model::set only (coordinates); MODE 4: eval_intramolecular
// (model.cu:352-399: ligand pairs + flexible-residue atoms against the rigid receptor + pairs among flexible / inflex
// atoms, energy only); MODE 5: the pair part of model::eval (other_pairs, then the ligand's -- all pairs).  Returns the
// energy in every lane; MODE 0 writes change[6 + T] to LDS.
template <int MODE, int PG = kPairGroup>
__device__ __forceinline__ float eval_conf(const VinaEnv &env, const VinaLigand &L, const float *conf, float v0, float v1,
                                           float v2, const WaveWork &w, float *change) {
  constexpr bool DERIV = MODE == 0;
  const int lane = threadIdx.x & 63;
  // 1. node frames, sequential down the tree (tree.h:152-156, 218-233).  The frame of the previous
  // node stays in registers: in DFS order the parent is usually the node just computed, so the
  // dependent chain runs register to register and LDS is only written (for the atom / derivative
  // stages), read back only at branch points.
  // (the trigonometry of all torsions is evaluated once, one node per lane: a single wave issues about
  // one instruction every four cycles, so the serial tree walk below is kept as short as possible)
  // Lane k keeps node k's constant description (relative origin / axis, parent) and the cos / sin of its half
  // torsion angle in registers; the serial walk below reads them with v_readlane, so the dependent chain has no
  // LDS round trip except at branch points.  Every lane runs the (wave-uniform) walk; lane 0 stores the frames.
  float ro0 = 0.f, ro1 = 0.f, ro2 = 0.f, ra0 = 0.f, ra1 = 0.f, ra2 = 0.f, cn_r = 1.f, sn_r = 0.f;
  int par_r = -1;
  if (lane < L.n_nodes) {
    const int k = lane;
    ro0 = L.rel_origin[3 * k], ro1 = L.rel_origin[3 * k + 1], ro2 = L.rel_origin[3 * k + 2];
    ra0 = L.rel_axis[3 * k], ra1 = L.rel_axis[3 * k + 1], ra2 = L.rel_axis[3 * k + 2];
    par_r = L.parent[k];
    if (k > 0) {
      const float angle = norm_angle(conf[7 + (k - 1)]);  // angle_to_quaternion, quaternion.h:284-291
      sincos_ref(angle / 2, sn_r, cn_r);
    }
  }
  if (5 * L.n_levels <= 4 * L.n_nodes) {
    // branched tree: one LEVEL per step instead of one node.  Lane k computes node k's frame from its parent's
    // (16 values fetched from the parent's lane with ds_bpermute) when its level comes up -- the same arithmetic per
    // node as the serial walk below, but siblings and cousins advance together and every lane stores its own frame.
    float q[4], M[9], o[3], ax[3] = {0.f, 0.f, 0.f};
    o[0] = conf[0], o[1] = conf[1], o[2] = conf[2];
    q[0] = conf[3], q[1] = conf[4], q[2] = conf[5], q[3] = conf[6];
    quat_to_r3(q, M);
    const int my_depth = lane < L.n_nodes ? L.depth[lane] : -1;
    const int src = par_r >= 0 ? par_r : 0;
    for (int d = 1; d < L.n_levels; d++) {
      float pM[9], pq[4], po[3];
#pragma unroll
      for (int i = 0; i < 9; i++) pM[i] = __shfl(M[i], src);
#pragma unroll
      for (int i = 0; i < 4; i++) pq[i] = __shfl(q[i], src);
#pragma unroll
      for (int i = 0; i < 3; i++) po[i] = __shfl(o[i], src);
      if (par_r == -2) {  // first_segment::set_conf (tree.h:272-277): the world's frame
#pragma unroll
        for (int i = 0; i < 9; i++) pM[i] = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
        pq[0] = 1.f, pq[1] = pq[2] = pq[3] = 0.f;
        po[0] = po[1] = po[2] = 0.f;
      }
      float tx, ty, tz, nax, nay, naz;
      mat_vec(pM, ro0, ro1, ro2, tx, ty, tz);
      const float no0 = po[0] + tx, no1 = po[1] + ty, no2 = po[2] + tz;
      mat_vec(pM, ra0, ra1, ra2, nax, nay, naz);
      float rq[4] = {cn_r, sn_r * nax, sn_r * nay, sn_r * naz}, nq[4], nM[9];
      quat_mul(rq, pq, nq);
      quat_norm_approx(nq);
      quat_to_r3(nq, nM);
      if (my_depth == d) {
#pragma unroll
        for (int i = 0; i < 9; i++) M[i] = nM[i];
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] = nq[i];
        o[0] = no0, o[1] = no1, o[2] = no2;
        ax[0] = nax, ax[1] = nay, ax[2] = naz;
      }
    }
    if (lane < L.n_nodes) {
      const int k = lane;
#pragma unroll
      for (int i = 0; i < 9; i++) w.M[9 * k + i] = M[i];
#pragma unroll
      for (int i = 0; i < 4; i++) w.q[4 * k + i] = q[i];
      w.origin[3 * k] = o[0], w.origin[3 * k + 1] = o[1], w.origin[3 * k + 2] = o[2];
      w.axis[3 * k] = ax[0], w.axis[3 * k + 1] = ax[1], w.axis[3 * k + 2] = ax[2];
    }
  } else
  {
    float q[4], M[9], o[3];
    int prev = -1;
    for (int k = 0; k < L.n_nodes; k++) {
      float ax = 0.f, ay = 0.f, az = 0.f;
      if (k == 0) {
        o[0] = conf[0], o[1] = conf[1], o[2] = conf[2];
        q[0] = conf[3], q[1] = conf[4], q[2] = conf[5], q[3] = conf[6];
      } else {
        const int p = __builtin_amdgcn_readlane(par_r, k);
        if (p == -2) {  // a residue's first segment: its parent is the world
#pragma unroll
          for (int i = 0; i < 9; i++) M[i] = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
          q[0] = 1.f, q[1] = q[2] = q[3] = 0.f;
          o[0] = o[1] = o[2] = 0.f;
        } else if (p != prev) {  // branch point: reload the parent's frame (stored earlier in this walk by lane 0)
#pragma unroll
          for (int i = 0; i < 9; i++) M[i] = w.M[9 * p + i];
#pragma unroll
          for (int i = 0; i < 4; i++) q[i] = w.q[4 * p + i];
#pragma unroll
          for (int i = 0; i < 3; i++) o[i] = w.origin[3 * p + i];
        }
        float tx, ty, tz;
        mat_vec(M, rl(ro0, k), rl(ro1, k), rl(ro2, k), tx, ty, tz);
        o[0] = o[0] + tx;
        o[1] = o[1] + ty;
        o[2] = o[2] + tz;
        mat_vec(M, rl(ra0, k), rl(ra1, k), rl(ra2, k), ax, ay, az);
        const float cn = rl(cn_r, k), sn = rl(sn_r, k);
        float rq[4] = {cn, sn * ax, sn * ay, sn * az}, nq[4];
        quat_mul(rq, q, nq);
        quat_norm_approx(nq);
        q[0] = nq[0], q[1] = nq[1], q[2] = nq[2], q[3] = nq[3];
      }
      quat_to_r3(q, M);
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) w.M[9 * k + i] = M[i];
#pragma unroll
        for (int i = 0; i < 4; i++) w.q[4 * k + i] = q[i];
        w.origin[3 * k] = o[0], w.origin[3 * k + 1] = o[1], w.origin[3 * k + 2] = o[2];
        w.axis[3 * k] = ax, w.axis[3 * k + 1] = ay, w.axis[3 * k + 2] = az;
      }
      prev = k;
    }
  }
  wave_sync();
  // 2. atom coordinates (atom_frame::set_coords, tree.h:128-131) + 3. receptor term.  For the first
  // kTapIt * 64 atoms the grid look-up is split: the corner loads are issued here and interpolated after the
  // pair stage (3b below), so the two L2 round trips overlap.
  constexpr bool GRID = MODE != 3 && MODE != 4 && MODE != 5;
  constexpr int kTapIt = 2;
  float e_part = 0.f;
  // strict mode (seq_add): the receptor term, the other_pairs sum, the ligand-pair sum and model::eval's user-grid
  // term, each accumulated in the reference's order, wave-uniform
  // (eval_intramolecular of a model with flexible residues -- flex4 -- adds the flex-rigid terms and then the flex-flex
  // pairs one by one onto the ligand's pair sum, model.cu:352-399: e_run below)
  const bool strict = env.strict != 0;
  const bool flex4 = MODE == 4 && L.lig_end > L.lig_begin && L.pair_cap != nullptr;
  float s_rec = 0.f, s_p1 = 0.f, s_p0 = 0.f, e_run = 0.f;
  GridTap tap[kTapIt];
  bool tap_on[kTapIt];
  auto place_atom = [&](int i, float &cx, float &cy, float &cz) {
    const int k = L.node_of_atom[i];
    if (k < 0) {  // inflex atom: stays where the file has it
      cx = L.local_xyz[3 * i], cy = L.local_xyz[3 * i + 1], cz = L.local_xyz[3 * i + 2];
    } else {
      float tx, ty, tz;
      mat_vec(w.M + 9 * k, L.local_xyz[3 * i], L.local_xyz[3 * i + 1], L.local_xyz[3 * i + 2], tx, ty, tz);
      cx = w.origin[3 * k] + tx, cy = w.origin[3 * k + 1] + ty, cz = w.origin[3 * k + 2] + tz;
    }
    w.coords[3 * i] = cx;
    w.coords[3 * i + 1] = cy;
    w.coords[3 * i + 2] = cz;
  };
#pragma unroll
  for (int it = 0; it < kTapIt; it++) {
    const int i = lane + 64 * it;
    tap_on[it] = false;
    if (i < L.n_atoms) {
      float cx, cy, cz;
      place_atom(i, cx, cy, cz);
      const int t = L.smt[i];
      if (GRID && !env.direct && !strict && i < L.n_movable && t > 1 && env.grid_off[t] >= 0) {  // hydrogens / types without a grid are skipped (cache.cpp:69-75)
        grid_fetch(env.geom, env.grid_data + env.grid_off[t], cx, cy, cz, env.slope, tap[it]);
        tap_on[it] = true;
      } else if (DERIV) {
        w.forces[3 * i] = 0.f;
        w.forces[3 * i + 1] = 0.f;
        w.forces[3 * i + 2] = 0.f;
      }
    }
  }
  for (int i = lane + 64 * kTapIt; i < L.n_atoms; i += 64) {
    float cx, cy, cz;
    place_atom(i, cx, cy, cz);
    const int t = L.smt[i];
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (GRID && !env.direct && !strict && i < L.n_movable && t > 1 && env.grid_off[t] >= 0)
      e_part += grid_evaluate<DERIV>(env.geom, env.grid_data + env.grid_off[t], cx, cy, cz, env.slope, v1, fx, fy, fz);
    if (DERIV) {
      w.forces[3 * i] = fx;
      w.forces[3 * i + 1] = fy;
      w.forces[3 * i + 2] = fz;
    }
  }
  wave_sync();
  if (GRID && !env.direct && strict) {  // cache::eval / eval_deriv with e summed in atom order (cache.cpp:52-83)
    for (int base = 0; base < L.n_movable; base += 64) {
      const int i = base + lane;
      float eg = 0.f;
      bool on = false;
      if (i < L.n_movable) {
        const int t = L.smt[i];
        if (t > 1 && env.grid_off[t] >= 0) {
          float fx, fy, fz;
          eg = grid_evaluate<DERIV>(env.geom, env.grid_data + env.grid_off[t], w.coords[3 * i], w.coords[3 * i + 1],
                                    w.coords[3 * i + 2], env.slope, v1, fx, fy, fz);
          on = true;
          if (DERIV) w.forces[3 * i] = fx, w.forces[3 * i + 1] = fy, w.forces[3 * i + 2] = fz;
        }
      }
      seq_add(s_rec, eg, __builtin_amdgcn_ballot_w64(on));
    }
    wave_sync();
  }
  if (MODE != 3 && MODE != 4 && MODE != 5 && env.direct) {
    // non_cache::eval / eval_deriv (non_cache.cpp:52-83,125-179): every ligand heavy atom against every
    // receptor atom within the cutoff -- the 64 lanes stride over the receptor, one ligand atom at a time
    for (int i = 0; i < L.n_movable; i++) {
      const int t1 = L.smt[i];
      if (t1 <= 1) continue;
      float adj[3], oobd[3] = {0.f, 0.f, 0.f}, oob = 0.f;
#pragma unroll
      for (int k = 0; k < 3; k++) {  // check_bounds(_deriv), non_cache.cpp:32-50,102-123
        const float c = w.coords[3 * i + k];
        adj[k] = c;
        if (c < env.box_begin[k]) {
          adj[k] = env.box_begin[k];
          oobd[k] = -1.f;
          oob += fabsf(c - env.box_begin[k]);
        } else if (c > env.box_end[k]) {
          adj[k] = env.box_end[k];
          oobd[k] = 1.f;
          oob += fabsf(c - env.box_end[k]);
        }
      }
      oob *= env.slope;
      float pe = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
      if (strict) {  // receptor atoms in index order (non_cache.cpp:60-75,140-160)
        for (int jb = 0; jb < env.n_rec; jb += 64) {
          const int j = jb + lane;
          float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
          bool on = false;
          if (j < env.n_rec) {
            const float4 r = env.rec[j];
            const float rx = adj[0] - r.x, ry = adj[1] - r.y, rz = adj[2] - r.z;
            const float r2 = rx * rx + ry * ry + rz * rz;
            if (r2 < env.cutoff_sqr) {
              on = true;
              const int t2 = __float_as_int(r.w);
              if (DERIV) {
                float dor;
                prec_eval_deriv(env, t1, t2, r2, c0, dor);
                c1 = dor * rx, c2 = dor * ry, c3 = dor * rz;
              } else {
                c0 = prec_eval(env, t1, t2, r2);
              }
            }
          }
          const unsigned long long m = __builtin_amdgcn_ballot_w64(on);
          seq_add(pe, c0, m);
          if (DERIV) {
            seq_add(dx, c1, m);
            seq_add(dy, c2, m);
            seq_add(dz, c3, m);
          }
        }
      } else {
      for (int j = lane; j < env.n_rec; j += 64) {
        const float4 r = env.rec[j];
        const float rx = adj[0] - r.x, ry = adj[1] - r.y, rz = adj[2] - r.z;
        const float r2 = rx * rx + ry * ry + rz * rz;
        if (r2 < env.cutoff_sqr) {
          const int t2 = __float_as_int(r.w);
          if (DERIV) {
            float e1, dor;
            prec_eval_deriv(env, t1, t2, r2, e1, dor);
            pe += e1;
            dx += dor * rx;
            dy += dor * ry;
            dz += dor * rz;
          } else {
            pe += prec_eval(env, t1, t2, r2);
          }
        }
      }
      pe = wave_sum(pe);
      }
      if (DERIV) {
        if (!strict) {
          dx = wave_sum(dx);
          dy = wave_sum(dy);
          dz = wave_sum(dz);
        }
        if (env.ug_data) {  // user grid at the atom's own coordinates, before the curl (non_cache.cpp:168-173)
          float ux, uy, uz;
          pe += grid_evaluate<true>(env.ug_geom, env.ug_data, w.coords[3 * i], w.coords[3 * i + 1], w.coords[3 * i + 2],
                                    env.slope, 1000.f, ux, uy, uz);  // grid::evaluate_user, grid.cpp:47-49
          dx += ux, dy += uy, dz += uz;
        }
        curl3(pe, dx, dy, dz, v1);
        if (lane == 0) {
          w.forces[3 * i] = dx + env.slope * oobd[0];
          w.forces[3 * i + 1] = dy + env.slope * oobd[1];
          w.forces[3 * i + 2] = dz + env.slope * oobd[2];
        }
      } else {
        curl1(pe, v1);
      }
      if (strict)
        s_rec += pe + oob;
      else if (lane == 0)
        e_part += pe + oob;
    }
    wave_sync();
  }
  // 4. intramolecular pairs (model.cu:38-60 / :22-36).  Table mode: a lane takes its pairs PG at a time
  // (six: one group covers 384 pairs, a typical drug-like ligand) and issues their table look-ups (L2 latency) together; out-of-cutoff pairs read entry 0 and are discarded, so
  // the group is straight-line code.  Per lane the energies still add up in increasing pair order.
  // eval_intramolecular keeps, of other_pairs, those with neither atom in the ligand (model.cu:386-397)
  const bool has_flex = L.lig_end > L.lig_begin && L.pair_cap != nullptr;
  auto pair_counts = [&](int p, const int2 &ab) -> bool {
    if (MODE != 4 || !has_flex || !L.pair_cap[p]) return true;
    const bool a_lig = ab.x >= L.lig_begin && ab.x < L.lig_end, b_lig = ab.y >= L.lig_begin && ab.y < L.lig_end;
    return !a_lig && !b_lig;
  };
  if ((MODE < 2 || MODE == 4 || MODE == 5) && !env.exact && !env.spline) {
    for (int p0 = lane; p0 < L.n_pairs; p0 += 64 * PG) {
      float rx[PG], ry[PG], rz[PG], rem[PG], capv[PG];
      float2 s1[PG], s2[PG];
      float fastv[PG];
      int2 sl[PG];
      bool in[PG];
      unsigned other = 0;  // bit u: pair u of this lane belongs to model::other_pairs
#pragma unroll
      for (int u = 0; u < PG; u++) {
        const int p = p0 + 64 * u;
        const bool valid = p < L.n_pairs;
        const int2 ab = L.pairs[valid ? p : 0];
        const bool oth = L.pair_cap && L.pair_cap[valid ? p : 0];
        other |= oth ? 1u << u : 0u;
        capv[u] = oth ? v2 : v0;
        if (DERIV) sl[u] = L.pair_slots[valid ? p : 0];
        rx[u] = w.coords[3 * ab.y] - w.coords[3 * ab.x];
        ry[u] = w.coords[3 * ab.y + 1] - w.coords[3 * ab.x + 1];
        rz[u] = w.coords[3 * ab.y + 2] - w.coords[3 * ab.x + 2];
        const float r2 = rx[u] * rx[u] + ry[u] * ry[u] + rz[u] * rz[u];
        in[u] = valid && r2 < env.cutoff_sqr && pair_counts(valid ? p : 0, ab);
        const long base = (long)tri_idx(L.smt[ab.x], L.smt[ab.y]) * env.n;
        const float r2f = env.factor * r2;
        const int i1 = (int)r2f;
        rem[u] = r2f - (float)i1;
        const long at = in[u] ? base + i1 : 0;
        if (DERIV) {
          s1[u] = env.smooth[at];
          s2[u] = env.smooth[at + 1];
        } else {
          fastv[u] = env.fast[at];
        }
      }
#pragma unroll
      for (int u = 0; u < PG; u++) {
        const int p = p0 + 64 * u;
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        float pe = 0.f;
        if (in[u]) {
          if (DERIV) {
            pe = s1[u].x + rem[u] * (s2[u].x - s1[u].x);
            const float dor = s1[u].y + rem[u] * (s2[u].y - s1[u].y);
            float fx = dor * rx[u], fy = dor * ry[u], fz = dor * rz[u];
            curl3(pe, fx, fy, fz, capv[u]);
            out = make_float4(fx, fy, fz, pe);
          } else {
            pe = fastv[u];
            curl1(pe, capv[u]);
          }
        }
        if (strict) {  // pairs in list order, other_pairs and the ligand's own summed apart (model.cu:206-216)
          const bool oth = (other >> u) & 1;
          if (L.pair_cap && !flex4) seq_add(s_p1, pe, __builtin_amdgcn_ballot_w64(in[u] && oth));
          seq_add(s_p0, pe, __builtin_amdgcn_ballot_w64(in[u] && !oth));
        } else {
          e_part += pe;
        }
        if (DERIV && p < L.n_pairs) {  // forces[a] -= f; forces[b] += f, as entries of the two atoms' lists
          w.cx[sl[u].x] = -out.x, w.cy[sl[u].x] = -out.y, w.cz[sl[u].x] = -out.z;
          w.cx[sl[u].y] = out.x, w.cy[sl[u].y] = out.y, w.cz[sl[u].y] = out.z;
        }
      }
    }
  }
  for (int pb = 0; (MODE < 2 || MODE == 4 || MODE == 5) && (env.exact || env.spline) && pb < L.n_pairs; pb += 64) {
    const int p = pb + lane;
    const bool valid = p < L.n_pairs;
    const int2 ab = L.pairs[valid ? p : 0];
    const float rx = w.coords[3 * ab.y] - w.coords[3 * ab.x], ry = w.coords[3 * ab.y + 1] - w.coords[3 * ab.x + 1],
                rz = w.coords[3 * ab.y + 2] - w.coords[3 * ab.x + 2];
    const float r2 = rx * rx + ry * ry + rz * rz;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool oth = L.pair_cap && L.pair_cap[valid ? p : 0];
    const float capx = oth ? v2 : v0;
    const bool in = valid && r2 < env.cutoff_sqr && pair_counts(valid ? p : 0, ab);
    float pe = 0.f;
    if (in) {
      if (DERIV) {
        float dor;
        prec_eval_deriv(env, L.smt[ab.x], L.smt[ab.y], r2, pe, dor);
        float fx = dor * rx, fy = dor * ry, fz = dor * rz;
        curl3(pe, fx, fy, fz, capx);
        out = make_float4(fx, fy, fz, pe);
      } else {
        pe = prec_eval(env, L.smt[ab.x], L.smt[ab.y], r2);
        curl1(pe, capx);
      }
    }
    if (strict) {
      if (L.pair_cap && !flex4) seq_add(s_p1, pe, __builtin_amdgcn_ballot_w64(in && oth));
      seq_add(s_p0, pe, __builtin_amdgcn_ballot_w64(in && !oth));
    } else {
      e_part += pe;
    }
    if (DERIV && valid) {
      const int2 sl = L.pair_slots[p];
      w.cx[sl.x] = -out.x, w.cy[sl.x] = -out.y, w.cz[sl.x] = -out.z;
      w.cx[sl.y] = out.x, w.cy[sl.y] = out.y, w.cz[sl.y] = out.z;
    }
  }
  if (MODE == 4 && has_flex && strict) {
    // strict order: e = 0 + the ligand's pair sum; then every flex-rigid term in (atom, receptor atom) order; then every
    // flex-flex pair of other_pairs in list order -- one running sum, as model.cu:352-399 adds them
    e_run = 0.f + rl(s_p0, 0);
    for (int i = 0; env.rec && i < L.n_movable; i++) {
      if (i >= L.lig_begin && i < L.lig_end) continue;
      const int t1 = L.smt[i];
      if (t1 <= 1) continue;
      const float ax = w.coords[3 * i], ay = w.coords[3 * i + 1], az = w.coords[3 * i + 2];
      for (int j0 = 0; j0 < env.n_rec; j0 += 64) {
        const int j = j0 + lane;
        const float4 r = env.rec[j < env.n_rec ? j : 0];
        const float rx = ax - r.x, ry = ay - r.y, rz = az - r.z;
        const float r2 = rx * rx + ry * ry + rz * rz;
        const bool in = j < env.n_rec && r2 < env.cutoff_sqr;
        float pe = 0.f;
        if (in) {
          pe = prec_eval(env, t1, __float_as_int(r.w), r2);
          curl1(pe, v1);
        }
        seq_add(e_run, pe, __builtin_amdgcn_ballot_w64(in));
      }
    }
    for (int pb = 0; pb < L.n_pairs; pb += 64) {
      const int p = pb + lane;
      const bool valid = p < L.n_pairs;
      const int2 ab = L.pairs[valid ? p : 0];
      const float rx = w.coords[3 * ab.y] - w.coords[3 * ab.x], ry = w.coords[3 * ab.y + 1] - w.coords[3 * ab.x + 1],
                  rz = w.coords[3 * ab.y + 2] - w.coords[3 * ab.x + 2];
      const float r2 = rx * rx + ry * ry + rz * rz;
      const bool in = valid && L.pair_cap[valid ? p : 0] && r2 < env.cutoff_sqr && pair_counts(valid ? p : 0, ab);
      float pe = 0.f;
      if (in) {
        if (env.exact || env.spline) {
          pe = prec_eval(env, L.smt[ab.x], L.smt[ab.y], r2);
        } else {
          const long base = (long)tri_idx(L.smt[ab.x], L.smt[ab.y]) * env.n;
          pe = env.fast[base + (int)(env.factor * r2)];
        }
        curl1(pe, v2);
      }
      seq_add(e_run, pe, __builtin_amdgcn_ballot_w64(in));
    }
  } else if (MODE == 4 && has_flex && env.rec) {
    // eval_intramolecular's flex-rigid term (model.cu:364-384): every heavy movable atom outside the ligand against
    // every heavy atom of the rigid receptor, each PAIR curled with v[1] (the igrid curls an atom's sum instead)
    for (int i = 0; i < L.n_movable; i++) {
      if (i >= L.lig_begin && i < L.lig_end) continue;
      const int t1 = L.smt[i];
      if (t1 <= 1) continue;
      const float ax = w.coords[3 * i], ay = w.coords[3 * i + 1], az = w.coords[3 * i + 2];
      for (int j = lane; j < env.n_rec; j += 64) {
        const float4 r = env.rec[j];
        const float rx = ax - r.x, ry = ay - r.y, rz = az - r.z;
        const float r2 = rx * rx + ry * ry + rz * rz;
        if (r2 < env.cutoff_sqr) {
          float pe = prec_eval(env, t1, __float_as_int(r.w), r2);
          curl1(pe, v1);
          e_part += pe;
        }
      }
    }
  }
  // 3b. interpolate the receptor grids whose corner loads were issued in stage 3
#pragma unroll
  for (int it = 0; it < kTapIt; it++) {
    if (tap_on[it]) {
      const int i = lane + 64 * it;
      float fx = 0.f, fy = 0.f, fz = 0.f;
      e_part += grid_finish<DERIV>(env.geom, tap[it], env.slope, v1, fx, fy, fz);
      if (DERIV) {
        w.forces[3 * i] = fx;
        w.forces[3 * i + 1] = fy;
        w.forces[3 * i + 2] = fz;
      }
    }
  }
  if (DERIV) {
    wave_sync();
    // 5. gather pair forces per atom, in pair order (forces[a] -= f; forces[b] += f)
    for (int i = lane; i < L.n_atoms; i += 64) {
      float fx = w.forces[3 * i], fy = w.forces[3 * i + 1], fz = w.forces[3 * i + 2];
      // the atom's list of contributions (pair order, zero padded to a multiple of four) is contiguous: four
      // 16-byte reads per LDS latency, no index indirection; the additions keep the pair order
      const int e_end = L.slot_start[i + 1];
      for (int e = L.slot_start[i]; e < e_end; e += 4) {
        const float4 x4 = *reinterpret_cast<const float4 *>(w.cx + e), y4 = *reinterpret_cast<const float4 *>(w.cy + e),
                     z4 = *reinterpret_cast<const float4 *>(w.cz + e);
        fx += x4.x, fy += y4.x, fz += z4.x;
        fx += x4.y, fy += y4.y, fz += z4.y;
        fx += x4.z, fy += y4.z, fz += z4.z;
        fx += x4.w, fy += y4.w, fz += z4.w;
      }
      w.forces[3 * i] = fx;
      w.forces[3 * i + 1] = fy;
      w.forces[3 * i + 2] = fz;
    }
    wave_sync();
    fold_forces(L, w, change);
  }
  float e_strict = 0.f;
  if (strict) {
    // compose the sums the way the reference's callers do (the pair loop above is not wave-uniform: lane 0 stays in
    // it longest and holds the complete sums)
    s_p1 = rl(s_p1, 0), s_p0 = rl(s_p0, 0);
    if (MODE == 0) e_strict = s_rec + ((0.f + s_p1) + s_p0);  // model::eval_deriv: e = ig.eval_deriv; ie = ...; e += ie
    else if (MODE == 1) e_strict = (s_rec + s_p1) + s_p0;      // model::evale adds other_pairs, model::eval the ligand's
    else if (MODE == 2) e_strict = s_rec;
    else if (flex4) e_strict = e_run;
    else e_strict = (0.f + s_p1) + s_p0;
  }
  if ((MODE == 1 || (MODE == 2 && env.ug_model)) && env.ug_data) {
    // model::eval's own user-grid term (model.cu:125-134): every atom of the ligand, hydrogens included, at slope
    // 1000 -- this (not the igrid) is how --user_grid reaches eval_adjusted and the final energies
    const int lb = L.lig_end > L.lig_begin ? L.lig_begin : 0, le = L.lig_end > L.lig_begin ? L.lig_end : L.n_atoms;
    for (int base = lb; base < le; base += 64) {
      const int i = base + lane;
      float ux, uy, uz, t = 0.f;
      if (i < le)
        t = grid_evaluate<false>(env.ug_geom, env.ug_data, w.coords[3 * i], w.coords[3 * i + 1], w.coords[3 * i + 2],
                                 1000.f, 1000.f, ux, uy, uz);
      if (strict)
        seq_add(e_strict, t, __builtin_amdgcn_ballot_w64(i < le));
      else
        e_part += t;
    }
  }
  wave_sync();
  if (strict) return e_strict;
  return wave_sum(e_part);
}

// ---------------------------------------------------------------------------------------------
// batch evaluation kernel
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(64) void vina_eval_kernel(VinaEnv env, VinaLigand L, const float *confs, float v0,
                                                       float v1, float v2, float *energy, float *change_out,
                                                       float *coords_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *p = lds;
  if (env.ligs) L = env.ligs[env.item_lig[blockIdx.x]];
  WaveWork w = carve_work(p, L);
  const int nt = L.n_nodes - 1, n = 6 + nt, nc = 7 + nt;
  const size_t cs = env.ligs ? env.conf_stride : nc, gs = env.ligs ? env.change_stride : n,
               xs = env.ligs ? env.coord_stride : 3 * L.n_atoms;
  float *conf = carve(p, nc);
  float *change = carve(p, n);
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < nc; i += 64) conf[i] = confs[b * cs + i];
  wave_sync();
  const float e = eval_conf<MODE>(env, L, conf, v0, v1, v2, w, change);
  if (lane == 0) energy[b] = e;
  if (MODE == 0 && change_out)
    for (int i = lane; i < n; i += 64) change_out[b * gs + i] = change[i];
  if (coords_out)
    for (int i = lane; i < 3 * L.n_atoms; i += 64) coords_out[b * xs + i] = w.coords[i];
}

// ---------------------------------------------------------------------------------------------
// non_cache_cnn as the igrid (non_cache_cnn.cpp:33-54,79-169): the receptor term is the CNN loss, computed
// by the CNN engine for the coordinates of launch A; launch B adds the out-of-box penalties of the search
// box `gd` and of the CNN cube `cnn_gd` to the CNN's per-atom gradient (hydrogens: zero force) and folds
// the forces into change[6 + T].  skip_interacting_pairs() is true for this igrid: no intramolecular term.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void vina_coords_kernel(VinaEnv env, VinaLigand L, const float *confs,
                                                         float *coords_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *p = lds;
  WaveWork w = carve_work(p, L);
  const int nc = 7 + L.n_nodes - 1;
  float *conf = carve(p, nc);
  float *change = carve(p, 6 + L.n_nodes - 1);
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < nc; i += 64) conf[i] = confs[(size_t)b * nc + i];
  wave_sync();
  eval_conf<3>(env, L, conf, 0.f, 0.f, 0.f, w, change);
  for (int i = lane; i < 3 * L.n_atoms; i += 64) coords_out[(size_t)b * 3 * L.n_atoms + i] = w.coords[i];
}

__global__ __launch_bounds__(64) void vina_extforce_kernel(VinaEnv env, VinaLigand L, const float *confs,
                                                           VinaExtArgs a, float *energy, float *change_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *p = lds;
  WaveWork w = carve_work(p, L);
  const int nt = L.n_nodes - 1, n = 6 + nt, nc = 7 + nt;
  float *conf = carve(p, nc);
  float *change = carve(p, n);
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < nc; i += 64) conf[i] = confs[(size_t)b * nc + i];
  wave_sync();
  eval_conf<3>(env, L, conf, 0.f, 0.f, 0.f, w, change);
  float pen = 0.f;
  int heavy_before = 0;  // non-hydrogen movable atoms in front of this 64-atom block (add_minus_forces' counter j)
  for (int base = 0; base < L.n_atoms; base += 64) {
    const int i = base + lane;
    if (i >= L.n_atoms) continue;  // (only in the last block: the ballot below then counts the live lanes, all it needs)
    float fx = 0.f, fy = 0.f, fz = 0.f;
    const bool heavy = L.smt[i] > 1 && i < L.n_movable;  // (inflex atoms: no minus_forces entry)
    const unsigned long long hm = __builtin_amdgcn_ballot_w64(heavy);
    const int rank = heavy_before + __builtin_popcountll(hm & ((1ull << lane) - 1ull));
    heavy_before += __builtin_popcountll(hm);
    if (heavy) {  // hydrogens: minus_forces = 0, no penalty (non_cache_cnn.cpp:92-96)
      float f[3] = {0.f, 0.f, 0.f}, dist = 0.f;
      if (a.forces) {
        // model::add_minus_forces (model.cu:247-259): the k-th non-hydrogen atom takes gradient entry k
        const float *g = a.forces + ((size_t)b * L.n_atoms + (a.per_atom_forces ? i : rank)) * 3;
        f[0] = g[0], f[1] = g[1], f[2] = g[2];
      }
#pragma unroll
      for (int k = 0; k < 3; k++) {  // check_bounds_deriv on gd, then on cnn_gd (non_cache.cpp:102-123)
        const float c = w.coords[3 * i + k];
        if (a.use_box) {
          if (c < a.box_begin[k]) {
            f[k] += -a.slope;
            dist += fabsf(c - a.box_begin[k]);
          } else if (c > a.box_end[k]) {
            f[k] += a.slope;
            dist += fabsf(c - a.box_end[k]);
          }
        }
        if (a.cnn_center) {
          const float lo = a.cnn_center[3 * b + k] - a.cnn_half, hi = a.cnn_center[3 * b + k] + a.cnn_half;
          if (c < lo) {
            f[k] += -a.slope;
            dist += fabsf(c - lo);
          } else if (c > hi) {
            f[k] += a.slope;
            dist += fabsf(c - hi);
          }
        }
      }
      pen += dist * a.slope;
      if (env.ug_data && a.forces) {  // this_e / deriv of non_cache_cnn.cpp:141-151: the user grid term, curled on its own
        float ux, uy, uz;
        float uge = grid_evaluate<true>(env.ug_geom, env.ug_data, w.coords[3 * i], w.coords[3 * i + 1], w.coords[3 * i + 2],
                                        a.slope, 1000.f, ux, uy, uz);
        curl3(uge, ux, uy, uz, a.v);
        pen += uge;
        f[0] += ux, f[1] += uy, f[2] += uz;
      }
      fx = f[0], fy = f[1], fz = f[2];
    }
    w.forces[3 * i] = fx;
    w.forces[3 * i + 1] = fy;
    w.forces[3 * i + 2] = fz;
  }
  wave_sync();
  float emp_total = 0.f;
  if (a.mix_force) {
    // mix_emp_force (non_cache_cnn.cpp:113-137,151-156): the empirical receptor term of every heavy atom, taken at
    // its coordinates clamped to the search box, curl-capped, is blended into the CNN force:
    //   minus_forces = (cnn + penalties + w (emp_deriv + search-box penalty force)) / (1 + w)
    // The 64 lanes stride the receptor for one ligand atom at a time (like the non_cache igrid).
    for (int i = 0; i < L.n_movable; i++) {
      const int t1 = L.smt[i];
      if (t1 <= 1) continue;
      float adj[3], oobd[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float c = w.coords[3 * i + k];
        adj[k] = c;
        if (a.use_box) {
          if (c < a.box_begin[k]) adj[k] = a.box_begin[k], oobd[k] = -1.f;
          else if (c > a.box_end[k]) adj[k] = a.box_end[k], oobd[k] = 1.f;
        }
      }
      float pe = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
      for (int j = lane; j < env.n_rec; j += 64) {
        const float4 r = env.rec[j];
        const float rx = adj[0] - r.x, ry = adj[1] - r.y, rz = adj[2] - r.z;
        const float r2 = rx * rx + ry * ry + rz * rz;
        if (r2 < env.cutoff_sqr) {
          float e1, dor;
          prec_eval_deriv(env, t1, __float_as_int(r.w), r2, e1, dor);
          pe += e1;
          dx += dor * rx;
          dy += dor * ry;
          dz += dor * rz;
        }
      }
      pe = wave_sum(pe);
      dx = wave_sum(dx);
      dy = wave_sum(dy);
      dz = wave_sum(dz);
      if (env.ug_data) {  // emp_e += uge; emp_deriv += ug_deriv (non_cache_cnn.cpp:146-149)
        float ux, uy, uz;
        pe += grid_evaluate<true>(env.ug_geom, env.ug_data, w.coords[3 * i], w.coords[3 * i + 1], w.coords[3 * i + 2],
                                  a.slope, 1000.f, ux, uy, uz);
        dx += ux, dy += uy, dz += uz;
      }
      curl3(pe, dx, dy, dz, a.v);
      if (lane == 0) {
        const float den = 1.0f + a.weight;
        w.forces[3 * i] = (w.forces[3 * i] + a.weight * (dx + a.slope * oobd[0])) / den;
        w.forces[3 * i + 1] = (w.forces[3 * i + 1] + a.weight * (dy + a.slope * oobd[1])) / den;
        w.forces[3 * i + 2] = (w.forces[3 * i + 2] + a.weight * (dz + a.slope * oobd[2])) / den;
        emp_total += pe;
      }
    }
    wave_sync();
  }
  if (change_out) {
    fold_forces(L, w, change);
    wave_sync();
    for (int i = lane; i < n; i += 64) change_out[(size_t)b * n + i] = change[i];
  }
  pen = wave_sum(pen);
  if (lane == 0) {
    float e = (a.e_in ? a.e_in[b] : 0.f) + pen;
    if (a.mix_energy) e = (e + a.weight * emp_total) / (1.0f + a.weight);  // non_cache_cnn.cpp:160-166
    energy[b] = e;
  }
}

// ---------------------------------------------------------------------------------------------
// cache::eval / cache::eval_deriv (cache.cpp:50-83) on given coordinates: the igrid seam itself (igrid.h:32-46) for
// callers that hold a `model` with coordinates rather than a conformation.  One wavefront per coordinate set;
// per-atom energies are added in atom order like the reference's loop.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void vina_cache_coords_kernel(VinaEnv env, const float *coords, const int *smt, int n_atoms,
                                                               float v, float *energy, float *minus_forces) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const float *xyz = coords + (size_t)b * n_atoms * 3;
  for (int i = lane; i < n_atoms; i += 64) {
    const int t = smt[i];
    float e = 0.f, fx = 0.f, fy = 0.f, fz = 0.f;
    if (t > 1 && t < kVinaTypes && env.grid_off[t] >= 0) {
      const float *g = env.grid_data + env.grid_off[t];
      if (minus_forces)
        e = grid_evaluate<true>(env.geom, g, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], env.slope, v, fx, fy, fz);
      else
        e = grid_evaluate<false>(env.geom, g, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], env.slope, v, fx, fy, fz);
    }
    lds[i] = e;
    if (minus_forces) {
      float *o = minus_forces + ((size_t)b * n_atoms + i) * 3;
      o[0] = fx, o[1] = fy, o[2] = fz;
    }
  }
  wave_sync();
  if (lane == 0) {
    float e = 0.f;
    for (int i = 0; i < n_atoms; i++) {
      const int t = smt[i];
      if (t > 1 && t < kVinaTypes && env.grid_off[t] >= 0) e += lds[i];
    }
    energy[b] = e;
  }
}

void launch_vina_cache_coords(const VinaEnv &env, const float *coords, const int *smt, int n_atoms, int B, float v,
                              float *energy, float *minus_forces, hipStream_t s) {
  hipLaunchKernelGGL(vina_cache_coords_kernel, dim3(B), dim3(64), (size_t)n_atoms * sizeof(float), s, env, coords, smt,
                     n_atoms, v, energy, minus_forces);
}

void launch_vina_coords(const VinaEnv &env, const VinaLigand &lig, const float *confs, int B, float *coords,
                        hipStream_t s) {
  const size_t lds = vina_wave_lds_bytes(lig.n_atoms, lig.n_nodes, lig.n_pairs, false, false);
  big_lds(vina_coords_kernel);
  hipLaunchKernelGGL(vina_coords_kernel, dim3(B), dim3(64), lds, s, env, lig, confs, coords);
}

void launch_vina_extforce(const VinaEnv &env, const VinaLigand &lig, const float *confs, int B, const VinaExtArgs &a,
                          float *energy, float *change, hipStream_t s) {
  const size_t lds = vina_wave_lds_bytes(lig.n_atoms, lig.n_nodes, lig.n_pairs, false, false);
  big_lds(vina_extforce_kernel);
  hipLaunchKernelGGL(vina_extforce_kernel, dim3(B), dim3(64), lds, s, env, lig, confs, a, energy, change);
}

// latency probe (tools/bench_vina.py): `reps` back-to-back evaluations of one conformation per wave
template <int MODE>
__global__ __launch_bounds__(64) void vina_eval_repeat_kernel(VinaEnv env, VinaLigand L, const float *confs, float v0,
                                                              float v1, float v2, int reps, float *energy) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *p = lds;
  L = stage_ligand(L, p);
  WaveWork w = carve_work(p, L);
  const int nt = L.n_nodes - 1, n = 6 + nt, nc = 7 + nt;
  float *conf = carve(p, nc);
  float *change = carve(p, n);
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < nc; i += 64) conf[i] = confs[(size_t)b * nc + i];
  wave_sync();
  float e = 0.f;
  for (int r = 0; r < reps; r++) {
    e += eval_conf<MODE>(env, L, conf, v0, v1, v2, w, change);
    if (lane == 0) conf[0] += 1e-7f * e;  // keep the iterations dependent
    wave_sync();
  }
  if (lane == 0) energy[b] = e;
}

void launch_vina_eval_repeat(const VinaEnv &env0, const VinaLigand &lig, const float *confs, int B, int mode, int reps,
                             float *energy, hipStream_t s) {
  VinaEnv env = env0;
  env.stage = 1;
  const size_t lds = vina_wave_lds_bytes(lig.n_atoms, lig.n_nodes, lig.n_pairs, false, true);
  switch (mode) {
    case 0: big_lds(vina_eval_repeat_kernel<0>); hipLaunchKernelGGL(vina_eval_repeat_kernel<0>, dim3(B), dim3(64), lds, s, env, lig, confs, 10.f, 10.f, 10.f, reps, energy); break;
    case 1: big_lds(vina_eval_repeat_kernel<1>); hipLaunchKernelGGL(vina_eval_repeat_kernel<1>, dim3(B), dim3(64), lds, s, env, lig, confs, 10.f, 10.f, 10.f, reps, energy); break;
    case 2: big_lds(vina_eval_repeat_kernel<2>); hipLaunchKernelGGL(vina_eval_repeat_kernel<2>, dim3(B), dim3(64), lds, s, env, lig, confs, 10.f, 10.f, 10.f, reps, energy); break;
    case 3: big_lds(vina_eval_repeat_kernel<3>); hipLaunchKernelGGL(vina_eval_repeat_kernel<3>, dim3(B), dim3(64), lds, s, env, lig, confs, 10.f, 10.f, 10.f, reps, energy); break;
    default: big_lds(vina_eval_repeat_kernel<4>); hipLaunchKernelGGL(vina_eval_repeat_kernel<4>, dim3(B), dim3(64), lds, s, env, lig, confs, 10.f, 10.f, 10.f, reps, energy); break;
  }
}

// ---------------------------------------------------------------------------------------------
// BFGS kernel: quasi_newton / bfgs<> with fast_line_search (bfgs.h:73-91,357-502)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int hidx(int i, int j) { return i <= j ? i + j * (j + 1) / 2 : j + i * (i + 1) / 2; }

// conf::increment (conf.h:54-59,113-118; quaternion.cu:32-62,96-100), executed by one lane
__device__ void conf_increment(float *x, const float *p, float alpha, int nt) {
  x[0] += alpha * p[0];
  x[1] += alpha * p[1];
  x[2] += alpha * p[2];
  const float rx = alpha * p[3], ry = alpha * p[4], rz = alpha * p[5];
  const float angle = sqrtf(rx * rx + ry * ry + rz * rz);
  float rq[4] = {1.f, 0.f, 0.f, 0.f};
  if (angle > VEPS) {
    const float inv = 1 / angle;
    angle_to_quat(inv * rx, inv * ry, inv * rz, angle, rq);
  }
  float nq[4];
  quat_mul(rq, x + 3, nq);
  quat_norm_approx(nq);
  x[3] = nq[0], x[4] = nq[1], x[5] = nq[2], x[6] = nq[3];
  for (int i = 0; i < nt; i++) {
    float t = x[7 + i] + norm_angle(alpha * p[6 + i]);
    x[7 + i] = norm_angle(t);
  }
}

struct BfgsWork {
  float *x_new, *g_new;  // conformation handed to eval_conf and the change it fills
  float *x, *g, *h;      // in/out conformation, final gradient, triangular inverse-Hessian estimate
};

__device__ __forceinline__ BfgsWork carve_bfgs(float *&pp, int n, int nc) {
  BfgsWork k;
  k.x_new = carve(pp, nc);
  k.g_new = carve(pp, n);
  k.x = carve(pp, nc);
  k.g = carve(pp, n);
  k.h = carve(pp, n * (n + 1) / 2);
  return k;
}

// Sequential dot product of two vectors held one element per lane (scalar_product, bfgs.h:45-50): the products
// are formed in parallel, the additions run in index order through v_readlane -- the same roundings as the
// reference's loop (the build has FMA contraction off), without an LDS round trip per element.
__device__ __forceinline__ float dot_lanes(float a, float b, int n) {
  const float prod = a * b;
  float t = 0.f;
  for (int i = 0; i < n; i++) t += rl(prod, i);
  return t;
}

// -(H v)[lane] with H triangular in LDS and v one element per lane (minus_mat_vec_product, bfgs.h:34-43):
// j ascending; four H entries are fetched per LDS latency.
__device__ __forceinline__ float minus_h_times(const float *h, float v, int n, int row) {
  float sum = 0.f;
  int j = 0;
  for (; j + 4 <= n; j += 4) {
    const float h0 = h[hidx(row, j)], h1 = h[hidx(row, j + 1)], h2 = h[hidx(row, j + 2)], h3 = h[hidx(row, j + 3)];
    sum += h0 * rl(v, j);
    sum += h1 * rl(v, j + 1);
    sum += h2 * rl(v, j + 2);
    sum += h3 * rl(v, j + 3);
  }
  for (; j < n; j++) sum += h[hidx(row, j)] * rl(v, j);
  return -sum;
}

// conf::increment (conf.h:54-59,113-118) with the conformation one element per lane: lanes 0-2 position, 3-6
// orientation (every lane forms the same quaternion and keeps its component), 7.. torsions.  p_r holds the
// direction one element per lane, p_up the same shifted up by one lane (torsion i: x[7+i], p[6+i]).
__device__ __forceinline__ float increment_lanes(float x_r, float p_r, float p_up, float alpha, int nt, int lane) {
  const float rx = alpha * rl(p_r, 3), ry = alpha * rl(p_r, 4), rz = alpha * rl(p_r, 5);
  const float angle = sqrtf(rx * rx + ry * ry + rz * rz);
  float rq[4] = {1.f, 0.f, 0.f, 0.f};
  if (angle > VEPS) {
    const float inv = 1 / angle;
    angle_to_quat(inv * rx, inv * ry, inv * rz, angle, rq);
  }
  const float xq[4] = {rl(x_r, 3), rl(x_r, 4), rl(x_r, 5), rl(x_r, 6)};
  float nq[4];
  quat_mul(rq, xq, nq);
  quat_norm_approx(nq);
  float out = x_r;
  if (lane < 3) {
    out = x_r + alpha * p_r;
  } else if (lane < 7) {
    out = lane == 3 ? nq[0] : lane == 4 ? nq[1] : lane == 5 ? nq[2] : nq[3];
  } else if (lane < 7 + nt) {
    const float t = x_r + norm_angle(alpha * p_up);
    out = norm_angle(t);
  }
  return out;
}

// Workgroup layout of the minimiser kernels: W wavefronts per chain.  Every wave keeps a full private copy of
// the chain's state (workspace `stride` floats apart in LDS) and runs the same deterministic code, so the copies
// stay bit-identical; the waves differ only inside the line search, where wave u evaluates trial t0 + u.
struct WaveTeam {
  int W, wv;        // waves per chain, this wave's index
  long stride;      // floats between the private workspaces of consecutive waves
  float *sh_f;      // [W] shared: trial energies
  int *sh_ok;       // [W] shared: trial accepted
};

// bfgs<> (bfgs.h:357-502) on the conformation in k.x (LDS, in/out); returns the final energy in every
// lane, leaves the final gradient in k.g.  All control flow is uniform over the workgroup.  The vectors
// (x, g, p, y, -Hy) live one element per lane in registers (6 + T <= 63 elements); only the conformation
// handed to eval_conf, the change it returns and H are in LDS.
//
// fast_line_search (bfgs.h:73-91) tries alpha = 1, 1/2, 1/4, ... and stops at the first trial that satisfies
// the sufficient-decrease test.  The trials do not depend on one another, so a team of W waves evaluates W
// consecutive trials at once and then takes the first accepted one in trial order: the same conformation,
// energy, gradient and alpha as the sequential search (on average 2.0 sequential evaluations per search
// become 1.16 rounds at W = 4 on the C3 complex).  `evals` counts what the sequential search would have
// evaluated.
//
// The start point is evaluated by the same eval_conf call as the trials (step -1, a "search" of one trial that
// is always accepted): eval_conf is inlined, and one call site keeps the minimiser kernels' code small enough
// to stay in the instruction cache while every access keeps its address space (LDS reads stay ds_read).
// ACC: accurate_line_search compiled in (its own kernel instantiations: the search's extra live state costs the
// register-budgeted default kernels spills otherwise)
template <int HREG = kHReg, int PG = kPairGroup, bool ACC = false>
__device__ __forceinline__ float bfgs_wave(const VinaEnv &env, const VinaLigand &L, const WaveWork &w, const BfgsWork &k,
                                           float v0, float v1, float v2, int max_iters, int &evals, const WaveTeam &tm,
                                           long long *eval_ticks = nullptr) {
  const int nt = L.n_nodes - 1, n = 6 + nt, nc = 7 + nt;
  const int lane = threadIdx.x & 63;
  const int row = lane < n ? lane : n - 1;
  const int W = tm.W, wv = tm.wv;
  float *x = k.x, *x_new = k.x_new, *g_new = k.g_new, *h = k.h;
  float x_r = lane < nc ? x[lane] : 0.f;
  const float x_orig = x_r;
  // H: up to HREG (kHReg) variables, lane i keeps the full (symmetric) row i in registers -- both triangles are updated
  // with the operands in the reference's order, so they stay bit-identical to the triangular update; larger
  // problems keep the triangle in LDS.
  const bool hreg = n <= HREG;
  float hrow[HREG];
#pragma unroll
  for (int j = 0; j < HREG; j++) hrow[j] = (j == lane && lane < n) ? 1.f : 0.f;
  if (!hreg) {
    for (int i = lane; i < n * (n + 1) / 2; i += 64) h[i] = 0.f;
    wave_sync();
    if (lane < n) h[hidx(lane, lane)] = 1.f;
  }
  auto minus_hrow_times = [&](float v) {  // -(H v)[lane], j ascending
    float sum = 0.f;
#pragma unroll
    for (int j0 = 0; j0 < HREG; j0 += 4) {
      if (j0 < n) {
#pragma unroll
        for (int u = 0; u < 4; u++) sum += hrow[j0 + u] * rl(v, j0 + u);
      }
    }
    return -sum;
  };
  float f0 = 0.f, f_orig = 0.f, g_r = 0.f, g_orig = 0.f;
  float xl_r = x_r;  // the last conformation handed to eval_conf: what the reference's `model` holds afterwards

  for (int step = -1; step < max_iters; step++) {
    const bool start = step < 0;
    float p_r = 0.f, p_up = 0.f, pg = 0.f;
    long long tb = eval_ticks ? wall_clock64() : 0;
    auto lapb = [&](int slot) {
      if (eval_ticks) {
        const long long t = wall_clock64();
        eval_ticks[slot] += t - tb;
        tb = t;
      }
    };
    // minimization_params::Simple (--simple_ascent; simple_gradient_ascent, bfgs.h:234-355): steepest descent with the
    // accurate line search -- p = -g every iteration, no quasi-Newton update (env.accurate_ls == 2, ACC kernels)
    const bool simple = ACC && env.accurate_ls == 2;
    if (!start) {
      p_r = simple ? (lane < n ? -g_r : 0.f) : hreg ? minus_hrow_times(g_r) : (lane < n ? minus_h_times(h, g_r, n, row) : 0.f);
      p_up = __shfl_up(p_r, 1);
      pg = dot_lanes(p_r, g_r, n);
    }
    lapb(2);
    float f1 = 0.f;
    int t_acc = -1, winner = 0;  // accepted trial (10 = none of the ten), wave that evaluated it
    // accurate_line_search (bfgs.h:104-180, after Numerical Recipes' lnsrch; --accurate_line_search): every trial's
    // step comes from the energies of the previous ones, so the trials are sequential; the waves of a team each run
    // the same search on their own workspace (identical results).  It shares the trial loop -- and with it the one
    // inlined eval_conf call site -- with fast_line_search.  fl arithmetic; the literals 2.0 / 3.0 are double as written.
    const bool acc_ls = ACC && !start;
    float alpha_acc = 0.f;  // its result, 0 = "line direction was wrong, give up"
    float ls_alpha = 1.0f, ls_alpha2 = 0.f, ls_f2 = 0.f, alamin = 0.f;
    const float slope = pg;
    if (acc_ls) {
      if (slope >= 0) break;  // x_new = x, g_new cleared, alpha = 0 -> bfgs gives up (bfgs.h:116-122,417-426)
      // compute_lambdamin (bfgs.h:93-102): max |p(i)| / max(|x(i)|, 1) with x in change indexing (conf.h:459-490):
      // position, quaternion_to_angle(orientation) (quaternion.cu:46-62), torsions
      const float qa = rl(x_r, 3), qb = rl(x_r, 4), qc = rl(x_r, 5), qd = rl(x_r, 6);
      float ang0 = 0.f, ang1 = 0.f, ang2 = 0.f;
      if (qa > -1 && qa < 1) {
        float angle = 2 * acosf_ref(qa);
        if (angle > VPI) angle -= 2 * VPI;
        float sn, cs_unused;
        sincos_ref(angle / 2, sn, cs_unused);
        if (!(fabsf(sn) < VEPS)) {
          const float sc = angle / sn;
          ang0 = qb * sc, ang1 = qc * sc, ang2 = qd * sc;
        }
      }
      const float x_tor = __shfl_down(x_r, 1);  // change index 6 + k <-> conf index 7 + k
      const float xi = lane < 3 ? x_r : lane == 3 ? ang0 : lane == 4 ? ang1 : lane == 5 ? ang2 : x_tor;
      float test = lane < n ? fabsf(p_r) / fmaxf(fabsf(xi), 1.0f) : 0.f;
      for (int m = 32; m >= 1; m >>= 1) test = fmaxf(test, __shfl_xor(test, m));
      alamin = VEPS / test;
    }
    // fast_line_search: trials t0 .. t0 + W - 1 in parallel
    for (int t0 = 0; t_acc < 0; t0 += W) {
      const int t = t0 + wv < 10 ? t0 + wv : 9;
      const float a_t = acc_ls ? ls_alpha : ldexpf(1.f, -t);  // 1 halved t times, exactly
      const float xn = start ? x_r : increment_lanes(x_r, p_r, p_up, a_t, nt, lane);
      if (lane < nc) x_new[lane] = xn;
      wave_sync();
      lapb(3);
      const float f_t = eval_conf<0, PG>(env, L, x_new, v0, v1, v2, w, g_new);
      lapb(0);
      if (start) {  // every wave evaluated the start point itself
        t_acc = 0;
        winner = wv;
        f1 = f_t;
      } else if (acc_ls) {
        evals++;
        xl_r = xn;
        winner = wv;
        f1 = f_t;
        if (ls_alpha < alamin || !isfinite(ls_alpha)) {  // too small a step
          t_acc = 0;
        } else if (f_t <= f0 + 1.0e-4f * ls_alpha * slope) {  // sufficient decrease
          alpha_acc = ls_alpha;
          t_acc = 0;
        } else {  // backtrack
          float tmplam;
          if (ls_alpha == 1.0f) {
            tmplam = (float)(-(double)slope / (2.0 * (double)(f_t - f0 - slope)));
          } else {
            const float rhs1 = f_t - f0 - ls_alpha * slope, rhs2 = ls_f2 - f0 - ls_alpha2 * slope;
            const float ca = (rhs1 / (ls_alpha * ls_alpha) - rhs2 / (ls_alpha2 * ls_alpha2)) / (ls_alpha - ls_alpha2);
            const float cb = (-ls_alpha2 * rhs1 / (ls_alpha * ls_alpha) + ls_alpha * rhs2 / (ls_alpha2 * ls_alpha2)) /
                             (ls_alpha - ls_alpha2);
            if (ca == 0.0f) {
              tmplam = (float)(-(double)slope / (2.0 * (double)cb));
            } else {
              const float disc = (float)((double)(cb * cb) - 3.0 * (double)ca * (double)slope);
              if (disc < 0) tmplam = 0.5f * ls_alpha;
              else if (cb <= 0) tmplam = (float)((double)(-cb + sqrtf(disc)) / (3.0 * (double)ca));
              else tmplam = -slope / (cb + sqrtf(disc));
            }
            if (tmplam > 0.5f * ls_alpha) tmplam = 0.5f * ls_alpha;  // always at least cut in half
          }
          ls_alpha2 = ls_alpha;
          ls_f2 = f_t;
          ls_alpha = fmaxf(tmplam, 0.1f * ls_alpha);  // never smaller than a tenth
        }
      } else if (W == 1) {
        if (f_t - f0 < 0.0001f * a_t * pg) {
          t_acc = t0;
        } else if (t0 + 1 >= 10) {
          t_acc = 10;
        }
        f1 = f_t;
      } else {
        const bool ok = t0 + wv < 10 && f_t - f0 < 0.0001f * a_t * pg;
        if (lane == 0) {
          tm.sh_f[wv] = f_t;
          tm.sh_ok[wv] = ok ? 1 : 0;
        }
        __syncthreads();
        int win = -1;
        for (int u = W - 1; u >= 0; u--)
          if (tm.sh_ok[u]) win = u;
        if (win >= 0) {
          t_acc = t0 + win;
          winner = win;
        } else if (t0 + W >= 10) {  // never accepted: the search ends on trial 9's evaluation (bfgs.h:82-90)
          t_acc = 10;
          winner = 9 - t0;
        }
        f1 = tm.sh_f[winner];
        __syncthreads();
      }
    }
    if (acc_ls && alpha_acc == 0.f) break;  // bfgs.h:417-426: give up (x, g, f0 stay)
    lapb(4);
    const float alpha = acc_ls ? alpha_acc : ldexpf(1.f, -t_acc);
    if (!acc_ls) evals += t_acc < 10 ? t_acc + 1 : 10;
    // the accepted trial's conformation and gradient, from the workspace of the wave that evaluated it
    const float *xw = x_new + (long)(winner - wv) * tm.stride, *gw = g_new + (long)(winner - wv) * tm.stride;
    const float xn_r = lane < nc ? xw[lane] : 0.f;
    const float gn_r = lane < n ? gw[lane] : 0.f;
    if (W > 1 && !start) __syncthreads();  // everyone has read before the next trial overwrites x_new / g_new
    if (start) {
      f0 = f_orig = f1;
      g_r = g_orig = gn_r;
      continue;
    }
    const float y_r = gn_r - g_r;
    f0 = f1;
    x_r = xn_r;
    if (ACC) xl_r = xn_r;
    g_r = gn_r;
    const float gradnormsq = dot_lanes(g_r, g_r, n);
    if (!(gradnormsq >= 1e-4f)) break;
    if (simple) continue;
    if (step == 0) {
      const float yy = dot_lanes(y_r, y_r, n);
      if (fabsf(yy) > VEPS) {
        const float dgl = alpha * dot_lanes(y_r, p_r, n) / yy;
        if (hreg) {
#pragma unroll
          for (int j = 0; j < HREG; j++)
            if (j == lane && lane < n) hrow[j] = dgl;
        } else {
          if (lane < n) h[hidx(lane, lane)] = dgl;
          wave_sync();
        }
      }
    }
    // bfgs_update (bfgs.h:52-66)
    const float yp = dot_lanes(y_r, p_r, n);
    if (!(alpha * yp < VEPS)) {
      const float mhy_r = hreg ? minus_hrow_times(y_r) : (lane < n ? minus_h_times(h, y_r, n, row) : 0.f);
      const float yhy = -dot_lanes(y_r, mhy_r, n);
      const float r = 1 / (alpha * yp);
      if (hreg) {
        // H(i, j) += alpha r (mhy_i p_j + mhy_j p_i) + alpha alpha (r r yhy + r) p_lo p_hi, lo = min(i, j): lane i
        // updates its whole row; the products keep the reference's operand order for the upper triangle
        const float ar = alpha * r, c = alpha * alpha * (r * r * yhy + r);
        const float cp_r = c * p_r;
#pragma unroll
        for (int j0 = 0; j0 < HREG; j0 += 4) {
          if (j0 < n) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int j = j0 + u;
              const float pj = rl(p_r, j), mj = rl(mhy_r, j), cpj = rl(cp_r, j);
              const float t = mhy_r * pj + mj * p_r;
              const float q = j >= lane ? cp_r * pj : cpj * p_r;
              hrow[j] += ar * t + q;
            }
          }
        }
      } else {
        // row i of the upper triangle per pass, lane j >= i owns H(i, j); four rows per LDS latency
        for (int i0 = 0; i0 < n; i0 += 4) {
          float hv[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int i = i0 + u;
            hv[u] = (i < n && lane >= i && lane < n) ? h[hidx(i, lane)] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int i = i0 + u;
            if (i < n) {
              const float mi = rl(mhy_r, i), pi = rl(p_r, i);
              const float upd = alpha * r * (mi * p_r + mhy_r * pi) + alpha * alpha * (r * r * yhy + r) * pi * p_r;
              if (lane >= i && lane < n) h[hidx(i, lane)] = hv[u] + upd;
            }
          }
        }
      }
    }
    wave_sync();
    lapb(5);
  }
  // What `model` holds after the call is the conformation of the last evaluation: x before the revert below (every
  // iteration ends with x = x_new, accepted trial or not).  Monte-Carlo's update_energy and gyration_radius read it
  // (monte_carlo.cpp:44-47, mutate.cpp:55); it is left in k.x_new.
  if (lane < nc) x_new[lane] = ACC ? xl_r : x_r;  // (xl_r = x_r unless an accurate line search gave up after its trials)
  if (!(f0 <= f_orig)) {  // bfgs.h:491-495
    f0 = f_orig;
    x_r = x_orig;
    g_r = g_orig;
  }
  if (lane < nc) x[lane] = x_r;
  if (lane < n) k.g[lane] = g_r;
  wave_sync();
  return f0;
}

// TP: the throughput tuning (see kPairGroupTp / kHRegTp) with the register budget of three waves per SIMD
template <bool TP, bool ACC = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TP ? 3 : 1, TP ? 3 : 2))) void vina_bfgs_kernel(
    VinaEnv env, VinaLigand L, float *confs, float v0, float v1, float v2, int max_iters, float *energy, float *grad_out,
    int *evals_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *pp = lds;
  if (env.stage) L = stage_ligand(L, pp);
  WaveWork w = carve_work(pp, L);
  const int nt = L.n_nodes - 1, n = 6 + nt, nc = 7 + nt;
  BfgsWork k = carve_bfgs(pp, n, nc);
  const int b = blockIdx.x, lane = threadIdx.x;
  int evals = 0;
  for (int i = lane; i < nc; i += 64) k.x[i] = confs[(size_t)b * nc + i];
  wave_sync();
  const WaveTeam solo{1, 0, 0, nullptr, nullptr};
  const float f0 = bfgs_wave<TP ? kHRegTp : kHReg, TP ? kPairGroupTp : kPairGroup, ACC>(env, L, w, k, v0, v1, v2, max_iters,
                                                                                       evals, solo);
  for (int i = lane; i < nc; i += 64) confs[(size_t)b * nc + i] = k.x[i];
  if (grad_out)
    for (int i = lane; i < n; i += 64) grad_out[(size_t)b * n + i] = k.g[i];
  if (lane == 0) {
    energy[b] = f0;
    if (evals_out) evals_out[b] = evals;
  }
}


// refine_structure (main.cpp:131-171): BFGS on the direct receptor term; the out-of-box slope starts at
// 10 and is raised 10x per try (at most 5) until every heavy atom is inside the box (non_cache::within,
// non_cache.cpp:84-101); a pose that never gets in reports max_fl.
template <bool ACC>
__global__ __launch_bounds__(64) void vina_refine_kernel(VinaEnv env, VinaLigand L, float *confs, float v0, float v1,
                                                         float v2, int max_iters, float *energy, int *tries_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *pp = lds;
  if (env.ligs) {
    const int l = env.item_lig[blockIdx.x];
    L = env.ligs[l];
    max_iters = env.lig_iters[l];
  }
  if (env.stage) L = stage_ligand(L, pp);
  WaveWork w = carve_work(pp, L);
  const int nt = L.n_nodes - 1, n = 6 + nt, nc = 7 + nt;
  const size_t cs = env.ligs ? env.conf_stride : nc;
  BfgsWork k = carve_bfgs(pp, n, nc);
  const int b = blockIdx.x, lane = threadIdx.x;
  int evals = 0;
  for (int i = lane; i < nc; i += 64) k.x[i] = confs[b * cs + i];
  wave_sync();
  env.direct = 1;
  float slope = 10.f, e = 0.f;
  int tries = 0;
  bool inside = false;
  for (int p = 0; p < 5 && !inside; p++) {
    env.slope = slope;
    e = bfgs_wave<kHReg, kPairGroup, ACC>(env, L, w, k, v0, v1, v2, max_iters, evals, WaveTeam{1, 0, 0, nullptr, nullptr});
    (void)eval_conf<3>(env, L, k.x, 0.f, 0.f, 0.f, w, nullptr);  // m.set(out.c)
    int bad = 0;
    for (int i = lane; i < L.n_atoms; i += 64)
      if (L.smt[i] > 1)
        for (int d = 0; d < 3; d++)
          if (w.coords[3 * i + d] < env.box_begin[d] - 0.0001f || w.coords[3 * i + d] > env.box_end[d] + 0.0001f) bad = 1;
    inside = !__any(bad);
    tries++;
    slope *= 10.f;
    wave_sync();
  }
  if (!inside) e = VMAXFL;
  for (int i = lane; i < nc; i += 64) confs[b * cs + i] = k.x[i];
  if (lane == 0) {
    energy[b] = e;
    if (tries_out) tries_out[b] = tries;
  }
}

void launch_vina_refine(const VinaEnv &env0, const VinaLigand &lig, float *confs, int B, float v0, float v1, float v2,
                        int max_iters, float *energy, int *tries, hipStream_t s) {
  VinaEnv env = env0;
  env.stage = want_stage(B) ? 1 : 0;
  const size_t lds = vina_wave_lds_bytes(lig.n_atoms, lig.n_nodes, lig.n_pairs, true, env.stage);
  if (env.accurate_ls) {
    big_lds(vina_refine_kernel<true>);
    hipLaunchKernelGGL(vina_refine_kernel<true>, dim3(B), dim3(64), lds, s, env, lig, confs, v0, v1, v2, max_iters, energy,
                       tries);
  } else {
    big_lds(vina_refine_kernel<false>);
    hipLaunchKernelGGL(vina_refine_kernel<false>, dim3(B), dim3(64), lds, s, env, lig, confs, v0, v1, v2, max_iters, energy,
                       tries);
  }
}

// ---------------------------------------------------------------------------------------------
// Monte-Carlo chain: monte_carlo::operator() (monte_carlo.cpp:99-148) -- one wavefront per chain for
// the WHOLE chain (mutate -> BFGS(hunt cap) -> Metropolis -> BFGS(full cap) -> container insert),
// no host round trip between steps.
// RNG: the reference's boost::mt19937 (= the standard MT19937, one 624-word state per wave in global memory, seeded
// on the host, regenerated by the wave's 64 lanes every 624 draws) under restatements of the Boost distributions
// random.cpp draws from -- the same ones oracle/ref_shims/boost/random.hpp gives the reference code in oracle/_ref,
// whose chains oracle/vina_ref.c follows bit for bit (tests/test_ref_vina.py).  On the device logf / cosf and the
// force sums differ in the last bits, so chains agree with the reference's for their first steps and statistically
// afterwards.
// ---------------------------------------------------------------------------------------------
// order the memory accesses of one wave's lanes (the wave runs in lock step; this is the compiler / counter fence)
__device__ __forceinline__ void wsync() {
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
}

struct McRng {
  unsigned *mt;  // this wave's 624-word state (global memory)
  int idx;       // next word; 624 = regenerate first
  // the state is rewritten by this wave's own lanes: read it past the (non-coherent) vector L1
  __device__ __forceinline__ unsigned ld(int i) const { return __hip_atomic_load(&mt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  // The sequential twist reads mt[i + 1] before it is replaced and mt[(i + 397) % 624] after it was (for i >= 227):
  // three ranges whose inputs are complete when the range starts, 64 words at a time, loads before stores.
  __device__ __forceinline__ void twist() {
    const int lane = threadIdx.x & 63;
    const int lo[3] = {0, 227, 454}, hi[3] = {227, 454, 624};
#pragma unroll 1
    for (int ph = 0; ph < 3; ph++)
#pragma unroll 1
      for (int i0 = lo[ph]; i0 < hi[ph]; i0 += 64) {
        const int i = i0 + lane;
        unsigned v = 0;
        if (i < hi[ph]) {
          const unsigned y = (ld(i) & 0x80000000u) | (ld(i + 1 < 624 ? i + 1 : 0) & 0x7fffffffu);
          v = ld(i + 397 < 624 ? i + 397 : i - 227) ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        wsync();
        if (i < hi[ph]) __hip_atomic_store(&mt[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        wsync();
      }
    idx = 0;
  }
  __device__ __forceinline__ unsigned u32() {
    if (idx >= 624) twist();
    unsigned y = ld(idx++);
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  // boost::uniform_real<float> (random_fl, random.cpp:27-35): u32 / 2^32 * (b - a) + a, redrawn if it reaches b
  __device__ __forceinline__ float fl(float a, float b) {
    for (;;) {
      const float result = (float)u32() / 4294967296.0f * (b - a) + a;
      if (result < b) return result;
    }
  }
  // boost::uniform_int<int> (random_int, random.cpp:44-52): equal buckets with rejection
  __device__ __forceinline__ int irange(int a, int b) {
    const unsigned range = (unsigned)b - (unsigned)a, brange = 0xffffffffu;
    if (range == 0) return a;
    unsigned bucket = brange / (range + 1);
    if (brange % (range + 1) == range) ++bucket;
    for (;;) {
      const unsigned q = u32() / bucket;
      if (q <= range) return (int)(q + (unsigned)a);
    }
  }
  // boost::normal_distribution<float>(0, 1), Box-Muller on two fresh uniforms (random_normal builds a new
  // distribution object per call, random.cpp:37-42)
  __device__ __forceinline__ float normal() {
    const float r1 = fl(0.f, 1.f), r2 = fl(0.f, 1.f);
    float sn, cs;
    sincos_ref(2.0f * 3.14159265358979323846f * r1, sn, cs);
    return sqrtf(-2.0f * logf_ref(1.0f - r2)) * cs;
  }
  __device__ __forceinline__ void inside_sphere(float &x, float &y, float &z) {  // random.cpp:66-75
    for (;;) {
      x = fl(-1, 1);
      y = fl(-1, 1);
      z = fl(-1, 1);
      if (x * x + y * y + z * z < 1) return;
    }
  }
};

// add_to_output_container (coords.cpp:25-56) + out.sort() for one chain, run by one wave: the candidate (tmp, hc,
// tmp_e) against the chain's physical container (s_e / s_conf / s_xyz in global memory) through `ord` (sorted position
// -> physical slot) with the scratch `rm`.  Returns the container's new size.  wsync() orders the lanes' accesses.
__device__ __forceinline__ int container_insert(const VinaMcArgs &a, float *s_e, float *s_conf, float *s_xyz, int cs_,
                                                int xs_, int *ord, float *rm, const float *tmp, const float *hc,
                                                float tmp_e, int n_out, int nc, int nh) {
  const int lane = threadIdx.x & 63;
  // rmsd to every saved pose, one pose per lane
  for (int o = lane; o < n_out; o += 64) {
    const float *ref = s_xyz + (size_t)ord[o] * xs_;
    float acc = 0.f;
    for (int i = 0; i < nh; i++) {  // rmsd_upper_bound: one vec_distance_sqr per atom (coords.cpp:25-32)
      const float dx = hc[3 * i] - ref[3 * i], dy = hc[3 * i + 1] - ref[3 * i + 1], dz = hc[3 * i + 2] - ref[3 * i + 2];
      acc += dx * dx + dy * dy + dz * dz;
    }
    rm[o & 63] = nh > 0 ? sqrtf(acc / (float)nh) : 0.f;
  }
  wsync();
  int closest = n_out;
  float closest_rmsd = VMAXFL;
  for (int o = 0; o < n_out && o < 64; o++) {  // first minimum, like find_closest
    const float r = rm[o];
    if (o == 0 || r < closest_rmsd) {
      closest = o;
      closest_rmsd = r;
    }
  }
  int pos = -1;
  if (closest < n_out && closest_rmsd < a.min_rmsd) {
    if (tmp_e < s_e[ord[closest]]) pos = closest;
  } else if (n_out < a.num_saved) {
    pos = n_out;
    if (lane == 0) ord[pos] = n_out;
    n_out++;
  } else if (n_out > 0 && tmp_e < s_e[ord[n_out - 1]]) {
    pos = n_out - 1;
  }
  wsync();
  if (pos >= 0) {
    const int phys = ord[pos];
    if (lane == 0) s_e[phys] = tmp_e;
    for (int i = lane; i < nc; i += 64) s_conf[(size_t)phys * cs_ + i] = tmp[i];
    for (int i = lane; i < 3 * nh; i += 64) s_xyz[(size_t)phys * xs_ + i] = hc[i];
    wsync();
    if (lane == 0) {  // out.sort(): keep `ord` ordered by energy
      int o = pos;
      while (o > 0 && s_e[ord[o]] < s_e[ord[o - 1]]) {
        const int t = ord[o];
        ord[o] = ord[o - 1];
        ord[o - 1] = t;
        o--;
      }
      while (o + 1 < n_out && s_e[ord[o + 1]] < s_e[ord[o]]) {
        const int t = ord[o];
        ord[o] = ord[o + 1];
        ord[o + 1] = t;
        o++;
      }
    }
    wsync();
  }
  return n_out;
}

// mutate_conf (mutate.cpp:35-73) of the conformation in x (LDS), one wave; mconf = what `model` holds (gyration radius)
template <int PG>
__device__ __forceinline__ void mutate_wave(const VinaEnv &env, const VinaLigand &L, const WaveWork &w, McRng &rng, float *x,
                                            const float *mconf, float amplitude) {
  const int lane = threadIdx.x & 63, nt = L.n_nodes - 1;
  const int which = rng.irange(0, 2 + nt - 1);
  if (which == 0) {
    float dx, dy, dz;
    rng.inside_sphere(dx, dy, dz);
    if (lane == 0) {
      x[0] += amplitude * dx;
      x[1] += amplitude * dy;
      x[2] += amplitude * dz;
    }
  } else if (which == 1) {
    (void)eval_conf<3, PG>(env, L, mconf, 0.f, 0.f, 0.f, w, nullptr);  // the coordinates `model` holds
    float acc = 0.f;
    int n_gyr = 0;
    for (int i = L.lig_begin; i < L.lig_end; i++) n_gyr += L.smt[i] > 1;
    if (env.strict) {  // the reference's loop adds the atoms one after the other (model.cpp:1002-1014)
      for (int base = L.lig_begin; base < L.lig_end; base += 64) {
        const int i = base + lane;
        const bool on = i < L.lig_end && L.smt[i < L.lig_end ? i : base] > 1;
        float d2 = 0.f;
        if (on) {
          const float dx = w.coords[3 * i] - w.origin[0], dy = w.coords[3 * i + 1] - w.origin[1],
                      dz = w.coords[3 * i + 2] - w.origin[2];
          d2 = dx * dx + dy * dy + dz * dz;
        }
        seq_add(acc, d2, __builtin_amdgcn_ballot_w64(on));
      }
    } else {
      for (int i = L.lig_begin + lane; i < L.lig_end; i += 64)
        if (L.smt[i] > 1) {
          const float dx = w.coords[3 * i] - w.origin[0], dy = w.coords[3 * i + 1] - w.origin[1],
                      dz = w.coords[3 * i + 2] - w.origin[2];
          acc += dx * dx + dy * dy + dz * dz;
        }
      acc = wave_sum(acc);
    }
    const float gr = n_gyr > 0 ? sqrtf(acc / (float)n_gyr) : 0.f;  // model::gyration_radius, model.cpp:1002-1014
    if (gr > VEPS) {
      float dx, dy, dz;
      rng.inside_sphere(dx, dy, dz);
      const float sc = amplitude / gr;
      if (lane == 0) {
        float rot[6] = {0.f, 0.f, 0.f, sc * dx, sc * dy, sc * dz};
        conf_increment(x, rot, 1.0f, 0);
      }
    }
  } else {
    const float tv = rng.fl(-VPI, VPI);
    if (lane == 0) x[7 + (which - 2)] = tv;
  }
  wave_sync();
}

// conf::randomize (conf.h:119-122,189-192) into x (LDS): every lane draws the same numbers
__device__ __forceinline__ void randomize_wave(McRng &rng, float *x, const float *c1, const float *c2, int nt) {
  const int lane = threadIdx.x & 63;
  float px = rng.fl(c1[0], c2[0]), py = rng.fl(c1[1], c2[1]), pz = rng.fl(c1[2], c2[2]);
  float q0, q1, q2, q3, nrm;
  do {  // random_orientation (quaternion.cu:81-94); abs(qt) scales by the largest component (quaternion.h:169-190)
    q0 = rng.normal();
    q1 = rng.normal();
    q2 = rng.normal();
    q3 = rng.normal();
    const float maxim = fmaxf(fmaxf(fabsf(q0), fabsf(q1)), fmaxf(fabsf(q2), fabsf(q3)));
    nrm = 0.f;
    if (maxim != 0.f) {
      const float mixam = (float)(1.0 / (double)maxim);
      const float v0 = q0 * mixam, v1 = q1 * mixam, v2 = q2 * mixam, v3 = q3 * mixam;
      float sum = v0 * v0;
      sum += v1 * v1;
      sum += v2 * v2;
      sum += v3 * v3;
      nrm = maxim * sqrtf(sum);
    }
  } while (!(nrm > VEPS));
  if (lane == 0) {
    x[0] = px, x[1] = py, x[2] = pz;
    x[3] = q0 / nrm, x[4] = q1 / nrm, x[5] = q2 / nrm, x[6] = q3 / nrm;
  }
  for (int t = 0; t < nt; t++) {
    float tv = rng.fl(-VPI, VPI);
    if (lane == 0) x[7 + t] = tv;
  }
}

// W = blockDim.x / 64 waves per chain (see WaveTeam): every wave replays the whole chain -- same RNG stream,
// same decisions -- and they share the work only inside the BFGS line searches.  Wave 0 alone owns the output
// container (global memory) and publishes what the others need (the container's size) through LDS.
template <bool PROF, bool TP, bool ACC = false>
__device__ __forceinline__ void mc_chain(VinaEnv env, VinaLigand L, VinaMcArgs a) {
  if (!PROF) a.prof = nullptr;  // the timing code folds away in the production instantiations
  constexpr int HR = TP ? kHRegTp : kHReg, PG = TP ? kPairGroupTp : kPairGroup;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *pp = lds;
  // workgroups start in index order and the launch ends with its slowest chain: screen launches hand the chains
  // out longest first (a.order), whatever order the caller listed them in
  const int chain = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
  if (a.ligs) {  // screen mode: this chain's ligand and search length
    const int l = a.chain_lig[chain];
    L = a.ligs[l];
    a.n_steps = a.lig_steps[l];
    a.max_iters = a.lig_iters[l];
  }
  if (env.stage) L = stage_ligand(L, pp);
  const int W = blockDim.x >> 6, wv = threadIdx.x >> 6;
  const int nt = L.n_nodes - 1, n = 6 + nt, nc = 7 + nt, nh = L.n_heavy;
  float *sh_f = carve(pp, W);
  int *sh_ok = reinterpret_cast<int *>(carve(pp, W));
  int *sh_n = reinterpret_cast<int *>(carve(pp, 4));
  long stride;
  {  // footprint of one wave's private workspace
    float *q = pp;
    (void)carve_work(q, L, false);
    (void)carve_bfgs(q, n, nc);
    (void)carve(q, nc);
    (void)carve(q, nc);
    (void)carve(q, 3 * nh);
    (void)carve(q, 64);
    (void)carve(q, a.num_saved);
    stride = q - pp;
  }
  pp += wv * stride;
  WaveWork w = carve_work(pp, L);
  BfgsWork k = carve_bfgs(pp, n, nc);
  float *tmp = carve(pp, nc);
  float *mconf = carve(pp, nc);  // the conformation the reference's `model` holds (see bfgs_wave's closing comment)
  float *hc = carve(pp, 3 * nh);
  float *rm = carve(pp, 64);                       // rmsd of the candidate to each saved pose (wave 0)
  int *ord = reinterpret_cast<int *>(carve(pp, a.num_saved));  // sorted position -> physical slot (wave 0)
  const WaveTeam tm{W, wv, stride, sh_f, sh_ok};
  const int b = chain, lane = threadIdx.x & 63;
  McRng rng{a.mt + ((size_t)b * a.mt_team + wv) * 624, 624};
  int evals = 0;
  // scratch container of this chain (physical slots)
  float *s_e = a.scratch_e + (size_t)b * a.num_saved;
  const int cs_ = a.conf_stride, xs_ = a.coord_stride;  // >= nc, >= 3 nh
  float *s_conf = a.scratch_conf + (size_t)b * a.num_saved * cs_;
  float *s_xyz = a.scratch_coords + (size_t)b * a.num_saved * xs_;

  {
    randomize_wave(rng, tmp, a.c1, a.c2, nt);
    // before the first evaluation `model` holds the input pose: zero torsions (its rigid placement does not matter
    // to the gyration radius, the only thing read from it)
    for (int i = lane; i < nc; i += 64) mconf[i] = i == 3 ? 1.f : 0.f;
  }
  wave_sync();
  float tmp_e = 0.f, best_e = VMAXFL;
  int n_out = 0;
  // optional phase timing (MI_VINA_MC_PROFILE): 100 MHz ticks of [mutate, hunt BFGS, energy + Metropolis,
  // second BFGS, energy + copy, container insert, evaluations inside both BFGS, accepted steps]
  long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = a.prof ? wall_clock64() : 0;
  auto lap = [&](int ph) {
    if (a.prof) {
      const long long t = wall_clock64();
      pt[ph] += t - t_prev;
      t_prev = t;
    }
  };
  long long *evt = a.prof ? &pt[6] : nullptr;

  for (int step = 0; step < a.n_steps; step++) {
    for (int i = lane; i < nc; i += 64) k.x[i] = tmp[i];
    wave_sync();
    mutate_wave<PG>(env, L, w, rng, k.x, mconf, a.amplitude);
    lap(0);
    // two passes through the same code: pass 0 = BFGS with the hunt caps + Metropolis, pass 1 (only for an
    // accepted candidate that is the best so far or while the container is not full) = BFGS with the authentic
    // caps + container insert (monte_carlo.cpp:120-143).  One call site for the minimiser keeps the kernel small.
    float cand_e = 0.f;
    for (int pass = 0; pass < 2; pass++) {
      const float *cap = pass == 0 ? a.hunt : a.auth;
      (void)bfgs_wave<HR, PG, ACC>(env, L, w, k, cap[0], cap[1], cap[2], a.max_iters, evals, tm, evt);
      lap(pass == 0 ? 1 : 3);
      // update_energy (monte_carlo.cpp:44-47): ig.eval on the coordinates `model` holds = the last evaluated
      // conformation (k.x_new), which is k.x unless bfgs reverted to its start
      bool reverted = false;
      for (int i = lane; i < nc; i += 64) reverted |= k.x_new[i] != k.x[i];
      reverted = __any(reverted);
      const float e_now = eval_conf<2, PG>(env, L, k.x_new, 0.f, a.auth[1], 0.f, w, nullptr);  // leaves its coords in w.coords
      if (pass == 0) {
        cand_e = e_now;
        bool accept = step == 0 || cand_e < tmp_e;
        if (!accept) {  // metropolis_accept, monte_carlo.cpp:38-42
          const float prob = expf_ref((tmp_e - cand_e) / a.temperature);
          accept = rng.fl(0.f, 1.f) < prob;
        }
        lap(2);
        if (!accept) {  // `model` keeps the rejected candidate's last evaluation
          for (int i = lane; i < nc; i += 64) mconf[i] = k.x_new[i];
          wave_sync();
          break;
        }
        tmp_e = cand_e;
        pt[7] += a.prof ? 1 : 0;
        if (!(tmp_e < best_e || n_out < a.num_saved)) {
          for (int i = lane; i < nc; i += 64) tmp[i] = mconf[i] = k.x[i];  // tmp = candidate; m.set(tmp.c)
          wave_sync();
          break;
        }
      } else {
        tmp_e = e_now;
        for (int i = lane; i < nc; i += 64) tmp[i] = mconf[i] = k.x[i];  // m.set(tmp.c), monte_carlo.cpp:131
        if (reverted) {  // the coordinates in w.coords are the last trial's, not tmp's
          wave_sync();
          (void)eval_conf<3, PG>(env, L, k.x, 0.f, 0.f, 0.f, w, nullptr);
        }
        for (int h = lane; h < nh; h += 64) {
          const int i = L.heavy_list[h];
          hc[3 * h] = w.coords[3 * i];
          hc[3 * h + 1] = w.coords[3 * i + 1];
          hc[3 * h + 2] = w.coords[3 * i + 2];
        }
        wave_sync();
        lap(4);
        if (wv == 0) {  // the container belongs to wave 0; wsync() orders its lanes' LDS / global accesses
          n_out = container_insert(a, s_e, s_conf, s_xyz, cs_, xs_, ord, rm, tmp, hc, tmp_e, n_out, nc, nh);
          if (lane == 0) sh_n[0] = n_out;
        }
        __syncthreads();
        n_out = sh_n[0];
        if (tmp_e < best_e) best_e = tmp_e;
        lap(5);
      }
    }
  }
  if (wv != 0) return;
  if (a.prof && lane == 0)
    for (int i = 0; i < 12; i++) a.prof[(size_t)b * 12 + i] = pt[i];
  // emit the container in sorted order
  for (int o = 0; o < n_out; o++) {
    const int phys = ord[o];
    if (lane == 0) a.out_e[(size_t)b * a.num_saved + o] = s_e[phys];
    for (int i = lane; i < nc; i += 64) a.out_conf[((size_t)b * a.num_saved + o) * cs_ + i] = s_conf[(size_t)phys * cs_ + i];
    for (int i = lane; i < 3 * nh; i += 64)
      a.out_coords[((size_t)b * a.num_saved + o) * xs_ + i] = s_xyz[(size_t)phys * xs_ + i];
  }
  if (lane == 0) {
    a.out_n[b] = n_out;
    if (a.evals) a.evals[b] = evals;
  }
}

// The two instantiations: latency tuning for teams of waves (one workgroup of W waves per chain), throughput tuning
// for one wave per chain with the register budget of two waves per SIMD.
template <bool PROF, bool ACC = false>
__global__ __launch_bounds__(256) void vina_mc_kernel(VinaEnv env, VinaLigand L, VinaMcArgs a) {
  mc_chain<PROF, false, ACC>(env, L, a);
}

#ifndef MI_MC_TP_WAVES
#define MI_MC_TP_WAVES 2  // waves per SIMD the throughput instantiation is register-allocated for (A/B: tools/experiments)
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MI_MC_TP_WAVES, MI_MC_TP_WAVES))) void vina_mc_tp_kernel(VinaEnv env, VinaLigand L,
                                                                                                   VinaMcArgs a) {
  mc_chain<false, true>(env, L, a);
}

// ---------------------------------------------------------------------------------------------
// Monte-Carlo with the CNN as the Metropolis energy: --cnn_scoring metrorescore / metrorefine
// (parallel_mc.cpp:145-155: mc(m, out, p, ig, ..., ig_metropolis = non_cache_cnn); monte_carlo.cpp:44-47 update_energy
// = adjust_center + ig_metropolis->eval).  The CNN scores whole batches, so all chains of a launch advance in lock
// step and stop where the reference calls update_energy: the host then evaluates non_cache_cnn::eval for every chain's
// `model` in ONE batch (mi_cnn_eval_batch) and resumes the chains.  Three resumable phases per step over the same
// device functions as vina_mc_kernel (one wave per chain; BFGS runs on the Vina grids like the reference's `ig`):
//   phase 0 / 2: [container insert of the previous step]  mutate -> BFGS(hunt cap)          -> model_out
//   phase 1    : Metropolis on the CNN energy -> [BFGS(full cap) if the pose is promising]     -> model_out
//   phase 3    : last container insert, emit
// Chain state between launches lives in global memory (st_f / st_i).
// ---------------------------------------------------------------------------------------------
template <bool ACC>  // (accurate_line_search / simple ascent: their own instantiation, like vina_mc_kernel's)
__global__ __launch_bounds__(64) void vina_mc_cnn_kernel(VinaEnv env, VinaLigand L, VinaMcArgs a, VinaMcCnnState st) {
  constexpr int HR = kHRegTp, PG = kPairGroupTp;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *pp = lds;
  if (env.stage) L = stage_ligand(L, pp);
  const int nt = L.n_nodes - 1, n = 6 + nt, nc = 7 + nt, nh = L.n_heavy;
  float *sh_f = carve(pp, 1);
  int *sh_ok = reinterpret_cast<int *>(carve(pp, 1));
  WaveWork w = carve_work(pp, L);
  BfgsWork k = carve_bfgs(pp, n, nc);
  float *tmp = carve(pp, nc);
  float *mconf = carve(pp, nc);
  float *hc = carve(pp, 3 * nh);
  float *rm = carve(pp, 64);
  int *ord = reinterpret_cast<int *>(carve(pp, a.num_saved));
  const WaveTeam tm{1, 0, 0, sh_f, sh_ok};
  const int b = blockIdx.x, lane = threadIdx.x & 63;
  float *sf = st.st_f + (size_t)b * st.f_stride;   // [tmp nc][cand nc][mconf nc][tmp_e, best_e]
  int *si = st.st_i + (size_t)b * st.i_stride;     // [n_out, rng idx, flag, -][ord num_saved]
  float *s_e = a.scratch_e + (size_t)b * a.num_saved;
  const int cs_ = a.conf_stride, xs_ = a.coord_stride;
  float *s_conf = a.scratch_conf + (size_t)b * a.num_saved * cs_;
  float *s_xyz = a.scratch_coords + (size_t)b * a.num_saved * xs_;
  McRng rng{a.mt + (size_t)b * a.mt_team * 624, st.phase == 0 ? 624 : si[1]};
  int evals = st.phase == 0 ? 0 : a.evals[b];
  float tmp_e = 0.f, best_e = VMAXFL;
  int n_out = 0, flag = 0;
  if (st.phase == 0) {
    randomize_wave(rng, tmp, a.c1, a.c2, nt);
    for (int i = lane; i < nc; i += 64) mconf[i] = i == 3 ? 1.f : 0.f;
  } else {
    for (int i = lane; i < nc; i += 64) tmp[i] = sf[i], mconf[i] = sf[2 * nc + i];
    tmp_e = sf[3 * nc], best_e = sf[3 * nc + 1];
    n_out = si[0], flag = si[2];
    for (int i = lane; i < a.num_saved; i += 64) ord[i] = si[4 + i];
  }
  wave_sync();
  if (st.phase == 1) {
    // metropolis_accept on the CNN energy of the candidate's `model` (monte_carlo.cpp:38-42,120-123)
    const float cand_e = st.ext_e[b];
    bool accept = st.step == 0 || cand_e < tmp_e;
    if (!accept) {
      const float prob = expf_ref((tmp_e - cand_e) / a.temperature);
      accept = rng.fl(0.f, 1.f) < prob;
    }
    flag = 1;  // rejected: `model` keeps the candidate's last evaluation (already in mconf)
    if (accept) {
      for (int i = lane; i < nc; i += 64) tmp[i] = mconf[i] = sf[nc + i];  // tmp = candidate; m.set(tmp.c)
      tmp_e = cand_e;
      flag = 2;
      wave_sync();
      if (tmp_e < best_e || n_out < a.num_saved) {  // refine with the full caps on the Vina grids (monte_carlo.cpp:128-131)
        for (int i = lane; i < nc; i += 64) k.x[i] = tmp[i];
        wave_sync();
        (void)bfgs_wave<HR, PG, ACC>(env, L, w, k, a.auth[0], a.auth[1], a.auth[2], a.max_iters, evals, tm, nullptr);
        for (int i = lane; i < nc; i += 64) tmp[i] = k.x[i], mconf[i] = k.x_new[i];
        flag = 3;
      }
    }
    wave_sync();
  } else {
    if (st.phase >= 2 && flag == 3) {
      // update_energy after the second minimisation, m.set(tmp.c), add_to_output_container (monte_carlo.cpp:131-139)
      tmp_e = st.ext_e[b];
      for (int i = lane; i < nc; i += 64) mconf[i] = tmp[i];
      wave_sync();
      (void)eval_conf<3, PG>(env, L, tmp, 0.f, 0.f, 0.f, w, nullptr);
      for (int h = lane; h < nh; h += 64) {
        const int i = L.heavy_list[h];
        hc[3 * h] = w.coords[3 * i], hc[3 * h + 1] = w.coords[3 * i + 1], hc[3 * h + 2] = w.coords[3 * i + 2];
      }
      wave_sync();
      n_out = container_insert(a, s_e, s_conf, s_xyz, cs_, xs_, ord, rm, tmp, hc, tmp_e, n_out, nc, nh);
      if (tmp_e < best_e) best_e = tmp_e;
    }
    flag = 0;
    if (st.phase != 3) {  // propose the next candidate: mutate -> BFGS with the hunt caps
      for (int i = lane; i < nc; i += 64) k.x[i] = tmp[i];
      wave_sync();
      mutate_wave<PG>(env, L, w, rng, k.x, mconf, a.amplitude);
      (void)bfgs_wave<HR, PG, ACC>(env, L, w, k, a.hunt[0], a.hunt[1], a.hunt[2], a.max_iters, evals, tm, nullptr);
      for (int i = lane; i < nc; i += 64) sf[nc + i] = k.x[i], mconf[i] = k.x_new[i];
      wave_sync();
    }
  }
  // what the CNN has to look at next: the conformation `model` holds
  for (int i = lane; i < nc; i += 64) {
    st.model_out[(size_t)b * nc + i] = mconf[i];
    sf[i] = tmp[i], sf[2 * nc + i] = mconf[i];
  }
  for (int i = lane; i < a.num_saved; i += 64) si[4 + i] = ord[i];
  if (lane == 0) {
    sf[3 * nc] = tmp_e, sf[3 * nc + 1] = best_e;
    si[0] = n_out, si[1] = rng.idx, si[2] = flag;
    a.evals[b] = evals;
  }
  if (st.phase == 3) {  // emit the container in sorted order
    wave_sync();
    for (int o = 0; o < n_out; o++) {
      const int phys = ord[o];
      if (lane == 0) a.out_e[(size_t)b * a.num_saved + o] = s_e[phys];
      for (int i = lane; i < nc; i += 64) a.out_conf[((size_t)b * a.num_saved + o) * cs_ + i] = s_conf[(size_t)phys * cs_ + i];
      for (int i = lane; i < 3 * nh; i += 64)
        a.out_coords[((size_t)b * a.num_saved + o) * xs_ + i] = s_xyz[(size_t)phys * xs_ + i];
    }
    if (lane == 0) a.out_n[b] = n_out;
  }
}

void launch_vina_mc_cnn(const VinaEnv &env0, const VinaLigand &lig, const VinaMcArgs &a0, const VinaMcCnnState &st, int B,
                        hipStream_t s) {
  VinaEnv env = env0;
  VinaMcArgs a = a0;
  env.stage = want_stage(B) ? 1 : 0;
  a.conf_stride = 7 + lig.n_nodes - 1;
  a.coord_stride = 3 * lig.n_heavy;
  const size_t lds = vina_mc_lds_bytes(lig.n_atoms, lig.n_nodes, lig.n_pairs, lig.n_heavy, a.num_saved, env.stage, 1);
  if (env.accurate_ls) {  // --accurate_line_search / --simple_ascent
    big_lds(vina_mc_cnn_kernel<true>);
    hipLaunchKernelGGL(vina_mc_cnn_kernel<true>, dim3(B), dim3(64), lds, s, env, lig, a, st);
  } else {
    big_lds(vina_mc_cnn_kernel<false>);
    hipLaunchKernelGGL(vina_mc_cnn_kernel<false>, dim3(B), dim3(64), lds, s, env, lig, a, st);
  }
}

// Waves per chain: a single docking job (few chains) is bound by the latency of dependent evaluations, so
// the line searches are spread over 4 waves; with many chains in flight the speculative trials would only take
// issue slots from other chains.
int vina_mc_team(int B) {
  if (const char *e = option(OPT_MI_VINA_MC_WAVES)) {  // experiments: force 1, 2 or 4
    const int w = atoi(e);
    if (w == 1 || w == 2 || w == 4) return w;
  }
  return B <= 256 ? 4 : B <= 512 ? 2 : 1;
}

size_t vina_mc_lds_bytes(int n_atoms, int n_nodes, int n_pairs, int n_heavy, int num_saved, bool stage, int W) {
  const size_t ligand = stage ? ligand_lds_floats(n_atoms, n_nodes, n_pairs, n_atoms) : 0;
  const size_t wave = vina_wave_lds_bytes(n_atoms, n_nodes, n_pairs, true, false) / sizeof(float) + 2 * pad4(7 + n_nodes - 1) +
                      pad4(3 * (size_t)n_heavy) + 64 + pad4(num_saved);
  return (ligand + 2 * pad4(W) + 4 + W * wave) * sizeof(float);
}

void launch_vina_mc(const VinaEnv &env0, const VinaLigand &lig, const VinaMcArgs &a0, int B, hipStream_t s) {
  VinaEnv env = env0;
  VinaMcArgs a = a0;
  env.stage = want_stage(B) ? 1 : 0;
  if (!a.ligs) {
    a.conf_stride = 7 + lig.n_nodes - 1;
    a.coord_stride = 3 * lig.n_heavy;
  }
  // `lig` sizes the workspace (screen mode: a synthetic description holding the maxima of the set)
  int W = vina_mc_team(B);
  size_t lds = vina_mc_lds_bytes(lig.n_atoms, lig.n_nodes, lig.n_pairs, lig.n_heavy, a.num_saved, env.stage, W);
  while (W > 1 && lds > kVinaMaxLds - 8 * 1024) {  // very large ligands: fewer waves per chain
    W /= 2;
    lds = vina_mc_lds_bytes(lig.n_atoms, lig.n_nodes, lig.n_pairs, lig.n_heavy, a.num_saved, env.stage, W);
  }
  if (env.accurate_ls) {  // --accurate_line_search: the team kernel at any chain count (sequential trials)
    big_lds((vina_mc_kernel<false, true>));
    hipLaunchKernelGGL((vina_mc_kernel<false, true>), dim3(B), dim3(64 * W), lds, s, env, lig, a);
  } else if (a.prof) {
    big_lds(vina_mc_kernel<true>);
    hipLaunchKernelGGL(vina_mc_kernel<true>, dim3(B), dim3(64 * W), lds, s, env, lig, a);
  } else if (W == 1) {
    big_lds(vina_mc_tp_kernel);
    hipLaunchKernelGGL(vina_mc_tp_kernel, dim3(B), dim3(64), lds, s, env, lig, a);
  } else {
    big_lds(vina_mc_kernel<false>);
    hipLaunchKernelGGL(vina_mc_kernel<false>, dim3(B), dim3(64 * W), lds, s, env, lig, a);
  }
}

// ---------------------------------------------------------------------------------------------
// cache::populate (cache.cpp:104-184): one thread per grid point, receptor atoms staged through
// LDS in tiles; per point the sum runs over receptor atoms in index order like the reference's
// index-ordered `possibilities` list, so the fp32 result is the same sum.  An atom counts only if it is on the
// candidate list of the point's 3 A cell, i.e. within the cut-off of the cell's brick as szv_grid_cache::get
// (szv_grid.h:107-144) built it (a.brick; degenerate bricks included: oracle/_ref reproduces them).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vina_populate_kernel(VinaPopulateArgs a) {
  __shared__ float4 tile[256];
  const long npts = (long)a.geom.dim[0] * a.geom.dim[1] * a.geom.dim[2];
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = idx < npts;
  const long ii = live ? idx : 0;
  const int x = (int)(ii % a.geom.dim[0]), y = (int)((ii / a.geom.dim[0]) % a.geom.dim[1]),
            z = (int)(ii / ((long)a.geom.dim[0] * a.geom.dim[1]));
  const float px = a.geom.init[0] + a.geom.factor_inv[0] * (float)x;  // index_to_argument, grid.h:54-57
  const float py = a.geom.init[1] + a.geom.factor_inv[1] * (float)y;
  const float pz = a.geom.init[2] + a.geom.factor_inv[2] * (float)z;
  const float2 bx = a.brick[x], by = a.brick[a.geom.dim[0] + y], bz = a.brick[a.geom.dim[0] + a.geom.dim[1] + z];
  float aff = 0.f;
  for (int base = 0; base < a.n_rec; base += 256) {
    __syncthreads();
    if (base + (int)threadIdx.x < a.n_rec) tile[threadIdx.x] = a.rec[base + threadIdx.x];
    __syncthreads();
    const int cnt = min(256, a.n_rec - base);
    for (int j = 0; j < cnt; j++) {
      const float4 r = tile[j];
      const float dx = r.x - px, dy = r.y - py, dz = r.z - pz;
      const float r2 = dx * dx + dy * dy + dz * dz;
      // brick_distance_sqr (brick.h:37-49): closest point of the cell's brick to the atom
      const float cx = fminf(fmaxf(r.x, bx.x), bx.y) - r.x, cy = fminf(fmaxf(r.y, by.x), by.y) - r.y,
                  cz = fminf(fmaxf(r.z, bz.x), bz.y) - r.z;
      if (r2 <= a.cutoff_sqr && cx * cx + cy * cy + cz * cz < a.cutoff_sqr) {
        const int t1 = __float_as_int(r.w);
        if (a.spline) {
          float e, dx;
          spline_eval(a.spline, a.sp_n, a.sp_fraction, a.cutoff, t1, a.lig_type, sqrtf(r2), e, dx);
          aff += e;
        } else {
          aff += a.fast[(long)tri_idx(t1, a.lig_type) * a.n + (int)(a.factor * r2)];
        }
      }
    }
  }
  // cache.cpp:177-179: `user_grid.evaluate_user(vec(x, y, z), slope)` -- the lattice INDICES go in as the location
  // (not the point's coordinates); reproduced as the reference computes it
  if (a.ug_data) {
    float ux, uy, uz;
    aff += grid_evaluate<false>(a.ug_geom, a.ug_data, (float)x, (float)y, (float)z, a.ug_slope, 1000.f, ux, uy, uz);
  }
  if (live) a.out[idx] = aff;
}

// diagnostic: sincos_ref on n arguments (mi_debug_sincos; the -m gpu test compares it with the host's libm)
__global__ void vina_sincos_probe_kernel(const float *x, int n, float *sn, float *cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sincos_ref(x[i], sn[i], cs[i]);
}
void launch_vina_sincos_probe(const float *x, int n, float *sn, float *cs, hipStream_t s) {
  hipLaunchKernelGGL(vina_sincos_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, n, sn, cs);
}
__global__ void vina_explog_probe_kernel(const float *x, int n, float *ex, float *lg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ex[i] = expf_ref(x[i]), lg[i] = logf_ref(fabsf(x[i]));
}
__global__ void vina_acos_probe_kernel(const float *x, int n, float *ac) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ac[i] = acosf_ref(x[i]);
}
void launch_vina_acos_probe(const float *x, int n, float *ac, hipStream_t s) {
  hipLaunchKernelGGL(vina_acos_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, n, ac);
}
void launch_vina_explog_probe(const float *x, int n, float *ex, float *lg, hipStream_t s) {
  hipLaunchKernelGGL(vina_explog_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, n, ex, lg);
}

void launch_vina_populate(const VinaPopulateArgs &a, hipStream_t s) {
  const long npts = (long)a.geom.dim[0] * a.geom.dim[1] * a.geom.dim[2];
  hipLaunchKernelGGL(vina_populate_kernel, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, s, a);
}

void launch_vina_eval(const VinaEnv &env, const VinaLigand &lig, const float *confs, int B, float v0, float v1,
                      float v2, int with_deriv, float *energy, float *change, float *coords, hipStream_t s) {
  const size_t lds = vina_wave_lds_bytes(lig.n_atoms, lig.n_nodes, lig.n_pairs, false, false);
  // with_deriv: 1 = model::eval_deriv, 0 = model::eval, 2 = cache::eval (grid term only), 4 = eval_intramolecular,
  // 5 = the pair part of model::eval
  auto go = [&](auto kernel) {
    big_lds(kernel);
    hipLaunchKernelGGL(kernel, dim3(B), dim3(64), lds, s, env, lig, confs, v0, v1, v2, energy, change, coords);
  };
  if (with_deriv == 1)
    go(vina_eval_kernel<0>);
  else if (with_deriv == 0)
    go(vina_eval_kernel<1>);
  else if (with_deriv == 4)
    go(vina_eval_kernel<4>);
  else if (with_deriv == 5)
    go(vina_eval_kernel<5>);
  else
    go(vina_eval_kernel<2>);
}

void launch_vina_bfgs(const VinaEnv &env0, const VinaLigand &lig, float *confs, int B, float v0, float v1, float v2,
                      int max_iters, float *energy, float *grad, int *evals, hipStream_t s) {
  VinaEnv env = env0;
  env.stage = want_stage(B) ? 1 : 0;
  const size_t lds = vina_wave_lds_bytes(lig.n_atoms, lig.n_nodes, lig.n_pairs, true, env.stage);
  if (env.accurate_ls) {
    big_lds((vina_bfgs_kernel<false, true>));
    hipLaunchKernelGGL((vina_bfgs_kernel<false, true>), dim3(B), dim3(64), lds, s, env, lig, confs, v0, v1, v2, max_iters,
                       energy, grad, evals);
  } else if (B > 2048) {  // thousands of chains: occupancy over per-chain latency
    big_lds(vina_bfgs_kernel<true>);
    hipLaunchKernelGGL(vina_bfgs_kernel<true>, dim3(B), dim3(64), lds, s, env, lig, confs, v0, v1, v2, max_iters, energy,
                       grad, evals);
  } else {
    big_lds(vina_bfgs_kernel<false>);
    hipLaunchKernelGGL(vina_bfgs_kernel<false>, dim3(B), dim3(64), lds, s, env, lig, confs, v0, v1, v2, max_iters, energy,
                       grad, evals);
  }
}

}  // namespace mig
