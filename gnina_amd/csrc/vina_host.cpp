// vina_host.cpp -- host side of the Vina engine: pair tables (precalculate_linear), receptor /
// cache / ligand upload, and the mi_vina_* C ABI (include/mi_gnina.h).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/mi_gnina.h"
#include "common.h"
#include "typer.h"
#include "vina.h"
#include "options.h"

namespace mig {

// atom_constants.h:101-133 (xs_hydrophobe / xs_donor / xs_acceptor columns)
static const unsigned char kHyd[kVinaTypes] = {0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                               0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1};
static const unsigned char kDon[kVinaTypes] = {0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 1, 1, 0,
                                               0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 0};
static const unsigned char kAcc[kVinaTypes] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 1, 1,
                                               0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
// main.cpp:1324-1328
static const float kDefaultWeights[5] = {-0.035579f, -0.005156f, 0.840245f, -0.035069f, -0.587439f};

static float slope_step(float x_bad, float x_good, float x) {  // everything.h:207-216
  if (x_bad < x_good) {
    if (x <= x_bad) return 0;
    if (x >= x_good) return 1;
  } else {
    if (x >= x_bad) return 0;
    if (x <= x_good) return 1;
  }
  return (x - x_bad) / (x_good - x_bad);
}

static float gaussian(float x, float width) {  // everything.h:48-50
  volatile float q = x / width;
  return expf(-(q * q));
}

// weighted_terms::eval_fast (weighted_terms.cpp:54-68) for the default Vina term set
static float pair_energy(const float *w, int t1, int t2, float r) {
  volatile float opt = smina_xs_radius(t1) + smina_xs_radius(t2);
  volatile float acc = 0;
  acc = acc + w[0] * gaussian(r - (opt + 0.0f), 0.5f);
  acc = acc + w[1] * gaussian(r - (opt + 3.0f), 2.0f);
  {
    volatile float d = r - (opt + 0.0f);
    acc = acc + w[2] * (d > 0 ? 0.0f : d * d);
  }
  acc = acc + w[3] * ((kHyd[t1] && kHyd[t2]) ? slope_step(1.5f, 0.5f, r - opt) : 0.0f);
  const bool hb = (kDon[t1] && kAcc[t2]) || (kDon[t2] && kAcc[t1]);
  acc = acc + w[4] * (hb ? slope_step(0.0f, -0.7f, r - opt) : 0.0f);
  return acc;
}

// one ligand's description on the device (arrays + the struct of pointers the kernels take)
struct LigandDev {
  VinaLigand lig{};
  std::vector<int32_t> h_smt;
  DevBuf<int> d_int;
  DevBuf<float> d_flt;
};

struct Vina {
  hipStream_t stream = nullptr;
  // tables (host copies kept for mi_vina_table)
  int n = 0;
  float factor = 32.f, cutoff_sqr = 64.f;
  std::vector<float> h_fast, h_se, h_sd;
  DevBuf<float2> d_smooth;
  DevBuf<float> d_fast;
  // receptor
  DevBuf<float4> d_rec;
  DevBuf<float2> d_brick;
  int n_rec = 0;
  // cache
  bool have_cache = false;
  VinaGridGeom geom{};
  long grid_off[kVinaTypes];
  size_t grid_pts = 0;
  DevBuf<float> d_grids;
  float slope = 1e3f;
  float box_begin[3] = {0, 0, 0}, box_end[3] = {0, 0, 0};
  // precalculate_splines (--approximation spline; the default of --minimize, main.cpp:1162-1165)
  bool use_spline = false;
  int sp_n = 0;
  float sp_fraction = 0, cutoff = 8;
  DevBuf<float4> d_spline;
  std::vector<float4> h_spline;
  bool accurate_ls = false;  // --accurate_line_search (also set for --simple_ascent, which uses the same search)
  bool simple_ascent = false;  // minimization_params::Simple: steepest descent, no quasi-Newton update
  bool strict = false;       // mi_vina_set_strict_order: energy sums in the reference's order
  // --user_grid
  bool have_ug = false;
  VinaGridGeom ug_geom{};
  DevBuf<float> d_ug;
  float w5[5];
  // ligand
  bool have_lig = false;
  VinaLigand lig{};
  std::vector<int32_t> h_lig_smt;  // host copy of the ligand atom types (CNN-in-the-loop calls type the ligand with them)
  LigandDev one;                   // storage behind `lig` (mi_vina_set_ligand)
  // screen: many ligands resident at once, docked by one launch (mi_vina_set_screen / mi_vina_mc_screen)
  std::vector<std::unique_ptr<LigandDev>> screen;
  DevBuf<VinaLigand> d_screen;
  DevBuf<int> d_chain_lig, d_lig_steps, d_lig_iters, d_chain_order;
  // scratch
  DevBuf<float> d_confs, d_energy, d_change, d_coords;
  DevBuf<float> d_ext_forces, d_ext_e, d_ext_centers;
  DevBuf<int> d_evals, d_out_n, d_smt_in;
  DevBuf<unsigned long long> d_seeds;
  DevBuf<unsigned> d_mt;
  DevBuf<float> d_mc_e, d_mc_conf, d_mc_xyz, d_sc_e, d_sc_conf, d_sc_xyz;
  ~Vina() {
    if (stream) (void)hipStreamDestroy(stream);
  }
};

static int tri(int t1, int t2) {
  if (t1 > t2) std::swap(t1, t2);
  return t1 + t2 * (t2 + 1) / 2;
}

// precalculate_linear ctor (precalculate.h:184-210) + init_from_smooth_fst (:135-158)
static void build_tables(Vina &v, const float *w, float cutoff, float factor) {
  v.factor = factor;
  v.cutoff_sqr = cutoff * cutoff;
  v.n = (int)(factor * v.cutoff_sqr) + 3;
  const int n = v.n, np = kVinaTypes * (kVinaTypes + 1) / 2;
  std::vector<float> rs(n + 2);
  for (int i = 0; i < n + 2; i++) rs[i] = sqrtf((float)i / factor);
  v.h_fast.assign((size_t)np * n, 0.f);
  v.h_se.assign((size_t)np * n, 0.f);
  v.h_sd.assign((size_t)np * n, 0.f);
  for (int t1 = 0; t1 < kVinaTypes; t1++)
    for (int t2 = t1; t2 < kVinaTypes; t2++) {
      float *se = &v.h_se[(size_t)tri(t1, t2) * n], *sd = &v.h_sd[(size_t)tri(t1, t2) * n];
      float *fa = &v.h_fast[(size_t)tri(t1, t2) * n];
      for (int i = 0; i < n; i++) se[i] = pair_energy(w, t1, t2, rs[i]);
      for (int i = 0; i < n; i++) {
        if (i == 0 || i == n - 1) {
          sd[i] = 0;
        } else {
          volatile float delta = rs[i + 1] - rs[i - 1];
          volatile float dr = delta * rs[i];
          sd[i] = (se[i + 1] - se[i - 1]) / dr;
        }
        const float f2 = (i + 1 >= n) ? 0.f : se[i + 1];
        volatile float s2 = f2 + se[i];
        fa[i] = s2 / 2;
      }
    }
  std::vector<float2> sm((size_t)np * n);
  for (size_t i = 0; i < sm.size(); i++) sm[i] = make_float2(v.h_se[i], v.h_sd[i]);
  v.d_smooth.upload(sm.data(), sm.size(), v.stream);
  v.d_fast.upload(v.h_fast.data(), v.h_fast.size(), v.stream);
  MIG_HIP(hipStreamSynchronize(v.stream));
}

// precalculate_splines(sf, factor) (precalculate.h:277-449) + Spline::initialize (splines.h:36-96): per type pair
// n = unsigned(factor * cutoff) samples of E(r) at r = i * (cutoff / n) plus (cutoff, 0); a cubic spline with zero first
// derivative at both ends through them.  The reference solves the (n + 1) x (n + 1) system by inverting the dense
// matrix in fp32 with Eigen (a third-party header library); the system is tridiagonal and diagonally dominant, here
// it is solved directly (Thomas algorithm, double accumulation), coefficients rounded to fp32 in the reference's
// formulas -- same spline to ~1e-6 of its scale, not bit-identical (stated in DESIGN.md §4).
static void build_splines(Vina &v, float cutoff, float factor) {
  const unsigned n = (unsigned)(factor * cutoff);
  MIG_CHECK(n >= 2 && n < 65536, 1, "spline approximation: factor * cutoff must give 2 .. 65535 intervals");
  const float fraction = cutoff / (float)n;
  const int np = kVinaTypes * (kVinaTypes + 1) / 2;
  v.h_spline.assign((size_t)np * n, make_float4(0.f, 0.f, 0.f, 0.f));
  std::vector<float> x(n + 1), y(n + 1), C(n + 1), ddy(n + 1);
  std::vector<double> cp(n + 1), dp(n + 1);
  for (unsigned i = 0; i < n; i++) x[i] = (float)i * fraction;
  x[n] = cutoff;
  const float hlast = x[n] - x[n - 1];
  for (int t1 = 0; t1 < kVinaTypes; t1++)
    for (int t2 = t1; t2 < kVinaTypes; t2++) {
      bool nonzero = false;
      for (unsigned i = 0; i < n; i++) {
        y[i] = pair_energy(v.w5, t1, t2, x[i]);
        nonzero = nonzero || y[i] != 0;
      }
      y[n] = 0;
      if (!nonzero) continue;  // "worth interpolating" (precalculate.h:361): an uninitialised Spline evaluates to 0
      const unsigned e = n;
      // rows: sub[i] ddy[i-1] + diag[i] ddy[i] + sup[i] ddy[i+1] = C[i]
      auto hi_of = [&](unsigned i) { return i == e - 1 ? hlast : fraction; };
      C[0] = 6 * ((y[1] - y[0]) / fraction);
      for (unsigned i = 1; i < e; i++) {
        const float hi = hi_of(i);
        C[i] = 6 * ((y[i + 1] - y[i]) / hi - (y[i] - y[i - 1]) / fraction);
      }
      C[e] = 6 * (-(y[e] - y[e - 1]) / hlast);
      auto diag = [&](unsigned i) -> double { return i == 0 ? 2.0 * fraction : i == e ? 2.0 * hlast : 2.0 * ((double)fraction + hi_of(i)); };
      auto sub = [&](unsigned i) -> double { return i == e ? hlast : hi_of(i); };   // coefficient of ddy[i-1] in row i
      auto sup = [&](unsigned i) -> double { return i == 0 ? fraction : hi_of(i); };  // coefficient of ddy[i+1] in row i
      cp[0] = sup(0) / diag(0);
      dp[0] = C[0] / diag(0);
      for (unsigned i = 1; i <= e; i++) {
        const double m = diag(i) - sub(i) * cp[i - 1];
        cp[i] = i < e ? sup(i) / m : 0.0;
        dp[i] = (C[i] - sub(i) * dp[i - 1]) / m;
      }
      ddy[e] = (float)dp[e];
      double nxt = dp[e];
      for (int i = (int)e - 1; i >= 0; i--) {
        nxt = dp[i] - cp[i] * nxt;
        ddy[i] = (float)nxt;
      }
      float4 *out = &v.h_spline[(size_t)tri(t1, t2) * n];
      for (unsigned i = 0; i < e; i++) {
        const float hi = hi_of(i);
        out[i].x = (ddy[i + 1] - ddy[i]) / (6 * hi);
        out[i].y = ddy[i] / 2;
        out[i].z = (y[i + 1] - y[i]) / hi - ddy[i + 1] * hi / 6 - ddy[i] * hi / 3;
        out[i].w = y[i];
      }
    }
  v.sp_n = (int)n;
  v.sp_fraction = fraction;
  v.cutoff = cutoff;
  v.d_spline.upload(v.h_spline.data(), v.h_spline.size(), v.stream);
  MIG_HIP(hipStreamSynchronize(v.stream));
}

static VinaEnv make_env(const Vina &v) {
  VinaEnv e{};
  e.smooth = v.d_smooth.p;
  e.fast = v.d_fast.p;
  e.n = v.n;
  e.factor = v.factor;
  e.cutoff_sqr = v.cutoff_sqr;
  e.geom = v.geom;
  e.grid_data = v.d_grids.p;
  for (int t = 0; t < kVinaTypes; t++) e.grid_off[t] = v.grid_off[t];
  e.slope = v.slope;
  e.direct = 0;
  e.exact = 0;
  e.stage = 0;
  e.rec = v.d_rec.p;
  e.n_rec = v.n_rec;
  for (int i = 0; i < 5; i++) e.w5[i] = v.w5[i];
  for (int i = 0; i < 3; i++) {
    e.box_begin[i] = v.box_begin[i];
    e.box_end[i] = v.box_end[i];
  }
  e.ug_geom = v.ug_geom;
  e.ug_data = v.have_ug ? v.d_ug.p : nullptr;
  e.accurate_ls = v.simple_ascent ? 2 : v.accurate_ls ? 1 : 0;
  e.strict = v.strict ? 1 : 0;
  e.spline = v.use_spline ? v.d_spline.p : nullptr;
  e.sp_n = v.sp_n;
  e.sp_fraction = v.sp_fraction;
  e.cutoff = v.cutoff;
  return e;
}

}  // namespace mig

using namespace mig;

#define VTRY try {
#define VCATCH_STATUS                        \
  }                                          \
  catch (const mig::Error &e) {              \
    mig::set_last_error(e.what());           \
    return e.code;                           \
  }                                          \
  catch (const std::exception &e) {          \
    mig::set_last_error(e.what());           \
    return MI_ERR_INVALID;                   \
  }

extern "C" {

mi_vina *mi_vina_create(const float *weights5, float cutoff, float factor) {
  try {
    MIG_CHECK(cutoff > 0 && factor > 0 && factor * cutoff * cutoff < 1e6f, 1, "bad cutoff / factor");
    std::unique_ptr<Vina> v(new Vina());
    MIG_HIP(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
    for (auto &o : v->grid_off) o = -1;
    for (int i = 0; i < 5; i++) v->w5[i] = (weights5 ? weights5 : kDefaultWeights)[i];
    build_tables(*v, v->w5, cutoff, factor);
    return reinterpret_cast<mi_vina *>(v.release());
  } catch (const std::exception &e) {
    mig::set_last_error(e.what());
    return nullptr;
  }
}

void mi_vina_destroy(mi_vina *v) { delete reinterpret_cast<Vina *>(v); }

int mi_vina_table_size(const mi_vina *v) { return v ? reinterpret_cast<const Vina *>(v)->n : 0; }

mi_status mi_vina_table(const mi_vina *vv, int t1, int t2, float *fast, float *smooth_e, float *smooth_dor) {
  VTRY
  MIG_CHECK(vv && t1 >= 0 && t1 < kVinaTypes && t2 >= 0 && t2 < kVinaTypes, 1, "bad arguments");
  const Vina &v = *reinterpret_cast<const Vina *>(vv);
  const size_t o = (size_t)tri(t1, t2) * v.n;
  if (fast) std::memcpy(fast, &v.h_fast[o], sizeof(float) * v.n);
  if (smooth_e) std::memcpy(smooth_e, &v.h_se[o], sizeof(float) * v.n);
  if (smooth_dor) std::memcpy(smooth_dor, &v.h_sd[o], sizeof(float) * v.n);
  return MI_OK;
  VCATCH_STATUS
}

mi_status mi_vina_set_receptor(mi_vina *vv, const float *xyz, const int32_t *smt, int n) {
  VTRY
  MIG_CHECK(vv && n >= 0 && (n == 0 || (xyz && smt)), 1, "bad receptor arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  // Receptor hydrogens are never interaction partners: cache::populate, non_cache::eval / eval_deriv and
  // non_cache_cnn's empirical term only see the atoms szv_grid hands out, and szv_grid drops hydrogens
  // (szv_grid.h:69-73,126-131: `!a.is_hydrogen() && a.acceptable_type()`).  Order of the rest is kept.
  std::vector<float4> rec;
  rec.reserve(n);
  for (int i = 0; i < n; i++) {
    MIG_CHECK(smt[i] >= 0 && smt[i] < kVinaTypes, 1, "receptor smina type out of range");
    if (smt[i] <= 1) continue;  // Hydrogen, PolarHydrogen (atom_constants.h:101-104)
    float w;
    int32_t t = smt[i];
    std::memcpy(&w, &t, 4);
    rec.push_back(make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], w));
  }
  v.d_rec.upload(rec.data(), rec.size(), v.stream);
  MIG_HIP(hipStreamSynchronize(v.stream));
  v.n_rec = (int)rec.size();
  v.have_cache = false;
  return MI_OK;
  VCATCH_STATUS
}

mi_status mi_vina_build_cache(mi_vina *vv, const float *begin3, const float *end3, const int32_t *n3,
                              const int32_t *lig_types, int n_types, float slope) {
  VTRY
  MIG_CHECK(vv && begin3 && end3 && n3 && lig_types && n_types > 0, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.n_rec > 0, 4, "mi_vina_set_receptor must be called first");
  for (int i = 0; i < 3; i++) {  // grid::init, grid.cpp:47-68
    MIG_CHECK(n3[i] > 0 && end3[i] > begin3[i], 1, "bad grid dims");
    v.geom.dim[i] = n3[i] + 1;
    v.geom.init[i] = begin3[i];
    v.box_begin[i] = begin3[i];
    v.box_end[i] = end3[i];
    const float range = end3[i] - begin3[i];
    v.geom.dim_m1[i] = (float)v.geom.dim[i] - 1.0f;
    v.geom.factor[i] = v.geom.dim_m1[i] / range;
    v.geom.factor_inv[i] = 1 / v.geom.factor[i];
  }
  v.grid_pts = (size_t)v.geom.dim[0] * v.geom.dim[1] * v.geom.dim[2];
  for (auto &o : v.grid_off) o = -1;
  int cnt = 0;
  for (int j = 0; j < n_types; j++) {
    const int t = lig_types[j];
    MIG_CHECK(t >= 0 && t < kVinaTypes, 1, "ligand smina type out of range");
    if (v.grid_off[t] < 0) v.grid_off[t] = (long)(cnt++) * (long)v.grid_pts;
  }
  v.d_grids.ensure((size_t)cnt * v.grid_pts);
  v.slope = slope;
  // szv_grid_cache::get (szv_grid.h:107-144) builds the candidate list of a 3 A cell the first time a point falls
  // into it, from the brick [floor(c/3)*3, ceil(c/3)*3] of THAT point -- a plane or a line when a coordinate is an
  // exact multiple of 3 -- and reuses it for the cell's later points.  cache::populate walks x, y, z upwards, so the
  // first point of a cell is its lowest lattice point per dimension.  Reproduced so that grids match the reference's
  // bit for bit on boxes aligned with that lattice (oracle/_ref; tests/test_ref_vina.py).
  std::vector<float2> brick;
  for (int d = 0; d < 3; d++) {
    const float gran = 3.0f;
    float cur_cell = 0, lo = 0, hi = 0;
    for (int i = 0; i < v.geom.dim[d]; i++) {
      const float c = v.geom.init[d] + v.geom.factor_inv[d] * (float)i;
      const float cell = std::floor(c / gran);
      if (i == 0 || cell != cur_cell) {
        cur_cell = cell;
        lo = std::floor(c / gran) * gran;
        hi = std::ceil(c / gran) * gran;
      }
      brick.push_back(make_float2(lo, hi));
    }
  }
  v.d_brick.upload(brick.data(), brick.size(), v.stream);
  for (int t = 0; t < kVinaTypes; t++) {
    if (v.grid_off[t] < 0) continue;
    VinaPopulateArgs a{};
    a.rec = v.d_rec.p;
    a.n_rec = v.n_rec;
    a.fast = v.d_fast.p;
    a.n = v.n;
    a.factor = v.factor;
    a.cutoff_sqr = v.cutoff_sqr;
    a.geom = v.geom;
    a.lig_type = t;
    a.out = v.d_grids.p + v.grid_off[t];
    a.brick = v.d_brick.p;
    a.ug_geom = v.ug_geom;
    a.ug_data = v.have_ug ? v.d_ug.p : nullptr;
    a.ug_slope = slope;
    a.spline = v.use_spline ? v.d_spline.p : nullptr;
    a.sp_n = v.sp_n;
    a.sp_fraction = v.sp_fraction;
    a.cutoff = v.cutoff;
    launch_vina_populate(a, v.stream);
  }
  MIG_HIP(hipGetLastError());
  MIG_HIP(hipStreamSynchronize(v.stream));
  v.have_cache = true;
  return MI_OK;
  VCATCH_STATUS
}

// --accurate_line_search (minimization_params::BFGSAccurateLineSearch, main.cpp / bfgs.h:395-400): which line search
// every BFGS of this handle runs from now on -- mi_vina_bfgs_batch, the refine entry points, the Monte-Carlo chains
// and the CNN refinement's host state machine.
mi_status mi_vina_set_line_search(mi_vina *vv, int kind) {
  VTRY
  MIG_CHECK(vv && kind >= 0 && kind <= 2, 1, "minimiser: 0 (bfgs, fast line search), 1 (bfgs, accurate line search) or 2 (simple ascent)");
  reinterpret_cast<Vina *>(vv)->accurate_ls = kind >= 1;
  reinterpret_cast<Vina *>(vv)->simple_ascent = kind == 2;
  return MI_OK;
  VCATCH_STATUS
}

// Diagnostic: the device's sinf / cosf (vina.hip sincos_ref, glibc's algorithm restated) on n host arguments.
mi_status mi_debug_sincos(const float *x, int n, float *sn, float *cs) {
  VTRY
  MIG_CHECK(x && sn && cs && n >= 0, 1, "bad arguments");
  if (n == 0) return MI_OK;
  DevBuf<float> dx, ds, dc;
  dx.upload(x, (size_t)n, nullptr);
  ds.ensure((size_t)n);
  dc.ensure((size_t)n);
  launch_vina_sincos_probe(dx.p, n, ds.p, dc.p, nullptr);
  MIG_HIP(hipMemcpy(sn, ds.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  MIG_HIP(hipMemcpy(cs, dc.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return MI_OK;
  VCATCH_STATUS
}

// Diagnostic: ex[i] = the device's expf(x[i]), lg[i] = its logf(|x[i]|) (vina.hip expf_ref / logf_ref).
mi_status mi_debug_explog(const float *x, int n, float *ex, float *lg) {
  VTRY
  MIG_CHECK(x && ex && lg && n >= 0, 1, "bad arguments");
  if (n == 0) return MI_OK;
  DevBuf<float> dx, de, dl;
  dx.upload(x, (size_t)n, nullptr);
  de.ensure((size_t)n);
  dl.ensure((size_t)n);
  launch_vina_explog_probe(dx.p, n, de.p, dl.p, nullptr);
  MIG_HIP(hipMemcpy(ex, de.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  MIG_HIP(hipMemcpy(lg, dl.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return MI_OK;
  VCATCH_STATUS
}

// Diagnostic: ac[i] = the device's acosf(x[i]) (vina.hip acosf_ref).
mi_status mi_debug_acos(const float *x, int n, float *ac) {
  VTRY
  MIG_CHECK(x && ac && n >= 0, 1, "bad arguments");
  if (n == 0) return MI_OK;
  DevBuf<float> dx, da;
  dx.upload(x, (size_t)n, nullptr);
  da.ensure((size_t)n);
  launch_vina_acos_probe(dx.p, n, da.p, nullptr);
  MIG_HIP(hipMemcpy(ac, da.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return MI_OK;
  VCATCH_STATUS
}

// Energy sums of every evaluation of this handle in the reference's order (vina.hip seq_add) instead of the DPP
// butterfly: with it, energies / gradients / BFGS and Monte-Carlo trajectories are bit-identical to the reference's
// (tests/test_gpu_vina_ref.py); ~1 us per evaluation, so a mode.
mi_status mi_vina_set_strict_order(mi_vina *vv, int on) {
  VTRY
  MIG_CHECK(vv && (on == 0 || on == 1), 1, "strict order: 0 or 1");
  reinterpret_cast<Vina *>(vv)->strict = on == 1;
  return MI_OK;
  VCATCH_STATUS
}

// --approximation (main.cpp:989,1384-1391): which precalculate the handle evaluates pair terms with from now on.
mi_status mi_vina_set_approximation(mi_vina *vv, int kind, float factor) {
  VTRY
  MIG_CHECK(vv, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(kind == MI_VINA_APPROX_LINEAR || kind == MI_VINA_APPROX_SPLINE, 1, "approximation: 0 (linear) or 1 (spline)");
  if (kind == MI_VINA_APPROX_LINEAR) {
    if (v.use_spline) v.have_cache = false;  // the lattice was populated from spline energies
    v.use_spline = false;
    return MI_OK;
  }
  MIG_CHECK(factor > 1.1920928955078125e-07f, 1, "approximation factor must be positive");
  build_splines(v, std::sqrt(v.cutoff_sqr), factor);
  v.use_spline = true;
  v.have_cache = false;  // grids of the other approximation (or of another spline factor) are not this one's
  return MI_OK;
  VCATCH_STATUS
}

// precalculate::eval_deriv of the handle's current approximation for one type pair (host copies of the tables /
// spline coefficients): (E, dE/dr / r) at squared distances r2 -- what the kernels look up.
mi_status mi_vina_pair_eval(mi_vina *vv, int t1, int t2, const float *r2, int n, float *e, float *dor) {
  VTRY
  MIG_CHECK(vv && r2 && e && dor && n >= 0 && t1 >= 0 && t1 < kVinaTypes && t2 >= 0 && t2 < kVinaTypes, 1, "bad arguments");
  const Vina &v = *reinterpret_cast<const Vina *>(vv);
  for (int i = 0; i < n; i++) {
    MIG_CHECK(r2[i] >= 0 && (v.use_spline || r2[i] <= v.cutoff_sqr), 1, "r2 outside the table");
    if (v.use_spline) {
      const float r = sqrtf(r2[i]);
      e[i] = dor[i] = 0;
      if (r >= v.cutoff) continue;  // Spline::eval_deriv, splines.h:100-118; dor = dx / r (precalculate.h:440)
      int idx = (int)(r / v.sp_fraction);
      if (idx > v.sp_n - 1) idx = v.sp_n - 1;
      const float4 k = v.h_spline[(size_t)tri(t1, t2) * v.sp_n + idx];
      const float lx = r - (float)idx * v.sp_fraction;
      e[i] = ((k.x * lx + k.y) * lx + k.z) * lx + k.w;
      dor[i] = ((3 * k.x * lx + 2 * k.y) * lx + k.z) / r;
    } else {  // precalculate_linear_element::eval_deriv, precalculate.h:97-133
      const float r2f = v.factor * r2[i];
      const int i1 = (int)r2f;
      const float rem = r2f - (float)i1;
      const float *se = &v.h_se[(size_t)tri(t1, t2) * v.n], *sd = &v.h_sd[(size_t)tri(t1, t2) * v.n];
      e[i] = se[i1] + rem * (se[i1 + 1] - se[i1]);
      dor[i] = sd[i1] + rem * (sd[i1 + 1] - sd[i1]);
    }
  }
  return MI_OK;
  VCATCH_STATUS
}

// grid::init(gd, user_in, ug_scaling_factor), grid.cpp:69-92: (n + 1)^3 points of which the file fills [0, n)^3 in
// z, y, x order (x fastest) with -(value * scale); the last plane of every dimension stays 0.
mi_status mi_vina_set_user_grid(mi_vina *vv, const float *begin3, const float *end3, const int32_t *n3,
                                const double *values, float scaling_factor) {
  VTRY
  MIG_CHECK(vv, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  if (!values) {  // no user grid
    v.have_ug = false;
    return MI_OK;
  }
  MIG_CHECK(begin3 && end3 && n3, 1, "bad arguments");
  VinaGridGeom g{};
  for (int i = 0; i < 3; i++) {
    MIG_CHECK(n3[i] > 0 && n3[i] < 4096 && end3[i] > begin3[i], 1, "bad user grid dims");
    g.dim[i] = n3[i] + 1;
    g.init[i] = begin3[i];
    const float range = end3[i] - begin3[i];
    g.dim_m1[i] = (float)g.dim[i] - 1.0f;
    g.factor[i] = g.dim_m1[i] / range;
    g.factor_inv[i] = 1 / g.factor[i];
  }
  std::vector<float> data((size_t)g.dim[0] * g.dim[1] * g.dim[2], 0.f);
  size_t k = 0;
  for (int z = 0; z < n3[2]; z++)
    for (int y = 0; y < n3[1]; y++)
      for (int x = 0; x < n3[0]; x++, k++)
        data[(size_t)x + (size_t)g.dim[0] * ((size_t)y + (size_t)g.dim[1] * z)] = (float)(-(values[k] * (double)scaling_factor));
  v.d_ug.upload(data.data(), data.size(), v.stream);
  MIG_HIP(hipStreamSynchronize(v.stream));
  v.ug_geom = g;
  v.have_ug = true;
  return MI_OK;
  VCATCH_STATUS
}

// setup_user_gd (main.cpp:635-670) + the value lines of grid::init: three header lines are skipped, then SPACING g,
// NELEMENTS nx ny nz, CENTER cx cy cz (fields split at every white-space character like boost::split with
// is_space -- no token compression), then one number per line.  Arithmetic in the reference's types (fl = float,
// atof = double).
mi_status mi_user_grid_parse(const char *text, size_t len, float *begin3, float *end3, int32_t *n3, double *values,
                             size_t cap, size_t *n_values) {
  VTRY
  MIG_CHECK(text && begin3 && end3 && n3 && n_values, 1, "bad arguments");
  size_t pos = 0;
  auto getline = [&](std::string &line) -> bool {
    if (pos >= len) {
      line.clear();
      return false;
    }
    size_t e = pos;
    while (e < len && text[e] != '\n') e++;
    line.assign(text + pos, e - pos);
    pos = e < len ? e + 1 : e;
    return true;
  };
  auto split = [](const std::string &line) {
    std::vector<std::string> t(1);
    for (char c : line) {
      if (c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f') t.emplace_back();
      else t.back().push_back(c);
    }
    return t;
  };
  auto field = [](const std::vector<std::string> &t, size_t i) { return i < t.size() ? atof(t[i].c_str()) : 0.0; };
  std::string line;
  for (int i = 0; i < 3; i++) getline(line);
  getline(line);
  const float gran = (float)field(split(line), 1);
  getline(line);
  std::vector<std::string> t = split(line);
  float size[3], center[3];
  for (int i = 0; i < 3; i++) size[i] = (float)((field(t, 1 + i) + 1) * gran);
  getline(line);
  t = split(line);
  for (int i = 0; i < 3; i++) center[i] = (float)(field(t, 1 + i) + 0.5 * gran);
  MIG_CHECK(gran > 0.f, 1, "user grid: SPACING must be positive");
  size_t total = 1;
  for (int i = 0; i < 3; i++) {
    n3[i] = (int32_t)std::ceil(size[i] / gran);
    MIG_CHECK(n3[i] > 0 && n3[i] < 4096, 1, "user grid: bad NELEMENTS");
    const float real_span = gran * (float)n3[i];
    begin3[i] = center[i] - real_span / 2;
    end3[i] = begin3[i] + real_span;
    total *= (size_t)n3[i];
  }
  *n_values = total;
  if (!values) return MI_OK;  // size query
  MIG_CHECK(cap >= total, 1, "user grid: value buffer too small");
  for (size_t k = 0; k < total; k++) {  // (a short file leaves atof("") = 0 like the reference's failed getline)
    getline(line);
    values[k] = atof(line.c_str());
  }
  return MI_OK;
  VCATCH_STATUS
}

mi_status mi_vina_cache_grid(mi_vina *vv, int smt, float *out, size_t n_floats) {
  VTRY
  MIG_CHECK(vv && out, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_cache && smt >= 0 && smt < kVinaTypes && v.grid_off[smt] >= 0, 4, "no grid for this type");
  MIG_CHECK(n_floats == v.grid_pts, 1, "grid size mismatch");
  MIG_HIP(hipMemcpy(out, v.d_grids.p + v.grid_off[smt], n_floats * sizeof(float), hipMemcpyDeviceToHost));
  return MI_OK;
  VCATCH_STATUS
}

static void build_ligand(const mi_ligand_desc *d, LigandDev &out, hipStream_t stream) {
  MIG_CHECK(d && d->n_atoms > 0 && d->n_nodes > 0 && d->n_pairs >= 0, 1, "bad ligand description");
  MIG_CHECK(d->smt && d->local_xyz && d->node_parent && d->node_atom_begin && d->node_atom_end &&
                d->node_rel_origin && d->node_rel_axis && (d->n_pairs == 0 || d->pairs),
            1, "NULL array in ligand description");
  const int na = d->n_atoms, nn = d->n_nodes, np = d->n_pairs;
  MIG_CHECK(vina_wave_lds_bytes(na, nn, np, true, true) <= 152 * 1024, 1,
            "ligand too large for the per-wave LDS workspace (160 KB per workgroup)");
  MIG_CHECK(nn <= 57, 1, "more than 56 rotatable bonds: the minimiser keeps the 7 + T conformation entries one per lane");
  const int n_mov = d->n_movable > 0 ? d->n_movable : na;
  MIG_CHECK(n_mov <= na && d->lig_begin >= 0 && d->lig_begin <= d->lig_end && d->lig_end <= na, 1,
            "bad n_movable / ligand atom range");
  std::vector<int> node_of(na, -1);
  for (int k = 0; k < nn; k++) {
    MIG_CHECK(d->node_parent[k] < k && (k == 0 ? d->node_parent[k] == -1 : (d->node_parent[k] >= 0 || d->node_parent[k] == -2)),
              1, "nodes must be in DFS pre-order (parent index < node index, root first; -2 = a residue's first segment)");
    MIG_CHECK(d->node_atom_begin[k] >= 0 && d->node_atom_begin[k] <= d->node_atom_end[k] && d->node_atom_end[k] <= na,
              1, "bad node atom range");
    for (int i = d->node_atom_begin[k]; i < d->node_atom_end[k]; i++) node_of[i] = k;
  }
  for (int i = 0; i < na; i++) {
    MIG_CHECK(node_of[i] >= 0 || i >= n_mov, 1, "movable atom not covered by any node");  // inflex atoms have none
    MIG_CHECK(node_of[i] < 0 || i < n_mov, 1, "an atom beyond n_movable belongs to a node");
    MIG_CHECK(d->smt[i] >= 0 && d->smt[i] < kVinaTypes, 1, "ligand smina type out of range");
  }
  // CSR of children (increasing index) and of pairs per atom (pair order)
  std::vector<int> child_start(nn + 1, 0), child_list;
  for (int k = 0; k < nn; k++) {
    child_start[k] = (int)child_list.size();
    for (int c = k + 1; c < nn; c++)
      if (d->node_parent[c] == k) child_list.push_back(c);
  }
  child_start[nn] = (int)child_list.size();
  std::vector<int> degree(na, 0);
  for (int p = 0; p < np; p++) {
    const int a = d->pairs[2 * p], b = d->pairs[2 * p + 1];
    MIG_CHECK(a >= 0 && a < na && b >= 0 && b < na && a != b, 1, "bad pair");
    degree[a]++;
    degree[b]++;
  }
  // contribution slots: atom i's list, in pair order, padded to a multiple of four (vina.h: slot_start / pair_slots)
  std::vector<int> aps(na + 1, 0), apl(2 * (size_t)np, 0), fill(na, 0);
  for (int i = 0; i < na; i++) aps[i + 1] = aps[i] + ((degree[i] + 3) & ~3);
  // (model::eval_deriv adds the forces of other_pairs before the ligand's own, model.cu:209-216: an atom's list takes
  // its other_pairs entries first, whatever the order of the caller's pair list)
  for (int kind = 1; kind >= 0; kind--)
    for (int p = 0; p < np; p++) {
      if ((d->pair_kind ? (d->pair_kind[p] != 0 ? 1 : 0) : 0) != kind) continue;
      const int a = d->pairs[2 * p], b = d->pairs[2 * p + 1];
      apl[2 * p] = aps[a] + fill[a]++;
      apl[2 * p + 1] = aps[b] + fill[b]++;
    }
  // pack ints: smt, node_of, parent, abeg, aend, child_start, child_list, pairs, aps, apl
  std::vector<int> ints;
  auto pushi = [&](const int *p, size_t n) {
    size_t off = ints.size();
    ints.insert(ints.end(), p, p + n);
    while (ints.size() % 4) ints.push_back(0);
    return off;
  };
  std::vector<int> heavy;  // model::get_heavy_atom_movable_coords: the movable non-hydrogen atoms
  for (int i = 0; i < n_mov; i++)
    if (d->smt[i] > 1) heavy.push_back(i);
  const size_t o_heavy = pushi(heavy.empty() ? aps.data() : heavy.data(), heavy.size());
  std::vector<int> depth(nn, 0);  // level of every node; the tree walk may advance one level per step
  int n_levels = 1;
  for (int k = 1; k < nn; k++) {
    depth[k] = d->node_parent[k] == -2 ? 1 : depth[d->node_parent[k]] + 1;
    n_levels = std::max(n_levels, depth[k] + 1);
  }
  const size_t o_depth = pushi(depth.data(), nn);
  const size_t o_smt = pushi(d->smt, na), o_node = pushi(node_of.data(), na), o_par = pushi(d->node_parent, nn),
               o_abeg = pushi(d->node_atom_begin, nn), o_aend = pushi(d->node_atom_end, nn),
               o_cs = pushi(child_start.data(), nn + 1),
               o_cl = pushi(child_list.empty() ? aps.data() : child_list.data(), child_list.size()),
               o_pairs = pushi(np ? d->pairs : aps.data(), 2 * (size_t)np), o_aps = pushi(aps.data(), na + 1),
               o_apl = pushi(apl.empty() ? aps.data() : apl.data(), apl.size());
  const size_t o_cap = d->pair_kind && np ? pushi(d->pair_kind, np) : 0;
  std::vector<float> flts;
  auto pushf = [&](const float *p, size_t n) {
    size_t off = flts.size();
    flts.insert(flts.end(), p, p + n);
    while (flts.size() % 4) flts.push_back(0.f);
    return off;
  };
  const size_t o_loc = pushf(d->local_xyz, 3 * (size_t)na), o_ro = pushf(d->node_rel_origin, 3 * (size_t)nn),
               o_ra = pushf(d->node_rel_axis, 3 * (size_t)nn);
  out.d_int.upload(ints.data(), ints.size(), stream);
  out.d_flt.upload(flts.data(), flts.size(), stream);
  MIG_HIP(hipStreamSynchronize(stream));
  out.h_smt.assign(d->smt, d->smt + na);
  VinaLigand &L = out.lig;
  L.n_atoms = na;
  L.n_nodes = nn;
  L.n_pairs = np;
  L.smt = out.d_int.p + o_smt;
  L.node_of_atom = out.d_int.p + o_node;
  L.parent = out.d_int.p + o_par;
  L.abeg = out.d_int.p + o_abeg;
  L.aend = out.d_int.p + o_aend;
  L.child_start = out.d_int.p + o_cs;
  L.child_list = out.d_int.p + o_cl;
  L.pairs = reinterpret_cast<const int2 *>(out.d_int.p + o_pairs);
  L.slot_start = out.d_int.p + o_aps;
  L.pair_slots = reinterpret_cast<const int2 *>(out.d_int.p + o_apl);
  L.n_heavy = (int)heavy.size();
  L.heavy_list = out.d_int.p + o_heavy;
  L.depth = out.d_int.p + o_depth;
  L.n_levels = n_levels;
  L.n_movable = n_mov;
  L.pair_cap = d->pair_kind && np ? out.d_int.p + o_cap : nullptr;
  L.lig_begin = d->lig_end > d->lig_begin ? d->lig_begin : 0;
  L.lig_end = d->lig_end > d->lig_begin ? d->lig_end : na;
  L.local_xyz = out.d_flt.p + o_loc;
  L.rel_origin = out.d_flt.p + o_ro;
  L.rel_axis = out.d_flt.p + o_ra;
}


mi_status mi_vina_set_ligand(mi_vina *vv, const mi_ligand_desc *d) {
  VTRY
  MIG_CHECK(vv && d, 1, "bad ligand description");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  build_ligand(d, v.one, v.stream);
  v.lig = v.one.lig;
  v.h_lig_smt = v.one.h_smt;
  v.have_lig = true;
  return MI_OK;
  VCATCH_STATUS
}

mi_status mi_vina_set_screen(mi_vina *vv, int n_lig, const mi_ligand_desc *descs) {
  VTRY
  MIG_CHECK(vv && n_lig >= 0 && (n_lig == 0 || descs), 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  std::vector<std::unique_ptr<LigandDev>> fresh;  // swapped in only when every description was accepted
  std::vector<VinaLigand> h(n_lig);
  for (int l = 0; l < n_lig; l++) {
    fresh.emplace_back(new LigandDev());
    build_ligand(descs + l, *fresh.back(), v.stream);
    h[l] = fresh.back()->lig;
  }
  if (n_lig) v.d_screen.upload(h.data(), h.size(), v.stream);
  MIG_HIP(hipStreamSynchronize(v.stream));
  v.screen.swap(fresh);
  return MI_OK;
  VCATCH_STATUS
}

int mi_vina_screen_size(const mi_vina *vv) { return vv ? (int)reinterpret_cast<const Vina *>(vv)->screen.size() : 0; }

mi_status mi_vina_screen_dims(const mi_vina *vv, int32_t *max_conf, int32_t *max_heavy, int32_t *max_atoms) {
  VTRY
  MIG_CHECK(vv && max_conf && max_heavy && max_atoms, 1, "bad arguments");
  const Vina &v = *reinterpret_cast<const Vina *>(vv);
  int mc = 0, mh = 0, ma = 0;
  for (const auto &l : v.screen) {
    mc = std::max(mc, 7 + l->lig.n_nodes - 1);
    mh = std::max(mh, l->lig.n_heavy);
    ma = std::max(ma, l->lig.n_atoms);
  }
  *max_conf = mc;
  *max_heavy = mh;
  *max_atoms = ma;
  return MI_OK;
  VCATCH_STATUS
}

namespace {
// the maxima over the screen's ligands: LDS sizing and row strides of the *_screen calls
VinaLigand screen_big(const Vina &v) {
  VinaLigand big{};
  for (const auto &l : v.screen) {
    big.n_atoms = std::max(big.n_atoms, l->lig.n_atoms);
    big.n_nodes = std::max(big.n_nodes, l->lig.n_nodes);
    big.n_pairs = std::max(big.n_pairs, l->lig.n_pairs);
    big.n_heavy = std::max(big.n_heavy, l->lig.n_heavy);
  }
  return big;
}

void screen_env(Vina &v, VinaEnv &env, const int32_t *item_ligand, int B, const VinaLigand &big) {
  const int nl = (int)v.screen.size();
  for (int b = 0; b < B; b++) MIG_CHECK(item_ligand[b] >= 0 && item_ligand[b] < nl, 1, "item_ligand out of range");
  v.d_chain_lig.upload(item_ligand, B, v.stream);
  env.ligs = v.d_screen.p;
  env.item_lig = v.d_chain_lig.p;
  env.conf_stride = 7 + big.n_nodes - 1;
  env.change_stride = 6 + big.n_nodes - 1;
  env.coord_stride = 3 * big.n_atoms;
}
}  // namespace

mi_status mi_vina_eval_screen(mi_vina *vv, const int32_t *item_ligand, const float *confs, int B, const float *v3,
                              int with_deriv, float *energy, float *change, float *coords) {
  VTRY
  MIG_CHECK(vv && item_ligand && confs && v3 && energy && B >= 0, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_cache && !v.screen.empty(), 4, "build the cache and set the screen's ligands first");
  if (B == 0) return MI_OK;
  const VinaLigand big = screen_big(v);
  VinaEnv env = make_env(v);
  screen_env(v, env, item_ligand, B, big);
  const size_t cs = env.conf_stride, gs = env.change_stride, xs = env.coord_stride;
  v.d_confs.upload(confs, (size_t)B * cs, v.stream);
  v.d_energy.ensure(B);
  if (change) v.d_change.ensure((size_t)B * gs);
  if (coords) v.d_coords.ensure((size_t)B * xs);
  env.direct = (with_deriv & MI_VINA_DIRECT) ? 1 : 0;
  env.exact = (with_deriv & MI_VINA_EXACT) ? 1 : 0;
  env.ug_model = (with_deriv & MI_VINA_USER_TERM) ? 1 : 0;
  with_deriv &= 7;
  launch_vina_eval(env, big, v.d_confs.p, B, v3[0], v3[1], v3[2], with_deriv, v.d_energy.p,
                   change ? v.d_change.p : nullptr, coords ? v.d_coords.p : nullptr, v.stream);
  MIG_HIP(hipGetLastError());
  MIG_HIP(hipMemcpyAsync(energy, v.d_energy.p, B * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (change && with_deriv == 1)
    MIG_HIP(hipMemcpyAsync(change, v.d_change.p, (size_t)B * gs * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (coords) MIG_HIP(hipMemcpyAsync(coords, v.d_coords.p, (size_t)B * xs * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipStreamSynchronize(v.stream));
  return MI_OK;
  VCATCH_STATUS
}

mi_status mi_vina_refine_screen(mi_vina *vv, const int32_t *item_ligand, float *confs, int B, const float *v3,
                                const int32_t *max_iters, float *energy, int32_t *tries) {
  VTRY
  MIG_CHECK(vv && item_ligand && confs && v3 && max_iters && energy && B >= 0, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_cache && !v.screen.empty(), 4, "build the cache and set the screen's ligands first");
  if (B == 0) return MI_OK;
  const VinaLigand big = screen_big(v);
  VinaEnv env = make_env(v);
  screen_env(v, env, item_ligand, B, big);
  v.d_lig_iters.upload(max_iters, v.screen.size(), v.stream);
  env.lig_iters = v.d_lig_iters.p;
  const size_t cs = env.conf_stride;
  v.d_confs.upload(confs, (size_t)B * cs, v.stream);
  v.d_energy.ensure(B);
  v.d_evals.ensure(B);
  launch_vina_refine(env, big, v.d_confs.p, B, v3[0], v3[1], v3[2], 0, v.d_energy.p, v.d_evals.p, v.stream);
  MIG_HIP(hipGetLastError());
  MIG_HIP(hipMemcpyAsync(confs, v.d_confs.p, (size_t)B * cs * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipMemcpyAsync(energy, v.d_energy.p, B * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (tries) MIG_HIP(hipMemcpyAsync(tries, v.d_evals.p, B * sizeof(int), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipStreamSynchronize(v.stream));
  return MI_OK;
  VCATCH_STATUS
}

mi_status mi_vina_eval_batch(mi_vina *vv, const float *confs, int B, const float *v3, int with_deriv, float *energy,
                             float *change, float *coords) {
  VTRY
  MIG_CHECK(vv && confs && v3 && energy && B >= 0, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_cache && v.have_lig, 4, "build the cache and set the ligand first");
  if (B == 0) return MI_OK;
  const int nt = v.lig.n_nodes - 1, n = 6 + nt, nc = 7 + nt;
  v.d_confs.upload(confs, (size_t)B * nc, v.stream);
  v.d_energy.ensure(B);
  if (change) v.d_change.ensure((size_t)B * n);
  if (coords) v.d_coords.ensure((size_t)B * 3 * v.lig.n_atoms);
  VinaEnv env = make_env(v);
  env.direct = (with_deriv & MI_VINA_DIRECT) ? 1 : 0;
  env.exact = (with_deriv & MI_VINA_EXACT) ? 1 : 0;
  env.ug_model = (with_deriv & MI_VINA_USER_TERM) ? 1 : 0;
  with_deriv &= 7;
  launch_vina_eval(env, v.lig, v.d_confs.p, B, v3[0], v3[1], v3[2], with_deriv, v.d_energy.p,
                   change ? v.d_change.p : nullptr, coords ? v.d_coords.p : nullptr, v.stream);
  MIG_HIP(hipGetLastError());
  MIG_HIP(hipMemcpyAsync(energy, v.d_energy.p, B * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (change && with_deriv == 1)
    MIG_HIP(hipMemcpyAsync(change, v.d_change.p, (size_t)B * n * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (coords)
    MIG_HIP(hipMemcpyAsync(coords, v.d_coords.p, (size_t)B * 3 * v.lig.n_atoms * sizeof(float), hipMemcpyDeviceToHost,
                           v.stream));
  MIG_HIP(hipStreamSynchronize(v.stream));
  return MI_OK;
  VCATCH_STATUS
}

mi_status mi_vina_bfgs_batch(mi_vina *vv, float *confs, int B, const float *v3, int max_iters, float *energy,
                             float *grad, int32_t *evals) {
  VTRY
  MIG_CHECK(vv && confs && v3 && energy && B >= 0 && max_iters >= 0, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_cache && v.have_lig, 4, "build the cache and set the ligand first");
  if (B == 0) return MI_OK;
  const int nt = v.lig.n_nodes - 1, n = 6 + nt, nc = 7 + nt;
  v.d_confs.upload(confs, (size_t)B * nc, v.stream);
  v.d_energy.ensure(B);
  v.d_change.ensure((size_t)B * n);
  v.d_evals.ensure(B);
  launch_vina_bfgs(make_env(v), v.lig, v.d_confs.p, B, v3[0], v3[1], v3[2], max_iters, v.d_energy.p, v.d_change.p,
                   v.d_evals.p, v.stream);
  MIG_HIP(hipGetLastError());
  MIG_HIP(hipMemcpyAsync(confs, v.d_confs.p, (size_t)B * nc * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipMemcpyAsync(energy, v.d_energy.p, B * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (grad) MIG_HIP(hipMemcpyAsync(grad, v.d_change.p, (size_t)B * n * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (evals) MIG_HIP(hipMemcpyAsync(evals, v.d_evals.p, B * sizeof(int), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipStreamSynchronize(v.stream));
  return MI_OK;
  VCATCH_STATUS
}

int mi_vina_ligand_heavy_atoms(const mi_vina *vv) {
  return vv && reinterpret_cast<const Vina *>(vv)->have_lig ? reinterpret_cast<const Vina *>(vv)->lig.n_heavy : 0;
}

// boost::mt19937 generator(seed) (parallel_mc.cpp:190-192 hands every task its own seed): the standard MT19937
// seeding, one 624-word state per wave of a chain's team (the waves replay the chain, each on a private copy).
static void upload_mt_states(Vina &v, const uint64_t *seeds, int B) {
  const int W = vina_mc_team(B);
  std::vector<unsigned> st((size_t)B * W * 624);
  for (int b = 0; b < B; b++) {
    unsigned *m = &st[(size_t)b * W * 624];
    m[0] = (unsigned)seeds[b];
    for (int i = 1; i < 624; i++) m[i] = 1812433253u * (m[i - 1] ^ (m[i - 1] >> 30)) + (unsigned)i;
    for (int w = 1; w < W; w++) std::memcpy(m + (size_t)w * 624, m, 624 * sizeof(unsigned));
  }
  v.d_mt.upload(st.data(), st.size(), v.stream);
}

mi_status mi_vina_mc_batch(mi_vina *vv, int B, const uint64_t *seeds, const float *corner1, const float *corner2,
                           const mi_mc_params *P, int32_t *out_n, float *out_e, float *out_conf, float *out_coords,
                           int32_t *evals) {
  VTRY
  MIG_CHECK(vv && seeds && corner1 && corner2 && P && out_n && out_e && B >= 0, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_cache && v.have_lig, 4, "build the cache and set the ligand first");
  MIG_CHECK(P->num_saved > 0 && P->num_saved <= 64 && P->n_steps >= 0 && P->max_iters >= 0 && P->temperature > 0, 1,
            "bad Monte-Carlo parameters (num_saved must be in [1, 64])");
  if (B == 0) return MI_OK;
  const int nt = v.lig.n_nodes - 1, nc = 7 + nt, nh = v.lig.n_heavy, S = P->num_saved;
  MIG_CHECK(vina_mc_lds_bytes(v.lig.n_atoms, v.lig.n_nodes, v.lig.n_pairs, nh, S, true, 1) <= 152 * 1024, 1,
            "ligand too large for the per-wave LDS workspace");
  v.d_seeds.upload(reinterpret_cast<const unsigned long long *>(seeds), B, v.stream);
  upload_mt_states(v, seeds, B);
  v.d_mc_e.ensure((size_t)B * S);
  v.d_mc_conf.ensure((size_t)B * S * nc);
  v.d_mc_xyz.ensure((size_t)B * S * 3 * nh + 1);
  v.d_sc_e.ensure((size_t)B * S);
  v.d_sc_conf.ensure((size_t)B * S * nc);
  v.d_sc_xyz.ensure((size_t)B * S * 3 * nh + 1);
  v.d_out_n.ensure(B);
  v.d_evals.ensure(B);
  VinaMcArgs a{};
  a.n_steps = P->n_steps;
  a.max_iters = P->max_iters;
  a.num_saved = S;
  a.temperature = P->temperature;
  a.amplitude = P->mutation_amplitude;
  a.min_rmsd = P->min_rmsd;
  for (int i = 0; i < 3; i++) {
    a.hunt[i] = P->hunt_cap[i];
    a.auth[i] = P->authentic_v[i];
    a.c1[i] = corner1[i];
    a.c2[i] = corner2[i];
  }
  a.seeds = v.d_seeds.p;
  a.mt = v.d_mt.p;
  a.mt_team = vina_mc_team(B);
  a.scratch_e = v.d_sc_e.p;
  a.scratch_conf = v.d_sc_conf.p;
  a.scratch_coords = v.d_sc_xyz.p;
  a.out_e = v.d_mc_e.p;
  a.out_conf = v.d_mc_conf.p;
  a.out_coords = v.d_mc_xyz.p;
  a.out_n = v.d_out_n.p;
  a.evals = v.d_evals.p;
  static const bool prof = option(OPT_MI_VINA_MC_PROFILE) != nullptr;
  DevBuf<long long> prof_buf;  // diagnostic only; released on every exit path
  long long *d_prof = nullptr;
  if (prof) {
    prof_buf.ensure((size_t)B * 12);
    a.prof = d_prof = prof_buf.p;
  }
  launch_vina_mc(make_env(v), v.lig, a, B, v.stream);
  MIG_HIP(hipGetLastError());
  if (prof) {  // diagnostic: mean per-chain time of each phase, 100 MHz ticks -> ms
    std::vector<long long> hp((size_t)B * 12);
    MIG_HIP(hipMemcpyAsync(hp.data(), d_prof, hp.size() * sizeof(long long), hipMemcpyDeviceToHost, v.stream));
    MIG_HIP(hipStreamSynchronize(v.stream));
    double m[12] = {0};
    for (int b = 0; b < B; b++)
      for (int i = 0; i < 12; i++) m[i] += (double)hp[(size_t)b * 12 + i] / B;
    fprintf(stderr,
            "[mi_vina_mc] B=%d steps=%d  per chain (ms): mutate %.2f  bfgs_hunt %.2f  energy+metropolis %.2f  "
            "bfgs_auth %.2f  energy+copy %.2f  insert %.2f | inside bfgs: evals %.2f  direction %.2f  increment %.2f  "
            "exchange %.2f  update %.2f | accepted %.0f\n",
            B, a.n_steps, m[0] * 1e-5, m[1] * 1e-5, m[2] * 1e-5, m[3] * 1e-5, m[4] * 1e-5, m[5] * 1e-5, m[6] * 1e-5,
            m[8] * 1e-5, m[9] * 1e-5, m[10] * 1e-5, m[11] * 1e-5, m[7]);
  }
  MIG_HIP(hipMemcpyAsync(out_n, v.d_out_n.p, B * sizeof(int), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipMemcpyAsync(out_e, v.d_mc_e.p, (size_t)B * S * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (out_conf)
    MIG_HIP(hipMemcpyAsync(out_conf, v.d_mc_conf.p, (size_t)B * S * nc * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (out_coords && nh > 0)
    MIG_HIP(hipMemcpyAsync(out_coords, v.d_mc_xyz.p, (size_t)B * S * 3 * nh * sizeof(float), hipMemcpyDeviceToHost,
                           v.stream));
  if (evals) MIG_HIP(hipMemcpyAsync(evals, v.d_evals.p, B * sizeof(int), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipStreamSynchronize(v.stream));
  return MI_OK;
  VCATCH_STATUS
}

mi_status mi_vina_mc_screen(mi_vina *vv, int B, const int32_t *chain_ligand, const uint64_t *seeds, const float *corner1,
                            const float *corner2, const mi_mc_params *P, int32_t *out_n, float *out_e, float *out_conf,
                            float *out_coords, int32_t *evals) {
  VTRY
  MIG_CHECK(vv && chain_ligand && seeds && corner1 && corner2 && P && out_n && out_e && B >= 0, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  const int nl = (int)v.screen.size();
  MIG_CHECK(v.have_cache && nl > 0, 4, "build the cache and set the screen's ligands first");
  const int S = P[0].num_saved;
  MIG_CHECK(S > 0 && S <= 64 && P[0].temperature > 0, 1, "bad Monte-Carlo parameters (num_saved must be in [1, 64])");
  if (B == 0) return MI_OK;
  const VinaLigand big = screen_big(v);  // the maxima over the set size the LDS workspace and the container strides
  std::vector<int> steps(nl), iters(nl);
  for (int l = 0; l < nl; l++) {
    MIG_CHECK(P[l].num_saved == S && P[l].n_steps >= 0 && P[l].max_iters >= 0, 1,
              "every ligand of a screen uses the same num_saved");
    steps[l] = P[l].n_steps;
    iters[l] = P[l].max_iters;
  }
  for (int b = 0; b < B; b++) MIG_CHECK(chain_ligand[b] >= 0 && chain_ligand[b] < nl, 1, "chain_ligand out of range");
  const int ncm = 7 + big.n_nodes - 1, nhm = big.n_heavy;
  MIG_CHECK(vina_mc_lds_bytes(big.n_atoms, big.n_nodes, big.n_pairs, nhm, S, true, 1) <= 152 * 1024, 1,
            "ligands too large for the per-wave LDS workspace");
  v.d_seeds.upload(reinterpret_cast<const unsigned long long *>(seeds), B, v.stream);
  upload_mt_states(v, seeds, B);
  v.d_chain_lig.upload(chain_ligand, B, v.stream);
  v.d_lig_steps.upload(steps.data(), nl, v.stream);
  v.d_lig_iters.upload(iters.data(), nl, v.stream);
  v.d_mc_e.ensure((size_t)B * S);
  v.d_mc_conf.ensure((size_t)B * S * ncm);
  v.d_mc_xyz.ensure((size_t)B * S * 3 * nhm + 1);
  v.d_sc_e.ensure((size_t)B * S);
  v.d_sc_conf.ensure((size_t)B * S * ncm);
  v.d_sc_xyz.ensure((size_t)B * S * 3 * nhm + 1);
  v.d_out_n.ensure(B);
  v.d_evals.ensure(B);
  VinaMcArgs a{};
  a.num_saved = S;
  a.temperature = P[0].temperature;
  a.amplitude = P[0].mutation_amplitude;
  a.min_rmsd = P[0].min_rmsd;
  for (int i = 0; i < 3; i++) {
    a.hunt[i] = P[0].hunt_cap[i];
    a.auth[i] = P[0].authentic_v[i];
    a.c1[i] = corner1[i];
    a.c2[i] = corner2[i];
  }
  a.seeds = v.d_seeds.p;
  a.mt = v.d_mt.p;
  a.mt_team = vina_mc_team(B);
  a.scratch_e = v.d_sc_e.p;
  a.scratch_conf = v.d_sc_conf.p;
  a.scratch_coords = v.d_sc_xyz.p;
  a.out_e = v.d_mc_e.p;
  a.out_conf = v.d_mc_conf.p;
  a.out_coords = v.d_mc_xyz.p;
  a.out_n = v.d_out_n.p;
  a.evals = v.d_evals.p;
  {  // longest searches first: steps x (pairs + atoms) as the cost of a chain; stable, so ties keep the caller's order
    std::vector<int> order(B);
    std::vector<double> cost(B);
    for (int b = 0; b < B; b++) {
      const VinaLigand &L = v.screen[chain_ligand[b]]->lig;
      order[b] = b;
      cost[b] = (double)steps[chain_ligand[b]] * (L.n_pairs + 8.0 * L.n_atoms);
    }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost[x] > cost[y]; });
    v.d_chain_order.upload(order.data(), B, v.stream);
    a.order = v.d_chain_order.p;
  }
  a.ligs = v.d_screen.p;
  a.chain_lig = v.d_chain_lig.p;
  a.lig_steps = v.d_lig_steps.p;
  a.lig_iters = v.d_lig_iters.p;
  a.conf_stride = ncm;
  a.coord_stride = 3 * nhm;
  launch_vina_mc(make_env(v), big, a, B, v.stream);
  MIG_HIP(hipGetLastError());
  MIG_HIP(hipMemcpyAsync(out_n, v.d_out_n.p, B * sizeof(int), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipMemcpyAsync(out_e, v.d_mc_e.p, (size_t)B * S * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (out_conf)
    MIG_HIP(hipMemcpyAsync(out_conf, v.d_mc_conf.p, (size_t)B * S * ncm * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (out_coords && nhm > 0)
    MIG_HIP(hipMemcpyAsync(out_coords, v.d_mc_xyz.p, (size_t)B * S * 3 * nhm * sizeof(float), hipMemcpyDeviceToHost,
                           v.stream));
  if (evals) MIG_HIP(hipMemcpyAsync(evals, v.d_evals.p, B * sizeof(int), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipStreamSynchronize(v.stream));
  return MI_OK;
  VCATCH_STATUS
}

mi_status mi_vina_refine_batch(mi_vina *vv, float *confs, int B, const float *v3, int max_iters, float *energy,
                               int32_t *tries) {
  VTRY
  MIG_CHECK(vv && confs && v3 && energy && B >= 0 && max_iters >= 0, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_cache && v.have_lig, 4, "build the cache (it defines the search box) and set the ligand first");
  if (B == 0) return MI_OK;
  const int nt = v.lig.n_nodes - 1, nc = 7 + nt;
  v.d_confs.upload(confs, (size_t)B * nc, v.stream);
  v.d_energy.ensure(B);
  v.d_evals.ensure(B);
  launch_vina_refine(make_env(v), v.lig, v.d_confs.p, B, v3[0], v3[1], v3[2], max_iters, v.d_energy.p, v.d_evals.p,
                     v.stream);
  MIG_HIP(hipGetLastError());
  MIG_HIP(hipMemcpyAsync(confs, v.d_confs.p, (size_t)B * nc * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipMemcpyAsync(energy, v.d_energy.p, B * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (tries) MIG_HIP(hipMemcpyAsync(tries, v.d_evals.p, B * sizeof(int), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipStreamSynchronize(v.stream));
  return MI_OK;
  VCATCH_STATUS
}

// ---------------------------------------------------------------------------------------------
// CNN in the optimisation loop: non_cache_cnn as quasi_newton's igrid (non_cache_cnn.cpp:33-54,79-169;
// selected at main.cpp:475-476 for --cnn_scoring refinement and above).
// ---------------------------------------------------------------------------------------------
// igrid::eval / eval_deriv of the cache igrid on coordinates (cache.cpp:50-83), see mi_gnina.h
mi_status mi_vina_cache_eval_coords(mi_vina *vv, const float *coords, const int32_t *smt, int n_atoms, int B, float v1,
                                    float *energy, float *minus_forces) {
  VTRY
  MIG_CHECK(vv && coords && smt && energy && n_atoms >= 0 && n_atoms <= 16384 && B >= 0, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_cache, 4, "build the cache first (mi_vina_build_cache)");
  if (B == 0 || n_atoms == 0) {
    for (int b = 0; b < B; b++) energy[b] = 0.f;
    return MI_OK;
  }
  v.d_coords.upload(coords, (size_t)B * n_atoms * 3, v.stream);
  v.d_smt_in.upload(smt, n_atoms, v.stream);
  v.d_energy.ensure(B);
  if (minus_forces) v.d_ext_forces.ensure((size_t)B * n_atoms * 3);
  launch_vina_cache_coords(make_env(v), v.d_coords.p, v.d_smt_in.p, n_atoms, B, v1, v.d_energy.p,
                           minus_forces ? v.d_ext_forces.p : nullptr, v.stream);
  MIG_HIP(hipGetLastError());
  MIG_HIP(hipMemcpyAsync(energy, v.d_energy.p, B * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (minus_forces)
    MIG_HIP(hipMemcpyAsync(minus_forces, v.d_ext_forces.p, (size_t)B * n_atoms * 3 * sizeof(float), hipMemcpyDeviceToHost,
                           v.stream));
  MIG_HIP(hipStreamSynchronize(v.stream));
  return MI_OK;
  VCATCH_STATUS
}

mi_status mi_vina_coords_batch(mi_vina *vv, const float *confs, int B, float *coords) {
  VTRY
  MIG_CHECK(vv && confs && coords && B >= 0, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_lig, 4, "set the ligand first");
  if (B == 0) return MI_OK;
  const int nc = 7 + v.lig.n_nodes - 1;
  v.d_confs.upload(confs, (size_t)B * nc, v.stream);
  v.d_coords.ensure((size_t)B * 3 * v.lig.n_atoms);
  launch_vina_coords(make_env(v), v.lig, v.d_confs.p, B, v.d_coords.p, v.stream);
  MIG_HIP(hipGetLastError());
  MIG_HIP(hipMemcpyAsync(coords, v.d_coords.p, (size_t)B * 3 * v.lig.n_atoms * sizeof(float), hipMemcpyDeviceToHost,
                         v.stream));
  MIG_HIP(hipStreamSynchronize(v.stream));
  return MI_OK;
  VCATCH_STATUS
}

namespace {

// One batched evaluation of non_cache_cnn::eval_deriv (with_deriv) / eval for the conformations in `confs`:
// coordinates on the device -> CNN forward(+backward) for all of them in one launch sequence -> penalties and
// the fold into change.  Scratch vectors live in the caller so the refinement loop does not reallocate.
struct CnnEvalScratch {
  std::vector<float> coords, lig_grad, pose, aff, loss, var;
  std::vector<float> lig_xyz, flex_xyz, lg, fg;  // combined model: ligand / side-chain rows and their gradients
};

mi_status cnn_eval(Vina &v, mi_scorer *sc, const float *confs, int B, const mi_cnn_box *box, const float *cnn_centers,
                   float slope, int with_deriv, float *energy, float *change, CnnEvalScratch &s) {
  const int na = v.lig.n_atoms, nt = v.lig.n_nodes - 1, n = 6 + nt;
  s.coords.resize((size_t)B * na * 3);
  s.pose.resize(B), s.aff.resize(B), s.loss.resize(B), s.var.resize(B);
  mi_status st = mi_vina_coords_batch(reinterpret_cast<mi_vina *>(&v), confs, B, s.coords.data());
  if (st != MI_OK) return st;
  // grid centre = NaN -> recomputed from the ligand on every forward (cnn_torch_scorer.cpp:137-142)
  const int lb = v.lig.lig_begin, le = v.lig.lig_end;
  const bool combined = le - lb != na;  // gnina's combined model: atoms = [flexible side chains | ligand | inflex]
  if (!combined) {
    if (with_deriv) {
      s.lig_grad.resize((size_t)B * na * 3);
      st = mi_scorer_score_grad(sc, s.coords.data(), v.h_lig_smt.data(), B, na, nullptr, s.pose.data(), s.aff.data(),
                                s.loss.data(), s.var.data(), s.lig_grad.data());
    } else {
      st = mi_scorer_score_batch(sc, s.coords.data(), v.h_lig_smt.data(), B, na, nullptr, s.pose.data(), s.aff.data(),
                                 s.loss.data(), s.var.data());
    }
  } else {
    // DLScorer::setLigand takes the ligand's atoms, setReceptor refreshes the movable side-chain atoms -- the first rows
    // of the scorer's receptor, declared with mi_scorer_set_flex (dl_scorer.cpp:36-193) -- and getGradient sends both
    // gradients back by movable-atom index (cnn_torch_scorer.cpp:208-228): s.lig_grad ends up indexed like the model's
    // atoms, [side chains | ligand | inflex (no force)].
    const int nl = le - lb, nf = lb;
    MIG_CHECK(mi_scorer_flex_count(sc) == nf, 1,
              "the combined model has " + std::to_string(nf) + " movable side-chain atoms but the scorer declares " +
                  std::to_string(mi_scorer_flex_count(sc)) + " flexible receptor rows (mi_scorer_set_flex: the first rows of "
                  "mi_pdbqt_read_receptor_flex's output)");
    s.lig_xyz.resize((size_t)B * nl * 3), s.flex_xyz.resize((size_t)B * nf * 3);
    for (int b = 0; b < B; b++) {
      const float *c = &s.coords[(size_t)b * na * 3];
      std::copy(c, c + 3 * nf, s.flex_xyz.begin() + (size_t)b * nf * 3);
      std::copy(c + 3 * lb, c + 3 * le, s.lig_xyz.begin() + (size_t)b * nl * 3);
    }
    if (with_deriv) s.lg.resize((size_t)B * nl * 3), s.fg.resize((size_t)B * nf * 3);
    st = mi_scorer_score_flex(sc, s.lig_xyz.data(), v.h_lig_smt.data() + lb, B, nl, nullptr, nf ? s.flex_xyz.data() : nullptr,
                              s.pose.data(), s.aff.data(), s.loss.data(), s.var.data(), with_deriv ? s.lg.data() : nullptr,
                              with_deriv && nf ? s.fg.data() : nullptr);
    if (st == MI_OK && with_deriv) {
      s.lig_grad.assign((size_t)B * na * 3, 0.f);
      for (int b = 0; b < B; b++) {
        float *g = &s.lig_grad[(size_t)b * na * 3];
        std::copy(s.fg.begin() + (size_t)b * nf * 3, s.fg.begin() + (size_t)(b + 1) * nf * 3, g);
        std::copy(s.lg.begin() + (size_t)b * nl * 3, s.lg.begin() + (size_t)(b + 1) * nl * 3, g + 3 * lb);
      }
    }
  }
  if (st != MI_OK) return st;
  // penalties + fold (confs are still in v.d_confs from the coordinate launch)
  DevBuf<float> &d_f = v.d_ext_forces, &d_e = v.d_ext_e, &d_c = v.d_ext_centers;
  VinaExtArgs a{};
  if (with_deriv) {
    d_f.upload(s.lig_grad.data(), s.lig_grad.size(), v.stream);
    a.forces = d_f.p;
  }
  d_e.upload(s.loss.data(), B, v.stream);
  a.e_in = d_e.p;
  a.use_box = box && box->use_search_box ? 1 : 0;
  for (int k = 0; k < 3; k++) {
    a.box_begin[k] = box ? box->box_begin[k] : 0.f;
    a.box_end[k] = box ? box->box_end[k] : 0.f;
  }
  if (cnn_centers) {
    d_c.upload(cnn_centers, (size_t)B * 3, v.stream);
    a.cnn_center = d_c.p;
    a.cnn_half = box->cnn_dimension / 2.0f;
  }
  a.slope = slope;
  a.per_atom_forces = box && box->per_atom_forces ? 1 : 0;
  a.v = box && box->v > 0 ? box->v : 1000.0f;  // the curl cap `v` of eval_deriv(m, v, user_grid): user-grid term, blend
  if (box && with_deriv && (box->mix_emp_force || box->mix_emp_energy)) {
    MIG_CHECK(v.n_rec > 0, 4, "mix_emp_force / mix_emp_energy need the receptor (mi_vina_set_receptor)");
    a.mix_force = box->mix_emp_force ? 1 : 0;
    a.mix_energy = box->mix_emp_energy ? 1 : 0;
    a.weight = box->empirical_weight;
    a.v = box->v;
  }
  v.d_energy.ensure(B);
  if (with_deriv) v.d_change.ensure((size_t)B * n);
  launch_vina_extforce(make_env(v), v.lig, v.d_confs.p, B, a, v.d_energy.p, with_deriv ? v.d_change.p : nullptr,
                       v.stream);
  MIG_HIP(hipGetLastError());
  MIG_HIP(hipMemcpyAsync(energy, v.d_energy.p, B * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (with_deriv)
    MIG_HIP(hipMemcpyAsync(change, v.d_change.p, (size_t)B * n * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipStreamSynchronize(v.stream));
  return MI_OK;
}

// --- host restatement of the optimiser state (bfgs.h:34-91,357-502) as a resumable state machine: every
// chain asks for one (energy, change) evaluation at a time, so that the evaluations of all chains of a
// round go to the device as one batch. ---
inline int hidx(int i, int j) { return i <= j ? i + j * (j + 1) / 2 : j + i * (i + 1) / 2; }

inline float dot_seq(const float *a, const float *b, int n) {  // scalar_product, bfgs.h:45-50
  float t = 0;
  for (int i = 0; i < n; i++) t += a[i] * b[i];
  return t;
}

constexpr float kPi = 3.14159265358979323846f, kEps = 1.1920928955078125e-07f;

void normalize_angle(float &x) {  // quaternion.h:259-282
  if (x > 3 * kPi) {
    float n = (x - kPi) / (2 * kPi);
    x -= 2 * kPi * std::ceil(n);
    normalize_angle(x);
  } else if (x < -3 * kPi) {
    float n = (-x - kPi) / (2 * kPi);
    x += 2 * kPi * std::ceil(n);
    normalize_angle(x);
  } else if (x > kPi) {
    x -= 2 * kPi;
  } else if (x < -kPi) {
    x += 2 * kPi;
  }
}

void conf_increment(float *x, const float *p, float alpha, int nt) {  // conf.h:54-59,103-118; quaternion.cu:32-62,96-100
  x[0] += alpha * p[0], x[1] += alpha * p[1], x[2] += alpha * p[2];
  const float rx = alpha * p[3], ry = alpha * p[4], rz = alpha * p[5];
  const float angle = std::sqrt(rx * rx + ry * ry + rz * rz);
  float rq[4] = {1, 0, 0, 0};
  if (angle > kEps) {
    const float inv = 1 / angle;
    float a = angle;
    normalize_angle(a);
    const float c = std::cos(a / 2), s = std::sin(a / 2);
    rq[0] = c, rq[1] = s * (inv * rx), rq[2] = s * (inv * ry), rq[3] = s * (inv * rz);
  }
  const float *q = x + 3;
  float nq[4] = {rq[0] * q[0] - rq[1] * q[1] - rq[2] * q[2] - rq[3] * q[3],
                 rq[0] * q[1] + rq[1] * q[0] + rq[2] * q[3] - rq[3] * q[2],
                 rq[0] * q[2] - rq[1] * q[3] + rq[2] * q[0] + rq[3] * q[1],
                 rq[0] * q[3] + rq[1] * q[2] - rq[2] * q[1] + rq[3] * q[0]};
  const float s2 = nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3];
  if (!(std::fabs(s2 - 1) < 1e-6f)) {  // quaternion_normalize_approx, quaternion.h:327-343
    const float inv = 1 / std::sqrt(s2);
    for (float &c : nq) c *= inv;
  }
  x[3] = nq[0], x[4] = nq[1], x[5] = nq[2], x[6] = nq[3];
  for (int i = 0; i < nt; i++) {
    float d = alpha * p[6 + i];
    normalize_angle(d);
    x[7 + i] += d;
    normalize_angle(x[7 + i]);
  }
}

struct BfgsChain {
  int n = 0, nt = 0, nc = 0, max_iters = 0;
  std::vector<float> x, x_new, x_orig, g, g_new, g_orig, p, y, mhy, h;
  float f0 = 0, f_orig = 0, alpha = 1, pg = 0;
  bool accurate = false;  // accurate_line_search (bfgs.h:104-180) instead of fast_line_search
  bool simple = false;    // simple_gradient_ascent (bfgs.h:234-355): p = -g every iteration, no Hessian estimate
  float ls_alpha2 = 0, ls_f2 = 0, alamin = 0;
  int step = 0, trial = 0;
  long evals = 0;
  enum { Init, Line, Done } state = Init;

  void start(const float *conf, int nt_, int max_iters_, bool accurate_ = false, bool simple_ = false) {
    nt = nt_, n = 6 + nt_, nc = 7 + nt_, max_iters = max_iters_;
    accurate = accurate_ || simple_;
    simple = simple_;
    x.assign(conf, conf + nc);
    x_new = x;
    g.assign(n, 0), g_new.assign(n, 0), p.assign(n, 0), y.assign(n, 0), mhy.assign(n, 0);
    h.assign((size_t)n * (n + 1) / 2, 0.f);
    for (int i = 0; i < n; i++) h[hidx(i, i)] = 1;
    step = 0;
    state = Init;
  }
  const float *request() const { return state == Init ? x.data() : x_new.data(); }  // conformation to evaluate next

  void begin_step() {
    if (step >= max_iters) {
      finish();
      return;
    }
    if (simple) {
      for (int i = 0; i < n; i++) p[i] = -g[i];  // set_to_neg, bfgs.h:262
    } else
    for (int i = 0; i < n; i++) {  // minus_mat_vec_product, bfgs.h:34-43
      float sum = 0;
      for (int j = 0; j < n; j++) sum += h[hidx(i, j)] * g[j];
      p[i] = -sum;
    }
    alpha = 1;
    trial = 0;
    pg = dot_seq(p.data(), g.data(), n);
    if (accurate) {
      if (pg >= 0) {  // not a descent direction: alpha = 0, bfgs gives up (bfgs.h:116-122,417-426)
        finish();
        return;
      }
      // compute_lambdamin (bfgs.h:93-102) with x in change indexing (conf.h:459-490): position,
      // quaternion_to_angle(orientation) (quaternion.cu:46-62), torsions
      float ang[3] = {0, 0, 0};
      const float c = x[3];
      if (c > -1 && c < 1) {
        float angle = 2 * std::acos(c);
        if (angle > kPi) angle -= 2 * kPi;
        const float sn = std::sin(angle / 2);
        if (!(std::fabs(sn) < kEps))
          for (int k = 0; k < 3; k++) ang[k] = x[4 + k] * (angle / sn);
      }
      float test = 0;
      for (int i = 0; i < n; i++) {
        const float xi = i < 3 ? x[i] : i < 6 ? ang[i - 3] : x[i + 1];
        const float t = std::fabs(p[i]) / std::max(std::fabs(xi), 1.0f);
        if (t > test) test = t;
      }
      alamin = kEps / test;
      ls_alpha2 = 0, ls_f2 = 0;
    }
    x_new = x;
    conf_increment(x_new.data(), p.data(), alpha, nt);
    state = Line;
  }
  void finish() {
    if (!(f0 <= f_orig)) {  // bfgs.h:493-497
      f0 = f_orig;
      x = x_orig;
      g = g_orig;
    }
    state = Done;
  }
  void feed(float f, const float *grad) {  // the evaluation of request() has arrived
    evals++;
    if (state == Init) {
      f0 = f_orig = f;
      g.assign(grad, grad + n);
      g_orig = g;
      x_orig = x;
      g_new = g;
      begin_step();
      return;
    }
    g_new.assign(grad, grad + n);
    const float f1 = f;
    if (accurate) {  // accurate_line_search, bfgs.h:128-179 (fl arithmetic; the literals 2.0 / 3.0 are double as written)
      const float slope = pg;
      if (alpha < alamin || !std::isfinite(alpha)) {  // too small a step: alpha = 0, give up
        finish();
        return;
      }
      if (!(f1 <= f0 + 1.0e-4f * alpha * slope)) {  // backtrack
        float tmplam;
        if (alpha == 1.0f) {
          tmplam = (float)(-(double)slope / (2.0 * (double)(f1 - f0 - slope)));
        } else {
          const float rhs1 = f1 - f0 - alpha * slope, rhs2 = ls_f2 - f0 - ls_alpha2 * slope;
          const float ca = (rhs1 / (alpha * alpha) - rhs2 / (ls_alpha2 * ls_alpha2)) / (alpha - ls_alpha2);
          const float cb = (-ls_alpha2 * rhs1 / (alpha * alpha) + alpha * rhs2 / (ls_alpha2 * ls_alpha2)) / (alpha - ls_alpha2);
          if (ca == 0.0f) {
            tmplam = (float)(-(double)slope / (2.0 * (double)cb));
          } else {
            const float disc = (float)((double)(cb * cb) - 3.0 * (double)ca * (double)slope);
            if (disc < 0) tmplam = 0.5f * alpha;
            else if (cb <= 0) tmplam = (float)((double)(-cb + std::sqrt(disc)) / (3.0 * (double)ca));
            else tmplam = -slope / (cb + std::sqrt(disc));
          }
          if (tmplam > 0.5f * alpha) tmplam = 0.5f * alpha;
        }
        ls_alpha2 = alpha;
        ls_f2 = f1;
        alpha = std::max(tmplam, 0.1f * alpha);
        x_new = x;
        conf_increment(x_new.data(), p.data(), alpha, nt);
        return;  // next trial
      }
    } else {  // fast_line_search, bfgs.h:73-91
      bool leave = f1 - f0 < 0.0001f * alpha * pg;
      if (!leave) {
        alpha *= 0.5f;
        if (++trial < 10) {
          x_new = x;
          conf_increment(x_new.data(), p.data(), alpha, nt);
          return;  // next trial
        }
      }
      if (alpha == 0) {
        finish();
        return;
      }
    }
    for (int i = 0; i < n; i++) y[i] = g_new[i] - g[i];
    f0 = f1;
    x = x_new;
    g = g_new;
    if (!(dot_seq(g.data(), g.data(), n) >= 1e-4f)) {  // bfgs.h:474
      finish();
      return;
    }
    if (simple) {  // simple_gradient_ascent has no Hessian estimate to maintain
      step++;
      begin_step();
      return;
    }
    if (step == 0) {  // initial Hessian scaling, bfgs.h:478-483
      const float yy = dot_seq(y.data(), y.data(), n);
      if (std::fabs(yy) > kEps) {
        const float dgl = alpha * dot_seq(y.data(), p.data(), n) / yy;
        for (int i = 0; i < n; i++) h[hidx(i, i)] = dgl;
      }
    }
    const float yp = dot_seq(y.data(), p.data(), n);  // bfgs_update, bfgs.h:52-66
    if (!(alpha * yp < kEps)) {
      for (int i = 0; i < n; i++) {
        float sum = 0;
        for (int j = 0; j < n; j++) sum += h[hidx(i, j)] * y[j];
        mhy[i] = -sum;
      }
      const float yhy = -dot_seq(y.data(), mhy.data(), n);
      const float r = 1 / (alpha * yp);
      for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++)
          h[hidx(i, j)] += alpha * r * (mhy[i] * p[j] + mhy[j] * p[i]) + alpha * alpha * (r * r * yhy + r) * p[i] * p[j];
    }
    step++;
    begin_step();
  }
};

}  // namespace

// Monte-Carlo with the CNN as the Metropolis energy (--cnn_scoring metrorescore / metrorefine): see mi_gnina.h and
// vina_mc_cnn_kernel.  update_energy = adjust_center (CNN cube centred on the heavy movable atoms of what `model`
// holds) + non_cache_cnn::eval (CNN loss + out-of-box penalties), one batch for all chains per stop.
mi_status mi_vina_mc_cnn_batch(mi_vina *vv, mi_scorer *sc, int B, const uint64_t *seeds, const float *corner1,
                               const float *corner2, const mi_mc_params *P, const mi_cnn_box *box, int32_t *out_n,
                               float *out_e, float *out_conf, float *out_coords, int32_t *evals, int32_t *cnn_evals) {
  VTRY
  MIG_CHECK(vv && sc && seeds && corner1 && corner2 && P && box && out_n && out_e && B >= 0, 1, "bad arguments");
  MIG_CHECK(box->cnn_dimension > 0, 1, "box->cnn_dimension must be the CNN grid dimension");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_cache && v.have_lig, 4, "build the cache and set the ligand first");
  MIG_CHECK(P->num_saved > 0 && P->num_saved <= 64 && P->n_steps >= 1 && P->max_iters >= 0 && P->temperature > 0, 1,
            "bad Monte-Carlo parameters (num_saved must be in [1, 64], n_steps >= 1)");
  if (B == 0) return MI_OK;
  const int nt = v.lig.n_nodes - 1, nc = 7 + nt, nh = v.lig.n_heavy, S = P->num_saved, na = v.lig.n_atoms;
  MIG_CHECK(vina_mc_lds_bytes(v.lig.n_atoms, v.lig.n_nodes, v.lig.n_pairs, nh, S, true, 1) <= 152 * 1024, 1,
            "ligand too large for the per-wave LDS workspace");
  {  // one MT19937 state per chain (single-wave chains)
    std::vector<unsigned> st((size_t)B * 624);
    for (int b = 0; b < B; b++) {
      unsigned *m = &st[(size_t)b * 624];
      m[0] = (unsigned)seeds[b];
      for (int i = 1; i < 624; i++) m[i] = 1812433253u * (m[i - 1] ^ (m[i - 1] >> 30)) + (unsigned)i;
    }
    v.d_mt.upload(st.data(), st.size(), v.stream);
  }
  v.d_mc_e.ensure((size_t)B * S);
  v.d_mc_conf.ensure((size_t)B * S * nc);
  v.d_mc_xyz.ensure((size_t)B * S * 3 * nh + 1);
  v.d_sc_e.ensure((size_t)B * S);
  v.d_sc_conf.ensure((size_t)B * S * nc);
  v.d_sc_xyz.ensure((size_t)B * S * 3 * nh + 1);
  v.d_out_n.ensure(B);
  v.d_evals.ensure(B);
  VinaMcArgs a{};
  a.n_steps = P->n_steps;
  a.max_iters = P->max_iters;
  a.num_saved = S;
  a.temperature = P->temperature;
  a.amplitude = P->mutation_amplitude;
  a.min_rmsd = P->min_rmsd;
  for (int i = 0; i < 3; i++) {
    a.hunt[i] = P->hunt_cap[i];
    a.auth[i] = P->authentic_v[i];
    a.c1[i] = corner1[i];
    a.c2[i] = corner2[i];
  }
  a.mt = v.d_mt.p;
  a.mt_team = 1;
  a.scratch_e = v.d_sc_e.p;
  a.scratch_conf = v.d_sc_conf.p;
  a.scratch_coords = v.d_sc_xyz.p;
  a.out_e = v.d_mc_e.p;
  a.out_conf = v.d_mc_conf.p;
  a.out_coords = v.d_mc_xyz.p;
  a.out_n = v.d_out_n.p;
  a.evals = v.d_evals.p;
  VinaMcCnnState st{};
  st.f_stride = 3 * nc + 4;
  st.i_stride = 4 + S;
  DevBuf<float> d_stf, d_model, d_ext;
  DevBuf<int> d_sti;
  d_stf.ensure((size_t)B * st.f_stride);
  d_sti.ensure((size_t)B * st.i_stride);
  d_model.ensure((size_t)B * nc);
  d_ext.ensure(B);
  st.st_f = d_stf.p;
  st.st_i = d_sti.p;
  st.model_out = d_model.p;
  st.ext_e = d_ext.p;
  std::vector<float> model((size_t)B * nc), coords((size_t)B * na * 3), centers((size_t)B * 3), energy(B);
  CnnEvalScratch scratch;
  long n_cnn = 0;
  auto update_energy = [&]() -> mi_status {  // monte_carlo.cpp:44-47 for every chain's `model`
    MIG_HIP(hipMemcpyAsync(model.data(), d_model.p, model.size() * sizeof(float), hipMemcpyDeviceToHost, v.stream));
    MIG_HIP(hipStreamSynchronize(v.stream));
    mi_status s1 = mi_vina_coords_batch(vv, model.data(), B, coords.data());
    if (s1 != MI_OK) return s1;
    for (int b = 0; b < B; b++) {  // adjust_center: DLScorer::set_center_from_model (dl_scorer.cpp:197-217)
      float s0 = 0, s1_ = 0, s2 = 0;
      unsigned cnt = 0;
      const float *xyz = &coords[(size_t)b * na * 3];
      for (int i = 0; i < v.lig.n_movable; i++)  // get_heavy_atom_movable_coords: flexible side chains included
        if (v.h_lig_smt[i] > 1) s0 += xyz[3 * i], s1_ += xyz[3 * i + 1], s2 += xyz[3 * i + 2], cnt++;
      centers[3 * b] = s0 / (float)cnt, centers[3 * b + 1] = s1_ / (float)cnt, centers[3 * b + 2] = s2 / (float)cnt;
    }
    s1 = cnn_eval(v, sc, model.data(), B, box, centers.data(), box->slope, 0, energy.data(), nullptr, scratch);
    if (s1 != MI_OK) return s1;
    n_cnn += B;
    MIG_HIP(hipMemcpyAsync(d_ext.p, energy.data(), B * sizeof(float), hipMemcpyHostToDevice, v.stream));
    return MI_OK;
  };
  const VinaEnv env = make_env(v);
  for (int step = 0; step < P->n_steps; step++) {
    st.step = step;
    st.phase = step == 0 ? 0 : 2;      // [insert of the previous step] + propose
    launch_vina_mc_cnn(env, v.lig, a, st, B, v.stream);
    MIG_HIP(hipGetLastError());
    mi_status s1 = update_energy();
    if (s1 != MI_OK) return s1;
    st.phase = 1;                      // Metropolis [+ second minimisation]
    launch_vina_mc_cnn(env, v.lig, a, st, B, v.stream);
    MIG_HIP(hipGetLastError());
    s1 = update_energy();
    if (s1 != MI_OK) return s1;
  }
  st.phase = 3;
  st.step = P->n_steps;
  launch_vina_mc_cnn(env, v.lig, a, st, B, v.stream);
  MIG_HIP(hipGetLastError());
  MIG_HIP(hipMemcpyAsync(out_n, v.d_out_n.p, B * sizeof(int), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipMemcpyAsync(out_e, v.d_mc_e.p, (size_t)B * S * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (out_conf)
    MIG_HIP(hipMemcpyAsync(out_conf, v.d_mc_conf.p, (size_t)B * S * nc * sizeof(float), hipMemcpyDeviceToHost, v.stream));
  if (out_coords && nh > 0)
    MIG_HIP(hipMemcpyAsync(out_coords, v.d_mc_xyz.p, (size_t)B * S * 3 * nh * sizeof(float), hipMemcpyDeviceToHost,
                           v.stream));
  if (evals) MIG_HIP(hipMemcpyAsync(evals, v.d_evals.p, B * sizeof(int), hipMemcpyDeviceToHost, v.stream));
  MIG_HIP(hipStreamSynchronize(v.stream));
  if (cnn_evals) *cnn_evals = (int32_t)std::min<long>(n_cnn, 2147483647L);
  return MI_OK;
  VCATCH_STATUS
}

namespace {

// boost::mt19937 (= std::mt19937) under the restated Boost distributions, host side: the same stream as McRng
// (vina.hip) and the reference's random.cpp:27-75
struct HostRng {
  unsigned mt[624];
  int idx = 624;
  explicit HostRng(unsigned seed) {
    mt[0] = seed;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (unsigned)i;
  }
  unsigned u32() {
    if (idx >= 624) {
      for (int i = 0; i < 624; i++) {
        const unsigned y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    unsigned y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  float fl(float a, float b) {  // uniform_real<float>
    for (;;) {
      const float r = (float)u32() / 4294967296.0f * (b - a) + a;
      if (r < b) return r;
    }
  }
  int irange(int a, int b) {  // uniform_int<int>
    const unsigned range = (unsigned)b - (unsigned)a, brange = 0xffffffffu;
    if (range == 0) return a;
    unsigned bucket = brange / (range + 1);
    if (brange % (range + 1) == range) ++bucket;
    for (;;) {
      const unsigned q = u32() / bucket;
      if (q <= range) return (int)(q + (unsigned)a);
    }
  }
  float normal() {  // normal_distribution<float>(0, 1), a fresh distribution per call (random.cpp:37-42)
    const float r1 = fl(0.f, 1.f), r2 = fl(0.f, 1.f);
    return std::sqrt(-2.0f * std::log(1.0f - r2)) * std::cos(2.0f * 3.14159265358979323846f * r1);
  }
  void inside_sphere(float &x, float &y, float &z) {
    for (;;) {
      x = fl(-1, 1), y = fl(-1, 1), z = fl(-1, 1);
      if (x * x + y * y + z * z < 1) return;
    }
  }
};

}  // namespace

// Monte-Carlo with the CNN as the igrid of BOTH the minimiser and the Metropolis step: --cnn_scoring all
// (parallel_mc.cpp:156-159: (*mc)(m, out, p, new_cnn, ..., new_cnn)).  The chain itself -- RNG stream, mutation, bfgs<> +
// fast_line_search, Metropolis, container -- advances on the host (n <= 40 variables per chain: negligible arithmetic);
// every evaluation the chains of the call ask for in a round, non_cache_cnn::eval_deriv inside the line searches and
// non_cache_cnn::eval at the update_energy points, is ONE device batch (coordinates -> CNN forward [+ backward] ->
// penalties -> fold).  The reference does the same work one chain, one evaluation, one B = 1 forward at a time.
mi_status mi_vina_mc_cnnall_batch(mi_vina *vv, mi_scorer *sc, int B, const uint64_t *seeds, const float *corner1,
                                  const float *corner2, const mi_mc_params *P, const mi_cnn_box *box, int32_t *out_n,
                                  float *out_e, float *out_conf, float *out_coords, int32_t *evals, int32_t *cnn_evals) {
  VTRY
  MIG_CHECK(vv && sc && seeds && corner1 && corner2 && P && box && out_n && out_e && B >= 0, 1, "bad arguments");
  MIG_CHECK(box->cnn_dimension > 0, 1, "box->cnn_dimension must be the CNN grid dimension");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_lig, 4, "set the ligand first");
  MIG_CHECK(P->num_saved > 0 && P->n_steps >= 1 && P->max_iters >= 0 && P->temperature > 0, 1, "bad Monte-Carlo parameters");
  if (B == 0) return MI_OK;
  const int nt = v.lig.n_nodes - 1, n = 6 + nt, nc = 7 + nt, na = v.lig.n_atoms, S = P->num_saved;
  std::vector<int> heavy;
  for (int i = 0; i < v.lig.n_movable; i++)  // the movable non-hydrogen atoms (flexible side chains included)
    if (v.h_lig_smt[i] > 1) heavy.push_back(i);
  const int nh = (int)heavy.size();
  MIG_CHECK(nh > 0, 1, "the ligand has no heavy atoms");
  struct Saved {
    float e;
    std::vector<float> conf, xyz;
  };
  struct Chain {
    HostRng rng;
    std::vector<float> tmp, cand, mconf;
    float tmp_e = 0, best_e = 3.402823466e+38f, cand_e = 0;
    std::vector<Saved> out;
    BfgsChain bf;
    std::vector<float> last_eval;
    bool have_center = false, accepted = false, second = false;
    float center[3] = {0, 0, 0};
    long evals = 0;
    explicit Chain(unsigned seed) : rng(seed) {}
  };
  std::vector<Chain> ch;
  ch.reserve(B);
  for (int b = 0; b < B; b++) {
    ch.emplace_back((unsigned)seeds[b]);
    Chain &c = ch.back();
    c.tmp.assign(nc, 0.f);
    // conf::randomize (conf.h:119-122,189-192)
    for (int k = 0; k < 3; k++) c.tmp[k] = c.rng.fl(corner1[k], corner2[k]);
    float q[4], nrm;
    do {  // random_orientation; abs(qt) scales by the largest component (quaternion.h:169-190)
      for (int k = 0; k < 4; k++) q[k] = c.rng.normal();
      const float mx = std::max(std::max(std::fabs(q[0]), std::fabs(q[1])), std::max(std::fabs(q[2]), std::fabs(q[3])));
      nrm = 0.f;
      if (mx != 0.f) {
        const float inv = (float)(1.0 / (double)mx);
        float sum = (q[0] * inv) * (q[0] * inv);
        sum += (q[1] * inv) * (q[1] * inv);
        sum += (q[2] * inv) * (q[2] * inv);
        sum += (q[3] * inv) * (q[3] * inv);
        nrm = mx * std::sqrt(sum);
      }
    } while (!(nrm > kEps));
    for (int k = 0; k < 4; k++) c.tmp[3 + k] = q[k] / nrm;
    for (int t = 0; t < nt; t++) c.tmp[7 + t] = c.rng.fl(-kPi, kPi);
    c.mconf.assign(nc, 0.f);  // before the first evaluation `model` holds the input pose (zero torsions; only the
    c.mconf[3] = 1.f;         // gyration radius is read from it, which no rigid placement changes)
  }
  CnnEvalScratch scratch;
  long n_cnn = 0;
  std::vector<float> req, cen, e_out, g_out, xyz;
  std::vector<int> idx;
  // one batched evaluation for the chains in idx: confs from `pick`, energies (and changes) back
  auto evaluate = [&](const std::vector<int> &who, auto pick, bool deriv, float vcap) -> mi_status {
    const int nb = (int)who.size();
    req.resize((size_t)nb * nc), cen.resize((size_t)nb * 3), e_out.resize(nb);
    if (deriv) g_out.resize((size_t)nb * n);
    bool centred = true;
    for (int i = 0; i < nb; i++) {
      const float *x = pick(ch[who[i]]);
      std::copy(x, x + nc, req.begin() + (size_t)i * nc);
      std::copy(ch[who[i]].center, ch[who[i]].center + 3, cen.begin() + (size_t)i * 3);
      centred = centred && ch[who[i]].have_center;
    }
    mi_cnn_box bx = *box;
    bx.v = vcap;
    n_cnn += nb;
    // (before a chain's first update_energy its non_cache_cnn has no cube yet: cnn_gd is default-constructed)
    return cnn_eval(v, sc, req.data(), nb, &bx, centred ? cen.data() : nullptr, box->slope, deriv ? 1 : 0, e_out.data(),
                    deriv ? g_out.data() : nullptr, scratch);
  };
  auto coords_of = [&](const std::vector<int> &who, auto pick) -> mi_status {
    const int nb = (int)who.size();
    req.resize((size_t)nb * nc), xyz.resize((size_t)nb * na * 3);
    for (int i = 0; i < nb; i++) {
      const float *x = pick(ch[who[i]]);
      std::copy(x, x + nc, req.begin() + (size_t)i * nc);
    }
    return mi_vina_coords_batch(vv, req.data(), nb, xyz.data());
  };
  // quasi_newton on non_cache_cnn for the chains in `who`, in lock step; leaves bf.x / bf.f0 and last_eval
  auto minimise = [&](const std::vector<int> &who, auto start_from, float vcap) -> mi_status {
    for (int b : who) {
      ch[b].bf.start(start_from(ch[b]), nt, P->max_iters, v.accurate_ls, v.simple_ascent);
      ch[b].last_eval.assign(start_from(ch[b]), start_from(ch[b]) + nc);
    }
    std::vector<int> active = who;
    while (!active.empty()) {
      mi_status st = evaluate(active, [](Chain &c) { return c.bf.request(); }, true, vcap);
      if (st != MI_OK) return st;
      std::vector<int> next;
      for (size_t i = 0; i < active.size(); i++) {
        Chain &c = ch[active[i]];
        c.last_eval.assign(c.bf.request(), c.bf.request() + nc);  // what `model` holds after this evaluation
        c.bf.feed(e_out[i], &g_out[i * n]);
        c.evals++;
        if (c.bf.state != BfgsChain::Done) next.push_back(active[i]);
      }
      active.swap(next);
    }
    return MI_OK;
  };
  // update_energy (monte_carlo.cpp:44-47): adjust_center on what `model` holds, then ig_metropolis->eval
  auto update_energy = [&](const std::vector<int> &who, float vcap) -> mi_status {
    mi_status st = coords_of(who, [](Chain &c) { return c.mconf.data(); });
    if (st != MI_OK) return st;
    for (size_t i = 0; i < who.size(); i++) {  // DLScorer::set_center_from_model (dl_scorer.cpp:197-217)
      Chain &c = ch[who[i]];
      float s0 = 0, s1 = 0, s2 = 0;
      for (int a : heavy) s0 += xyz[(i * na + a) * 3], s1 += xyz[(i * na + a) * 3 + 1], s2 += xyz[(i * na + a) * 3 + 2];
      c.center[0] = s0 / (float)nh, c.center[1] = s1 / (float)nh, c.center[2] = s2 / (float)nh;
      c.have_center = true;
    }
    return evaluate(who, [](Chain &c) { return c.mconf.data(); }, false, vcap);
  };
  std::vector<int> all(B);
  for (int b = 0; b < B; b++) all[b] = b;
  for (int step = 0; step < P->n_steps; step++) {
    // mutate_conf (mutate.cpp:35-73); the rotation needs the gyration radius of what `model` holds
    std::vector<int> which(B), need;
    for (int b = 0; b < B; b++) {
      ch[b].cand = ch[b].tmp;
      which[b] = ch[b].rng.irange(0, 2 + nt - 1);
      if (which[b] == 1) need.push_back(b);
    }
    if (!need.empty()) {
      mi_status st = coords_of(need, [](Chain &c) { return c.mconf.data(); });
      if (st != MI_OK) return st;
    }
    size_t ni = 0;
    for (int b = 0; b < B; b++) {
      Chain &c = ch[b];
      if (which[b] == 0) {
        float dx, dy, dz;
        c.rng.inside_sphere(dx, dy, dz);
        c.cand[0] += P->mutation_amplitude * dx, c.cand[1] += P->mutation_amplitude * dy, c.cand[2] += P->mutation_amplitude * dz;
      } else if (which[b] == 1) {
        const float *co = &xyz[ni++ * na * 3];
        float acc = 0;  // model::gyration_radius (model.cpp:1002-1014): the LIGAND's heavy atoms about its root origin
        unsigned n_gyr = 0;
        for (int a : heavy) {
          if (a < v.lig.lig_begin || a >= v.lig.lig_end) continue;  // (side-chain atoms are in `heavy` for the container)
          const float dx = co[3 * a] - c.mconf[0], dy = co[3 * a + 1] - c.mconf[1], dz = co[3 * a + 2] - c.mconf[2];
          acc += dx * dx + dy * dy + dz * dz;
          n_gyr++;
        }
        const float gr = n_gyr > 0 ? std::sqrt(acc / (float)n_gyr) : 0.f;
        if (gr > kEps) {
          float dx, dy, dz;
          c.rng.inside_sphere(dx, dy, dz);
          const float sc_ = P->mutation_amplitude / gr;
          const float rot[6] = {0.f, 0.f, 0.f, sc_ * dx, sc_ * dy, sc_ * dz};
          conf_increment(c.cand.data(), rot, 1.0f, 0);
        }
      } else {
        c.cand[7 + (which[b] - 2)] = c.rng.fl(-kPi, kPi);
      }
    }
    mi_status st = minimise(all, [](Chain &c) { return c.cand.data(); }, P->hunt_cap[1]);
    if (st != MI_OK) return st;
    for (int b = 0; b < B; b++) ch[b].cand = ch[b].bf.x, ch[b].mconf = ch[b].last_eval;
    st = update_energy(all, P->authentic_v[1]);
    if (st != MI_OK) return st;
    std::vector<int> again;
    for (int b = 0; b < B; b++) {
      Chain &c = ch[b];
      c.cand_e = e_out[b];
      bool accept = step == 0 || c.cand_e < c.tmp_e;
      if (!accept) accept = c.rng.fl(0.f, 1.f) < std::exp((c.tmp_e - c.cand_e) / P->temperature);  // metropolis_accept
      c.accepted = accept;
      c.second = false;
      if (!accept) continue;
      c.tmp = c.cand, c.tmp_e = c.cand_e, c.mconf = c.tmp;  // m.set(tmp.c)
      if (c.tmp_e < c.best_e || (int)c.out.size() < S) {
        c.second = true;
        again.push_back(b);
      }
    }
    if (again.empty()) continue;
    st = minimise(again, [](Chain &c) { return c.tmp.data(); }, P->authentic_v[1]);
    if (st != MI_OK) return st;
    for (int b : again) ch[b].tmp = ch[b].bf.x, ch[b].mconf = ch[b].last_eval;
    st = update_energy(again, P->authentic_v[1]);
    if (st != MI_OK) return st;
    for (size_t i = 0; i < again.size(); i++) ch[again[i]].tmp_e = e_out[i], ch[again[i]].mconf = ch[again[i]].tmp;
    st = coords_of(again, [](Chain &c) { return c.tmp.data(); });
    if (st != MI_OK) return st;
    for (size_t i = 0; i < again.size(); i++) {  // add_to_output_container (coords.cpp:25-56) + sort
      Chain &c = ch[again[i]];
      Saved sv;
      sv.e = c.tmp_e;
      sv.conf = c.tmp;
      sv.xyz.resize((size_t)nh * 3);
      for (int h = 0; h < nh; h++)
        for (int k = 0; k < 3; k++) sv.xyz[3 * h + k] = xyz[(i * na + heavy[h]) * 3 + k];
      size_t closest = c.out.size();
      float closest_rmsd = 3.402823466e+38f;
      for (size_t o = 0; o < c.out.size(); o++) {
        float acc = 0;
        for (int h = 0; h < nh; h++) {
          const float dx = sv.xyz[3 * h] - c.out[o].xyz[3 * h], dy = sv.xyz[3 * h + 1] - c.out[o].xyz[3 * h + 1],
                      dz = sv.xyz[3 * h + 2] - c.out[o].xyz[3 * h + 2];
          acc += dx * dx + dy * dy + dz * dz;
        }
        const float r = std::sqrt(acc / (float)nh);
        if (o == 0 || r < closest_rmsd) closest = o, closest_rmsd = r;
      }
      if (closest < c.out.size() && closest_rmsd < P->min_rmsd) {
        if (sv.e < c.out[closest].e) c.out[closest] = sv;
      } else if ((int)c.out.size() < S) {
        c.out.push_back(sv);
      } else if (!c.out.empty() && sv.e < c.out.back().e) {
        c.out.back() = sv;
      }
      std::stable_sort(c.out.begin(), c.out.end(), [](const Saved &a, const Saved &b) { return a.e < b.e; });
      if (c.tmp_e < c.best_e) c.best_e = c.tmp_e;
    }
  }
  for (int b = 0; b < B; b++) {
    const Chain &c = ch[b];
    out_n[b] = (int32_t)c.out.size();
    for (size_t o = 0; o < c.out.size(); o++) {
      out_e[(size_t)b * S + o] = c.out[o].e;
      if (out_conf) std::copy(c.out[o].conf.begin(), c.out[o].conf.end(), out_conf + ((size_t)b * S + o) * nc);
      if (out_coords) std::copy(c.out[o].xyz.begin(), c.out[o].xyz.end(), out_coords + ((size_t)b * S + o) * 3 * nh);
    }
    if (evals) evals[b] = (int32_t)c.evals;
  }
  if (cnn_evals) *cnn_evals = (int32_t)std::min<long>(n_cnn, 2147483647L);
  return MI_OK;
  VCATCH_STATUS
}

mi_status mi_cnn_eval_batch(mi_vina *vv, mi_scorer *sc, const float *confs, int B, const mi_cnn_box *box,
                            const float *cnn_centers, int with_deriv, float *energy, float *change) {
  VTRY
  MIG_CHECK(vv && sc && confs && energy && B >= 0 && (!with_deriv || change), 1, "bad arguments");
  MIG_CHECK(!cnn_centers || (box && box->cnn_dimension > 0), 1, "cnn_centers needs box->cnn_dimension");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_lig, 4, "set the ligand first");
  if (B == 0) return MI_OK;
  CnnEvalScratch s;
  return cnn_eval(v, sc, confs, B, box, cnn_centers, box ? box->slope : 0.f, with_deriv, energy, change, s);
  VCATCH_STATUS
}

mi_status mi_cnn_refine_batch(mi_vina *vv, mi_scorer *sc, float *confs, int B, const mi_cnn_box *box, int max_iters,
                              float *energy, int32_t *tries, int32_t *evals) {
  VTRY
  MIG_CHECK(vv && sc && confs && energy && box && B >= 0 && max_iters >= 0, 1, "bad arguments");
  MIG_CHECK(box->cnn_dimension > 0, 1, "box->cnn_dimension must be the CNN grid dimension");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_lig, 4, "set the ligand first");
  if (B == 0) return MI_OK;
  const int na = v.lig.n_atoms, nt = v.lig.n_nodes - 1, n = 6 + nt, nc = 7 + nt;
  CnnEvalScratch scratch;
  // adjust_center (non_cache_cnn.cpp:57-67): the CNN cube is centred on the heavy movable atoms of the
  // starting pose (DLScorer::set_center_from_model, dl_scorer.cpp:197-217) and stays there
  std::vector<float> coords((size_t)B * na * 3), centers((size_t)B * 3);
  mi_status st = mi_vina_coords_batch(vv, confs, B, coords.data());
  if (st != MI_OK) return st;
  auto heavy_center = [&](const float *xyz, float *c) {
    float s0 = 0, s1 = 0, s2 = 0;
    unsigned cnt = 0;
    for (int i = 0; i < v.lig.n_movable; i++)
      if (v.h_lig_smt[i] > 1) s0 += xyz[3 * i], s1 += xyz[3 * i + 1], s2 += xyz[3 * i + 2], cnt++;
    c[0] = s0 / (float)cnt, c[1] = s1 / (float)cnt, c[2] = s2 / (float)cnt;
  };
  for (int b = 0; b < B; b++) heavy_center(&coords[(size_t)b * na * 3], &centers[3 * b]);
  const float margin = 0.0001f, half = box->cnn_dimension / 2.0f;
  auto within = [&](const float *xyz, const float *cen) {  // non_cache_cnn::within = cube OR search box (:74-76)
    bool in_cnn = true, in_box = true;
    for (int i = 0; i < v.lig.n_movable; i++) {
      if (v.h_lig_smt[i] <= 1) continue;
      for (int k = 0; k < 3; k++) {
        const float c = xyz[3 * i + k];
        if (c < cen[k] - half - margin || c > cen[k] + half + margin) in_cnn = false;
        if (box->use_search_box && (c < box->box_begin[k] - margin || c > box->box_end[k] + margin)) in_box = false;
      }
    }
    return in_cnn || in_box;
  };
  // refine_structure (main.cpp:131-171): slope 10, x10 per try, at most 5 tries, until within
  std::vector<BfgsChain> chain(B);
  std::vector<float> slope(B, 10.f);
  std::vector<int> n_try(B, 0), active;
  std::vector<char> done(B, 0);
  std::vector<long> total_evals(B, 0);
  for (int b = 0; b < B; b++) chain[b].start(confs + (size_t)b * nc, nt, max_iters, v.accurate_ls, v.simple_ascent);
  std::vector<float> req, req_centers, e_out, g_out;
  for (;;) {
    // chains sharing a slope value are evaluated together (the slope is a kernel argument)
    active.clear();
    float cur_slope = 0;
    for (int b = 0; b < B; b++)
      if (!done[b]) {
        if (active.empty()) cur_slope = slope[b];
        if (slope[b] == cur_slope) active.push_back(b);
      }
    if (active.empty()) break;
    const int nb = (int)active.size();
    req.resize((size_t)nb * nc), req_centers.resize((size_t)nb * 3), e_out.resize(nb), g_out.resize((size_t)nb * n);
    for (int i = 0; i < nb; i++) {
      std::copy(chain[active[i]].request(), chain[active[i]].request() + nc, req.begin() + (size_t)i * nc);
      std::copy(&centers[3 * active[i]], &centers[3 * active[i]] + 3, req_centers.begin() + (size_t)i * 3);
    }
    st = cnn_eval(v, sc, req.data(), nb, box, req_centers.data(), cur_slope, 1, e_out.data(), g_out.data(), scratch);
    if (st != MI_OK) return st;
    std::vector<int> finished;
    for (int i = 0; i < nb; i++) {
      const int b = active[i];
      chain[b].feed(e_out[i], &g_out[(size_t)i * n]);
      if (chain[b].state == BfgsChain::Done) finished.push_back(b);
    }
    if (finished.empty()) continue;
    // the chains whose BFGS just ended: inside the box? (needs the coordinates of the final conformation)
    std::vector<float> fc((size_t)finished.size() * nc), fxyz((size_t)finished.size() * na * 3);
    for (size_t i = 0; i < finished.size(); i++)
      std::copy(chain[finished[i]].x.begin(), chain[finished[i]].x.end(), fc.begin() + i * nc);
    st = mi_vina_coords_batch(vv, fc.data(), (int)finished.size(), fxyz.data());
    if (st != MI_OK) return st;
    for (size_t i = 0; i < finished.size(); i++) {
      const int b = finished[i];
      total_evals[b] += chain[b].evals;
      n_try[b]++;
      const bool in = within(&fxyz[i * na * 3], &centers[3 * b]);
      if (in || n_try[b] >= 5) {
        done[b] = 1;
        std::copy(chain[b].x.begin(), chain[b].x.end(), confs + (size_t)b * nc);
        energy[b] = in ? chain[b].f0 : 3.402823466e+38f;  // out.e = max_fl (main.cpp:160-161)
      } else {
        slope[b] *= 10;
        std::vector<float> x = chain[b].x;
        chain[b].start(x.data(), nt, max_iters, v.accurate_ls, v.simple_ascent);
      }
    }
  }
  for (int b = 0; b < B; b++) {
    if (tries) tries[b] = n_try[b];
    if (evals) evals[b] = (int32_t)total_evals[b];
  }
  return MI_OK;
  VCATCH_STATUS
}

// do_search's final energies (main.cpp:339-344): intramolecular = eval_intramolecular(exact_prec) and
// e = conf_independent(eval(exact_prec, non_cache) - intramolecular) with num_tors_div (everything.h:796-814)
// do_search's docking branch (main.cpp:231,339-344): e = eval_adjusted(sf, exact_prec, nc_new, ...) where nc_new is a
// non_cache built on the run's LINEAR precalculate (main.cpp:231) -- its receptor term is the table look-up
// p->eval = eval_fast -- while the pair terms of model::eval and eval_intramolecular use exact_prec:
//   e = (non_cache::eval [linear, fast] + ligand pairs [exact]) - intramolecular [exact], then num_tors_div
// (everything.h:796-814: w = 0.1 * (weight + 1) and 1 + w * num_tors / 5.0 are evaluated in double).
static float conf_independent_default(float inter_plus_intra, float intra, float num_tors) {
  const float weight = (float)(5 * 0.05846 / 0.1 - 1);  // main.cpp:1329
  const float w = (float)(0.1 * (weight + 1));
  const float x = inter_plus_intra - intra;
  const float y = (float)(1 + w * num_tors / 5.0);
  // smooth_div, everything.h:52-56
  if (std::fabs(x) < 1.1920928955078125e-07f) return 0;
  if (std::fabs(y) < 1.1920928955078125e-07f) return (x * y > 0) ? 3.402823466e+38f : -3.402823466e+38f;
  return x / y;
}

mi_status mi_vina_final_energies(mi_vina *vv, const float *confs, int B, const float *v3, float num_tors,
                                 float *e_final, float *intramolecular) {
  VTRY
  MIG_CHECK(vv && confs && v3 && e_final && B >= 0, 1, "bad arguments");
  if (B == 0) return MI_OK;
  // e = model::eval(exact_prec, non_cache [linear]) = non_cache::eval + other_pairs + ligand pairs (+ user grid);
  // intramolecular = eval_intramolecular(exact_prec) = ligand pairs + flexible atoms vs rigid receptor + pairs among
  // flexible / inflex atoms (model.cu:352-399).  With a rigid receptor the two pair sums are the same number.
  std::vector<float> inter(B), pairs(B), intra(B);
  mi_status st = mi_vina_eval_batch(vv, confs, B, v3, 2 | MI_VINA_DIRECT | MI_VINA_USER_TERM, inter.data(), nullptr, nullptr);
  if (st != MI_OK) return st;
  st = mi_vina_eval_batch(vv, confs, B, v3, 4 | MI_VINA_EXACT, intra.data(), nullptr, nullptr);
  if (st != MI_OK) return st;
  const Vina &v = *reinterpret_cast<const Vina *>(vv);
  const bool flex = v.lig.n_movable != v.lig.n_atoms || v.lig.pair_cap != nullptr;
  if (flex) {
    st = mi_vina_eval_batch(vv, confs, B, v3, 5 | MI_VINA_EXACT, pairs.data(), nullptr, nullptr);
    if (st != MI_OK) return st;
  } else {
    pairs = intra;
  }
  for (int b = 0; b < B; b++) {
    e_final[b] = conf_independent_default(inter[b] + pairs[b], intra[b], num_tors);
    if (intramolecular) intramolecular[b] = intra[b];
  }
  return MI_OK;
  VCATCH_STATUS
}

mi_status mi_vina_final_energies_screen(mi_vina *vv, const int32_t *item_ligand, const float *confs, int B,
                                        const float *v3, const float *num_tors, float *e_final, float *intramolecular) {
  VTRY
  MIG_CHECK(vv && item_ligand && confs && v3 && num_tors && e_final && B >= 0, 1, "bad arguments");
  if (B == 0) return MI_OK;
  std::vector<float> inter(B), intra(B);
  mi_status st = mi_vina_eval_screen(vv, item_ligand, confs, B, v3, 2 | MI_VINA_DIRECT | MI_VINA_USER_TERM, inter.data(), nullptr, nullptr);
  if (st != MI_OK) return st;
  st = mi_vina_eval_screen(vv, item_ligand, confs, B, v3, 4 | MI_VINA_EXACT, intra.data(), nullptr, nullptr);
  if (st != MI_OK) return st;
  for (int b = 0; b < B; b++) {
    e_final[b] = conf_independent_default(inter[b] + intra[b], intra[b], num_tors[item_ligand[b]]);
    if (intramolecular) intramolecular[b] = intra[b];
  }
  return MI_OK;
  VCATCH_STATUS
}

// do_search's ranking tail (main.cpp:348-361): sort by CNNscore (descending, default), CNNaffinity
// (descending) or Energy (ascending), then remove_redundant (main.cpp:182-192): keep a pose only if its
// RMSD (rmsd_upper_bound over heavy atoms, coords.cpp:25-32) to every pose kept so far exceeds min_rmsd.
// Host only.  order_out receives the indices of the kept poses, best first.
mi_status mi_rank_poses(const float *cnnscore, const float *cnnaffinity, const float *energy, const float *coords,
                        int n_poses, int n_heavy, int sort_order, float min_rmsd, int32_t *order_out,
                        int32_t *n_out) {
  VTRY
  MIG_CHECK(n_poses >= 0 && n_heavy >= 0 && order_out && n_out && (n_poses == 0 || coords), 1, "bad arguments");
  const float *key = sort_order == MI_SORT_ENERGY ? energy : (sort_order == MI_SORT_CNNAFFINITY ? cnnaffinity : cnnscore);
  MIG_CHECK(n_poses == 0 || key, 1, "the array of the requested sort key is NULL");
  std::vector<int> idx(n_poses);
  for (int i = 0; i < n_poses; i++) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
    return sort_order == MI_SORT_ENERGY ? key[a] < key[b] : key[a] > key[b];
  });
  std::vector<int> kept;
  for (int i : idx) {
    int closest = (int)kept.size();
    float best = 3.402823466e+38f;
    for (size_t k = 0; k < kept.size(); k++) {  // find_closest, coords.cpp:34-41
      float acc = 0;
      const float *a = coords + (size_t)i * n_heavy * 3, *b = coords + (size_t)kept[k] * n_heavy * 3;
      for (int h = 0; h < n_heavy; h++) {
        const float dx = a[3 * h] - b[3 * h], dy = a[3 * h + 1] - b[3 * h + 1], dz = a[3 * h + 2] - b[3 * h + 2];
        acc += dx * dx + dy * dy + dz * dz;
      }
      const float r = n_heavy > 0 ? std::sqrt(acc / n_heavy) : 0;
      if (k == 0 || r < best) {
        closest = (int)k;
        best = r;
      }
    }
    if (closest >= (int)kept.size() || best > min_rmsd) kept.push_back(i);
  }
  for (size_t k = 0; k < kept.size(); k++) order_out[k] = kept[k];
  *n_out = (int)kept.size();
  return MI_OK;
  VCATCH_STATUS
}

// merge_output_containers (parallel_mc.cpp:165-181): fold the per-chain containers into one with
// add_to_output_container (coords.cpp:43-56) at min_rmsd (gnina: 2.0) and max_size, then sort.  Host only.
// in_*: [B][S]... as produced by mi_vina_mc_batch; out arrays sized max_size.
mi_status mi_merge_mc_outputs(const int32_t *in_n, const float *in_e, const float *in_conf, const float *in_coords,
                              int B, int S, int conf_len, int n_heavy, float min_rmsd, int max_size, int32_t *out_n,
                              float *out_e, float *out_conf, float *out_coords) {
  VTRY
  MIG_CHECK(in_n && in_e && in_conf && in_coords && out_n && out_e && out_conf && out_coords && B >= 0 && S > 0 &&
                max_size > 0 && conf_len > 0 && n_heavy >= 0,
            1, "bad arguments");
  for (int b = 0; b < B; b++) MIG_CHECK(in_n[b] >= 0 && in_n[b] <= S, 1, "chain output count outside [0, S]");
  struct Ent {
    float e;
    const float *conf, *xyz;
  };
  std::vector<Ent> out;
  auto rmsd = [&](const float *a, const float *b) {
    float acc = 0;
    for (int i = 0; i < 3 * n_heavy; i++) {
      const float d = a[i] - b[i];
      acc += d * d;
    }
    return n_heavy > 0 ? std::sqrt(acc / n_heavy) : 0.f;
  };
  for (int b = 0; b < B; b++)
    for (int o = 0; o < in_n[b]; o++) {
      Ent t{in_e[(size_t)b * S + o], in_conf + ((size_t)b * S + o) * conf_len,
            in_coords + ((size_t)b * S + o) * 3 * n_heavy};
      size_t closest = out.size();
      float best = 3.402823466e+38f;
      for (size_t k = 0; k < out.size(); k++) {
        const float r = rmsd(t.xyz, out[k].xyz);
        if (k == 0 || r < best) {
          closest = k;
          best = r;
        }
      }
      if (closest < out.size() && best < min_rmsd) {
        if (t.e < out[closest].e) out[closest] = t;
      } else if ((int)out.size() < max_size) {
        out.push_back(t);
      } else if (!out.empty() && t.e < out.back().e) {
        out.back() = t;
      }
      std::stable_sort(out.begin(), out.end(), [](const Ent &x, const Ent &y) { return x.e < y.e; });
    }
  *out_n = (int)out.size();
  for (size_t k = 0; k < out.size(); k++) {
    out_e[k] = out[k].e;
    std::memcpy(out_conf + k * conf_len, out[k].conf, sizeof(float) * conf_len);
    std::memcpy(out_coords + k * 3 * n_heavy, out[k].xyz, sizeof(float) * 3 * n_heavy);
  }
  return MI_OK;
  VCATCH_STATUS
}

// latency probe: ms for `reps` dependent evaluations (mode as in mi_vina_eval_batch, 3 = coordinates only)
mi_status mi_vina_eval_latency(mi_vina *vv, const float *confs, int B, int mode, int reps, float *ms_out) {
  VTRY
  MIG_CHECK(vv && confs && B > 0 && reps > 0 && ms_out, 1, "bad arguments");
  Vina &v = *reinterpret_cast<Vina *>(vv);
  MIG_CHECK(v.have_cache && v.have_lig, 4, "build the cache and set the ligand first");
  const int nc = 7 + v.lig.n_nodes - 1;
  v.d_confs.upload(confs, (size_t)B * nc, v.stream);
  v.d_energy.ensure(B);
  const int tmode = mode == 1 ? 0 : (mode == 0 ? 1 : mode);  // ABI mode -> kernel template mode
  hipEvent_t e0, e1;
  MIG_HIP(hipEventCreate(&e0));
  MIG_HIP(hipEventCreate(&e1));
  launch_vina_eval_repeat(make_env(v), v.lig, v.d_confs.p, B, tmode, 2, v.d_energy.p, v.stream);  // warm-up
  MIG_HIP(hipEventRecord(e0, v.stream));
  launch_vina_eval_repeat(make_env(v), v.lig, v.d_confs.p, B, tmode, reps, v.d_energy.p, v.stream);
  MIG_HIP(hipEventRecord(e1, v.stream));
  MIG_HIP(hipStreamSynchronize(v.stream));
  MIG_HIP(hipEventElapsedTime(ms_out, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return MI_OK;
  VCATCH_STATUS
}

void *mi_vina_stream(mi_vina *vv) { return vv ? (void *)reinterpret_cast<Vina *>(vv)->stream : nullptr; }

}  // extern "C"
