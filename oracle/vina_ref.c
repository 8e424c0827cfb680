/*
 * oracle/vina_ref.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar fp32 like gnina's `fl` = float, common.h:47; no FMA
 * contraction) of the smina/Vina scoring + local optimisation path that gnina's Monte-Carlo
 * docking loop runs, each function citing the reference file:line
 * it follows under /root/reference/gninasrc/lib/.
 *
 * Parity status: "parity unpinned" by stored vectors -- the reference keeps NO golden numbers
 * for this path; its own tests only assert CPU == CUDA within 0.01 on random
 * molecules (test/gnina/test_gpucode.cpp:142-147, test_cache.cu:148-153, test_tree.cu:175-187)
 * and a few inequalities.  The source cannot be compiled here (Boost, OpenBabel, CUDA headers),
 * so this restatement of the cited lines is the oracle; tests/test_oracle_vina.py checks it
 * against analytic identities (finite-difference gradients, rigid-motion invariance, table
 * construction rules) and the parity tests hold the HIP kernels to it.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NT 28
#define V_PI 3.14159265358979323846f /* common.h: const fl pi = fl(3.1415926535897931) */
#define V_EPS 1.1920928955078125e-07f /* std::numeric_limits<float>::epsilon(), common.h:328 */
#define V_MAXFL 3.402823466e+38f

/* atom_constants.h:101-133: xs_radius and the xs_hydrophobe / xs_donor / xs_acceptor flags */
static const float XS_R[NT] = {0.37f, 0.37f, 1.9f, 1.9f, 1.9f, 1.9f, 1.8f, 1.8f, 1.8f, 1.8f, 1.7f, 1.7f, 1.7f, 1.7f,
                               2.0f,  2.0f,  2.1f, 1.5f, 1.8f, 2.0f, 2.2f, 1.2f, 1.2f, 1.2f, 1.2f, 1.2f, 1.2f, 1.92f};
static const unsigned char XS_HYD[NT] = {0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1};
static const unsigned char XS_DON[NT] = {0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 0};
static const unsigned char XS_ACC[NT] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static inline int is_hydrogen(int t) { return t == 0 || t == 1; }

/* default weights, main.cpp:1324-1328 (stored as fl) */
static const float DEFAULT_W[5] = {-0.035579f, -0.005156f, 0.840245f, -0.035069f, -0.587439f};

/* everything.h:207-216 */
static float slope_step(float x_bad, float x_good, float x) {
  if (x_bad < x_good) {
    if (x <= x_bad) return 0;
    if (x >= x_good) return 1;
  } else {
    if (x >= x_bad) return 0;
    if (x <= x_good) return 1;
  }
  return (x - x_bad) / (x_good - x_bad);
}

/* everything.h:48-50 */
static float gaussian(float x, float width) {
  float q = x / width;
  return expf(-(q * q));
}

/* weighted_terms::eval_fast (weighted_terms.cpp:54-68) with the five default terms:
 * gauss(o=0,w=0.5) everything.h:153-179, gauss(o=3,w=2), repulsion(o=0) :181-205,
 * hydrophobic(g=0.5,b=1.5) :218-247, non_dir_h_bond(g=-0.7,b=0) :480-506; all cutoff 8. */
float ora_vina_pair_energy(const float *w, int t1, int t2, float r) {
  if (!w) w = DEFAULT_W;
  float opt = XS_R[t1] + XS_R[t2]; /* optimal_distance, everything.h:149-151 */
  float acc = 0;
  acc += w[0] * gaussian(r - (opt + 0.0f), 0.5f);
  acc += w[1] * gaussian(r - (opt + 3.0f), 2.0f);
  {
    float d = r - (opt + 0.0f);
    acc += w[2] * (d > 0 ? 0.0f : d * d);
  }
  acc += w[3] * ((XS_HYD[t1] && XS_HYD[t2]) ? slope_step(1.5f, 0.5f, r - opt) : 0.0f);
  {
    int hb = (XS_DON[t1] && XS_ACC[t2]) || (XS_DON[t2] && XS_ACC[t1]); /* atom_constants.h:195-201 */
    acc += w[4] * (hb ? slope_step(0.0f, -0.7f, r - opt) : 0.0f);
  }
  return acc;
}

/* ---------------------------------------------------------------------------------------------
 * precalculate_linear (precalculate.h:165-272, element :82-163)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int n;            /* sz(factor * cutoff_sqr) + 3 */
  float factor, cutoff_sqr;
  float *rs;        /* [n+2] sqrt(i / factor), precalculate.h:266-271 */
  float *fast;      /* [npairs][n] */
  float *se, *sd;   /* smooth (E, dor) [npairs][n] */
} ora_vina_tables;

static inline int tri(int t1, int t2) { /* triangular_matrix index, i <= j */
  if (t1 > t2) {
    int t = t1;
    t1 = t2;
    t2 = t;
  }
  return t1 + t2 * (t2 + 1) / 2;
}

ora_vina_tables *ora_vina_tables_create(const float *w, float cutoff, float factor) {
  ora_vina_tables *T = (ora_vina_tables *)calloc(1, sizeof(*T));
  T->factor = factor;
  T->cutoff_sqr = cutoff * cutoff;
  T->n = (int)(factor * T->cutoff_sqr) + 3;
  const int n = T->n, np = NT * (NT + 1) / 2;
  T->rs = (float *)malloc(sizeof(float) * (n + 2));
  for (int i = 0; i < n + 2; i++) T->rs[i] = sqrtf((float)i / factor);
  T->fast = (float *)calloc((size_t)np * n, sizeof(float));
  T->se = (float *)calloc((size_t)np * n, sizeof(float));
  T->sd = (float *)calloc((size_t)np * n, sizeof(float));
  for (int t1 = 0; t1 < NT; t1++)
    for (int t2 = t1; t2 < NT; t2++) {
      float *se = T->se + (size_t)tri(t1, t2) * n, *sd = T->sd + (size_t)tri(t1, t2) * n;
      float *fa = T->fast + (size_t)tri(t1, t2) * n;
      for (int i = 0; i < n; i++) se[i] = ora_vina_pair_energy(w, t1, t2, T->rs[i]);
      /* init_from_smooth_fst, precalculate.h:135-158 */
      for (int i = 0; i < n; i++) {
        if (i == 0 || i == n - 1) {
          sd[i] = 0;
        } else {
          float delta = T->rs[i + 1] - T->rs[i - 1];
          float r = T->rs[i];
          sd[i] = (se[i + 1] - se[i - 1]) / (delta * r);
        }
        float f1 = se[i];
        float f2 = (i + 1 >= n) ? 0 : se[i + 1];
        fa[i] = (f2 + f1) / 2;
      }
    }
  return T;
}

void ora_vina_tables_free(ora_vina_tables *T) {
  if (!T) return;
  free(T->rs);
  free(T->fast);
  free(T->se);
  free(T->sd);
  free(T);
}

int ora_vina_tables_n(const ora_vina_tables *T) { return T->n; }

void ora_vina_tables_get(const ora_vina_tables *T, int t1, int t2, float *fast, float *se, float *sd) {
  size_t o = (size_t)tri(t1, t2) * T->n;
  if (fast) memcpy(fast, T->fast + o, sizeof(float) * T->n);
  if (se) memcpy(se, T->se + o, sizeof(float) * T->n);
  if (sd) memcpy(sd, T->sd + o, sizeof(float) * T->n);
}

/* precalculate_linear_element::eval_fast, precalculate.h:90-95 */
/* ---------------------------------------------------------------------------------------------
 * precalculate_splines (precalculate.h:277-449) + Spline (splines.h): --approximation spline, the default of --minimize.
 * Process-wide switch of this test library (like the user grid): while on, the two table entry points below evaluate
 * the spline instead.  Spline::initialize inverts the dense (n + 1)^2 system in fp32 with Eigen (third-party, not
 * restated: its LU rounding is not part of gnina); the system is tridiagonal, so it is solved directly here
 * (elimination in double, coefficients rounded to fp32 in the reference's formulas).  Agreement with oracle/_ref:
 * ~1e-6 of the spline's scale (tests/test_ref_vina.py), not bit for bit.
 * ------------------------------------------------------------------------------------------- */
static struct {
  int on, n;
  float fraction, cutoff;
  float *k; /* [npairs][n][4] a b c d */
} g_sp;
void ora_vina_set_approximation(int kind, float factor, float cutoff, const float *w) {
  free(g_sp.k);
  g_sp.k = NULL;
  g_sp.on = 0;
  if (kind != 1) return;
  const unsigned n = (unsigned)(factor * cutoff);
  const float fraction = cutoff / (float)n;
  const int np = NT * (NT + 1) / 2;
  g_sp.k = (float *)calloc((size_t)np * n * 4, sizeof(float));
  float *x = (float *)malloc(sizeof(float) * (n + 1)), *y = (float *)malloc(sizeof(float) * (n + 1));
  float *C = (float *)malloc(sizeof(float) * (n + 1)), *ddy = (float *)malloc(sizeof(float) * (n + 1));
  double *cp = (double *)malloc(sizeof(double) * (n + 1)), *dp = (double *)malloc(sizeof(double) * (n + 1));
  for (unsigned i = 0; i < n; i++) x[i] = (float)i * fraction;
  x[n] = cutoff;
  const float hlast = x[n] - x[n - 1];
  const unsigned e = n;
  for (int t1 = 0; t1 < NT; t1++)
    for (int t2 = t1; t2 < NT; t2++) {
      int nonzero = 0;
      for (unsigned i = 0; i < n; i++) {
        y[i] = ora_vina_pair_energy(w, t1, t2, x[i]);
        if (y[i] != 0) nonzero = 1;
      }
      y[n] = 0;
      if (!nonzero) continue;
      C[0] = 6 * ((y[1] - y[0]) / fraction);
      for (unsigned i = 1; i < e; i++) {
        float hi = i == e - 1 ? hlast : fraction;
        C[i] = 6 * ((y[i + 1] - y[i]) / hi - (y[i] - y[i - 1]) / fraction);
      }
      C[e] = 6 * (-(y[e] - y[e - 1]) / hlast);
      /* column i of the reference's matrix: hi ddy[i-1] + 2 (fraction + hi) ddy[i] + hi ddy[i+1] = C[i] */
      cp[0] = 0.5; /* fraction / (2 fraction) */
      dp[0] = C[0] / (2.0 * fraction);
      for (unsigned i = 1; i <= e; i++) {
        double hi = i >= e - 1 ? hlast : fraction;
        double diag = i == e ? 2.0 * hlast : 2.0 * ((double)fraction + hi);
        double m = diag - hi * cp[i - 1];
        cp[i] = i < e ? hi / m : 0.0;
        dp[i] = (C[i] - hi * dp[i - 1]) / m;
      }
      double nxt = dp[e];
      ddy[e] = (float)nxt;
      for (int i = (int)e - 1; i >= 0; i--) {
        nxt = dp[i] - cp[i] * nxt;
        ddy[i] = (float)nxt;
      }
      float *out = g_sp.k + (size_t)tri(t1, t2) * n * 4;
      for (unsigned i = 0; i < e; i++) {
        float hi = i == e - 1 ? hlast : fraction;
        out[4 * i] = (ddy[i + 1] - ddy[i]) / (6 * hi);
        out[4 * i + 1] = ddy[i] / 2;
        out[4 * i + 2] = (y[i + 1] - y[i]) / hi - ddy[i + 1] * hi / 6 - ddy[i] * hi / 3;
        out[4 * i + 3] = y[i];
      }
    }
  free(x), free(y), free(C), free(ddy), free(cp), free(dp);
  g_sp.n = (int)n;
  g_sp.fraction = fraction;
  g_sp.cutoff = cutoff;
  g_sp.on = 1;
}
/* Spline::eval_deriv (splines.h:100-118) */
static void spline_eval(int t1, int t2, float r, float *val, float *dx) {
  *val = *dx = 0;
  if (r >= g_sp.cutoff) return;
  int idx = (int)(r / g_sp.fraction);
  if (idx > g_sp.n - 1) idx = g_sp.n - 1;
  const float *k = g_sp.k + ((size_t)tri(t1, t2) * g_sp.n + idx) * 4;
  float lx = r - (float)idx * g_sp.fraction;
  *val = ((k[0] * lx + k[1]) * lx + k[2]) * lx + k[3];
  *dx = (3 * k[0] * lx + 2 * k[1]) * lx + k[2];
}

float ora_vina_eval_fast(const ora_vina_tables *T, int t1, int t2, float r2) {
  if (g_sp.on) { /* precalculate_splines::eval_fast, precalculate.h:407-411 */
    float e, dx;
    spline_eval(t1, t2, sqrtf(r2), &e, &dx);
    return e;
  }
  int i = (int)(T->factor * r2);
  return T->fast[(size_t)tri(t1, t2) * T->n + i];
}

/* precalculate_linear_element::eval_deriv, precalculate.h:97-133 (single component) */
void ora_vina_table_eval_deriv(const ora_vina_tables *T, int t1, int t2, float r2, float *e, float *dor) {
  if (g_sp.on) { /* precalculate_splines::eval_deriv, precalculate.h:413-442 (no slow terms in the default set) */
    float r = sqrtf(r2), dx;
    spline_eval(t1, t2, r, e, &dx);
    *dor = dx / r;
    return;
  }
  float r2f = T->factor * r2;
  int i1 = (int)r2f, i2 = i1 + 1;
  float rem = r2f - (float)i1;
  const float *se = T->se + (size_t)tri(t1, t2) * T->n, *sd = T->sd + (size_t)tri(t1, t2) * T->n;
  float e1 = se[i1], e2 = se[i2], d1 = sd[i1], d2 = sd[i2];
  *e = e1 + rem * (e2 - e1);
  *dor = d1 + rem * (d2 - d1);
}

/* curl.h:29-42 */
static void curl3(float *e, float *d, float v) {
  if (*e > 0 && v < 0.1f * V_MAXFL) {
    float tmp = (v < V_EPS) ? 0 : (v / (v + *e));
    *e *= tmp;
    float t2 = tmp * tmp;
    d[0] *= t2;
    d[1] *= t2;
    d[2] *= t2;
  }
}
static void curl1(float *e, float v) {
  if (*e > 0 && v < 0.1f * V_MAXFL) {
    float tmp = (v < V_EPS) ? 0 : (v / (v + *e));
    *e *= tmp;
  }
}

/* ---------------------------------------------------------------------------------------------
 * receptor grids: cache::populate (cache.cpp:104-184), grid::init (grid.cpp:47-68),
 * grid::evaluate_aux (grid.cpp:96-186); array3d is x-fastest (array3d.h:91-96)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  float begin[3], end[3];
  int n[3]; /* intervals; points = n + 1 */
} ora_grid_dims;

/* setup_grid_dims, main.cpp:622-634 (box_granularity 0.375) */
void ora_vina_setup_grid_dims(const float *center, const float *size, ora_grid_dims *gd) {
  const float gran = 0.375f;
  for (int i = 0; i < 3; i++) {
    gd->n[i] = (int)ceilf(size[i] / gran);
    float real_span = gran * (float)gd->n[i];
    gd->begin[i] = center[i] - real_span / 2;
    gd->end[i] = gd->begin[i] + real_span;
  }
}

typedef struct {
  float init[3], factor[3], factor_inv[3], dim_m1[3];
  int dim[3];
} grid_geom;

static grid_geom geom_of(const ora_grid_dims *gd) {
  grid_geom g;
  for (int i = 0; i < 3; i++) {
    g.dim[i] = gd->n[i] + 1;
    g.init[i] = gd->begin[i];
    float range = gd->end[i] - gd->begin[i];
    g.dim_m1[i] = (float)g.dim[i] - 1.0f;
    g.factor[i] = g.dim_m1[i] / range;
    g.factor_inv[i] = 1 / g.factor[i];
  }
  return g;
}

/* One ligand-type grid.  data: [(nz+1)][(ny+1)][(nx+1)] with x fastest.  Sum over receptor atoms in
 * index order (szv_grid possibilities are index ordered), r2 <= cutoff_sqr.  Receptor hydrogens take no part:
 * cache::populate and non_cache::eval only ever see the atoms szv_grid hands out, and szv_grid drops
 * hydrogens (szv_grid.h compute_relevant / get: `!a.is_hydrogen() && a.acceptable_type()`).  [Found by pinning
 * this file against oracle/_ref: round 1 had them in.] */
/* szv_grid_cache::get (szv_grid.h:107-144): the candidate list of a 3 A cell is built the FIRST time a point
 * falls into the cell, from the brick [floor(c/3)*3, ceil(c/3)*3] of THAT point's coordinates -- a degenerate brick
 * (a plane or a line) when the coordinate is an exact multiple of 3 -- and reused for every later point of the
 * cell.  cache::populate walks x, then y, then z upwards, so the first point of a cell is its lowest lattice point
 * per dimension: per dimension and lattice index, the brick bounds inherited from that point. */
void ora_vina_cell_bricks(const ora_grid_dims *gd, int d, float *lo, float *hi) {
  grid_geom g = geom_of(gd);
  const float gran = 3.0f;
  float cur_cell = 0, cur_lo = 0, cur_hi = 0;
  for (int i = 0; i < g.dim[d]; i++) {
    float c = g.init[d] + g.factor_inv[d] * (float)i;
    float cell = floorf(c / gran);
    if (i == 0 || cell != cur_cell) {
      cur_cell = cell;
      cur_lo = floorf(c / gran) * gran;
      cur_hi = ceilf(c / gran) * gran;
    }
    lo[i] = cur_lo;
    hi[i] = cur_hi;
  }
}

static float closest_between(float b, float e, float x) { return x <= b ? b : (x >= e ? e : x); } /* brick.h:28-35 */

float ora_vina_grid_evaluate(const ora_grid_dims *gd, const float *data, const float *loc, float slope, float v,
                             float *deriv);

/* --user_grid (main.cpp:1342-1350; grid::init(gd, user_in, scale) grid.cpp:69-92): process-wide state of this test
 * library, set by the test before it builds grids / evaluates.  values: the file's numbers, [nz][ny][nx]; the
 * (n + 1)^3 array keeps 0 in the last plane of every dimension.  values == NULL removes it. */
static struct {
  int on;
  ora_grid_dims gd;
  float *data;
  float cache_slope; /* slope of the cache the grid is baked into (cache::populate passes its own) */
} g_ug;
void ora_vina_set_user_grid(const float *begin, const float *end, const int *n, const double *values, float scale,
                            float cache_slope) {
  free(g_ug.data);
  g_ug.data = NULL;
  g_ug.on = 0;
  if (!values) return;
  for (int i = 0; i < 3; i++) g_ug.gd.begin[i] = begin[i], g_ug.gd.end[i] = end[i], g_ug.gd.n[i] = n[i];
  const size_t d0 = n[0] + 1, d1 = n[1] + 1, d2 = n[2] + 1;
  g_ug.data = (float *)calloc(d0 * d1 * d2, sizeof(float));
  size_t k = 0;
  for (int z = 0; z < n[2]; z++)
    for (int y = 0; y < n[1]; y++)
      for (int x = 0; x < n[0]; x++, k++) g_ug.data[x + d0 * (y + d1 * (size_t)z)] = (float)(-(values[k] * scale));
  g_ug.cache_slope = cache_slope;
  g_ug.on = 1;
}
/* grid::evaluate_user (grid.cpp:47-49) */
static float user_grid_eval(const float *loc, float slope, float *deriv) {
  return ora_vina_grid_evaluate(&g_ug.gd, g_ug.data, loc, slope, 1000.0f, deriv);
}

void ora_vina_cache_populate(const ora_vina_tables *T, const ora_grid_dims *gd, const float *rec_xyz,
                             const int32_t *rec_smt, int n_rec, int lig_type, float *data) {
  grid_geom g = geom_of(gd);
  float *blo[3], *bhi[3];
  for (int d = 0; d < 3; d++) {
    blo[d] = (float *)malloc(sizeof(float) * g.dim[d]);
    bhi[d] = (float *)malloc(sizeof(float) * g.dim[d]);
    ora_vina_cell_bricks(gd, d, blo[d], bhi[d]);
  }
  for (int z = 0; z < g.dim[2]; z++)
    for (int y = 0; y < g.dim[1]; y++)
      for (int x = 0; x < g.dim[0]; x++) {
        float px = g.init[0] + g.factor_inv[0] * (float)x; /* index_to_argument, grid.h:54-57 */
        float py = g.init[1] + g.factor_inv[1] * (float)y;
        float pz = g.init[2] + g.factor_inv[2] * (float)z;
        float aff = 0;
        for (int i = 0; i < n_rec; i++) {
          if (is_hydrogen(rec_smt[i])) continue;
          const float *a = rec_xyz + 3 * i;
          float cx = closest_between(blo[0][x], bhi[0][x], a[0]) - a[0];
          float cy = closest_between(blo[1][y], bhi[1][y], a[1]) - a[1];
          float cz = closest_between(blo[2][z], bhi[2][z], a[2]) - a[2];
          if (!(cx * cx + cy * cy + cz * cz < T->cutoff_sqr)) continue; /* not in the cell's candidate list */
          float dx = a[0] - px, dy = a[1] - py, dz = a[2] - pz;
          float r2 = dx * dx + dy * dy + dz * dz; /* vec_distance_sqr: sqr(x)+sqr(y)+sqr(z) */
          if (r2 <= T->cutoff_sqr) aff += ora_vina_eval_fast(T, rec_smt[i], lig_type, r2);
        }
        if (g_ug.on) { /* cache.cpp:177-179: the lattice INDICES are the location handed to evaluate_user */
          const float idx[3] = {(float)x, (float)y, (float)z};
          aff += user_grid_eval(idx, g_ug.cache_slope, NULL);
        }
        data[(size_t)x + (size_t)g.dim[0] * ((size_t)y + (size_t)g.dim[1] * z)] = aff;
      }
  for (int d = 0; d < 3; d++) {
    free(blo[d]);
    free(bhi[d]);
  }
}

/* grid::evaluate_aux (grid.cpp:96-186). deriv may be NULL. */
float ora_vina_grid_evaluate(const ora_grid_dims *gd, const float *data, const float *loc, float slope, float v,
                             float *deriv) {
  grid_geom g = geom_of(gd);
  float s[3], miss[3] = {0, 0, 0};
  int region[3], a[3];
  for (int i = 0; i < 3; i++) {
    s[i] = (loc[i] - g.init[i]) * g.factor[i];
    if (s[i] < 0) {
      miss[i] = -s[i];
      region[i] = -1;
      a[i] = 0;
      s[i] = 0;
    } else if (s[i] >= g.dim_m1[i]) {
      miss[i] = s[i] - g.dim_m1[i];
      region[i] = 1;
      a[i] = g.dim[i] - 2;
      s[i] = 1;
    } else {
      region[i] = 0;
      a[i] = (int)s[i];
      s[i] -= (float)a[i];
    }
  }
  const float penalty = slope * (miss[0] * g.factor_inv[0] + miss[1] * g.factor_inv[1] + miss[2] * g.factor_inv[2]);
#define D(X, Y, Z) data[(size_t)(X) + (size_t)g.dim[0] * ((size_t)(Y) + (size_t)g.dim[1] * (Z))]
  const int x0 = a[0], y0 = a[1], z0 = a[2], x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
  const float f000 = D(x0, y0, z0), f100 = D(x1, y0, z0), f010 = D(x0, y1, z0), f110 = D(x1, y1, z0);
  const float f001 = D(x0, y0, z1), f101 = D(x1, y0, z1), f011 = D(x0, y1, z1), f111 = D(x1, y1, z1);
#undef D
  const float x = s[0], y = s[1], z = s[2], mx = 1 - x, my = 1 - y, mz = 1 - z;
  float f = f000 * mx * my * mz + f100 * x * my * mz + f010 * mx * y * mz + f110 * x * y * mz + f001 * mx * my * z +
            f101 * x * my * z + f011 * mx * y * z + f111 * x * y * z;
  if (deriv) {
    float gr[3];
    gr[0] = f000 * (-1) * my * mz + f100 * 1 * my * mz + f010 * (-1) * y * mz + f110 * 1 * y * mz +
            f001 * (-1) * my * z + f101 * 1 * my * z + f011 * (-1) * y * z + f111 * 1 * y * z;
    gr[1] = f000 * mx * (-1) * mz + f100 * x * (-1) * mz + f010 * mx * 1 * mz + f110 * x * 1 * mz +
            f001 * mx * (-1) * z + f101 * x * (-1) * z + f011 * mx * 1 * z + f111 * x * 1 * z;
    gr[2] = f000 * mx * my * (-1) + f100 * x * my * (-1) + f010 * mx * y * (-1) + f110 * x * y * (-1) +
            f001 * mx * my * 1 + f101 * x * my * 1 + f011 * mx * y * 1 + f111 * x * y * 1;
    curl3(&f, gr, v);
    for (int i = 0; i < 3; i++) {
      float ge = (region[i] == 0) ? gr[i] : 0;
      deriv[i] = g.factor[i] * ge + slope * (float)region[i];
    }
    return f + penalty;
  }
  curl1(&f, v);
  return f + penalty;
}

/* ---------------------------------------------------------------------------------------------
 * ligand torsion tree: conf -> coordinates, forces -> change
 * (tree.h:29-59,123-140,152-203,206-233,293-401; quaternion.h:243-257,284-303,327-364)
 * Nodes are stored in DFS pre-order; node 0 is the rigid root, node k>0 owns torsion k-1
 * (the iterator order of branches_set_conf / branches_derivative).
 * conf   = [pos 3][quat a,b,c,d][torsion x (n_nodes-1)]
 * change = [force 3][torque 3][torsion derivative x (n_nodes-1)]
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int n_atoms;
  const int32_t *smt;      /* [n_atoms] */
  const float *local_xyz;  /* [n_atoms][3] coordinates in the owning node's frame */
  int n_nodes;
  const int32_t *parent;   /* [n_nodes], parent[0] = -1 */
  const int32_t *abeg, *aend; /* atom range of each node */
  const float *rel_origin; /* [n_nodes][3] (node 0 unused) */
  const float *rel_axis;   /* [n_nodes][3] */
  int n_pairs;
  const int32_t *pairs;    /* [n_pairs][2] intramolecular interacting pairs (model.cpp:682-703) */
  /* Flexible receptor residues (model.h: atoms = [flex movable | ligand | inflex]; tree.h:266-283,374-393):
   *  - a node with parent -2 is a residue's first_segment: its frame hangs off the world (rel_origin / rel_axis are
   *    absolute, no rigid-body entries), conf = [7 + T_ligand + T_flex];
   *  - atoms outside every node are inflex: fixed at local_xyz, partners in pairs only;
   *  - n_movable: atoms [0, n_movable) get the receptor term and keep forces (0 = all atoms);
   *  - pair_kind [n_pairs]: 0 = ligand-internal (cap v[0]), 1 = model::other_pairs (cap v[2]); NULL = all 0.  Pairs
   *    of kind 1 come first (model::eval_deriv adds other_pairs, then the ligand's; model.cu:209-216);
   *  - [lig_begin, lig_end): the ligand's atoms (gyration radius); 0, 0 = all atoms. */
  int n_movable;
  const int32_t *pair_kind;
  int lig_begin, lig_end;
} ora_ligand;

static int n_movable_of(const ora_ligand *L) { return L->n_movable > 0 ? L->n_movable : L->n_atoms; }

static void g_normalize_angle(float *x) { /* quaternion.h:259-282 */
  if (*x > 3 * V_PI) {
    float n = (*x - V_PI) / (2 * V_PI);
    *x -= 2 * V_PI * ceilf(n);
    g_normalize_angle(x);
  } else if (*x < -3 * V_PI) {
    float n = (-*x - V_PI) / (2 * V_PI);
    *x += 2 * V_PI * ceilf(n);
    g_normalize_angle(x);
  } else if (*x > V_PI) {
    *x -= 2 * V_PI;
  } else if (*x < -V_PI) {
    *x += 2 * V_PI;
  }
}

static void angle_to_quat(const float *axis, float angle, float *q) { /* quaternion.h:284-291 */
  g_normalize_angle(&angle);
  float c = cosf(angle / 2), s = sinf(angle / 2);
  q[0] = c;
  q[1] = s * axis[0];
  q[2] = s * axis[1];
  q[3] = s * axis[2];
}

static void quat_mul(const float *l, const float *r, float *o) { /* quaternion.h:293-303 */
  const float a = l[0], b = l[1], c = l[2], d = l[3], ar = r[0], br = r[1], cr = r[2], dr = r[3];
  o[0] = +a * ar - b * br - c * cr - d * dr;
  o[1] = +a * br + b * ar + c * dr - d * cr;
  o[2] = +a * cr - b * dr + c * ar + d * br;
  o[3] = +a * dr + b * cr - c * br + d * ar;
}

static void quat_normalize_approx(float *q) { /* quaternion.h:243-257, tolerance 1e-6 */
  const float s = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (fabsf(s - 1) < 1e-6f) return;
  const float a = sqrtf(s);
  const float inv = 1 / a;
  q[0] *= inv;
  q[1] *= inv;
  q[2] *= inv;
  q[3] *= inv;
}

static void quat_to_r3(const float *q, float *m) { /* quaternion.h:327-364; m[i + 3 j] = M(i,j) */
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

static void mat_vec(const float *m, const float *v, float *o) { /* common.h:243-246 */
  o[0] = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  o[1] = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  o[2] = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
}

/* heterotree<rigid_body>::set_conf (tree.h:361-367) -> rigid_body::set_conf (:152-156),
 * segment::set_conf (:218-233), atom_frame::set_coords (:128-131).
 * Outputs: coords [n_atoms][3], node origins [n_nodes][3], node axes [n_nodes][3]. */
void ora_vina_set_conf(const ora_ligand *L, const float *conf, float *coords, float *origin, float *axis) {
  float *q = (float *)malloc(sizeof(float) * 4 * L->n_nodes);
  float *M = (float *)malloc(sizeof(float) * 9 * L->n_nodes);
  /* inflex atoms belong to no node: they stay where the file has them */
  memcpy(coords, L->local_xyz, sizeof(float) * 3 * (size_t)L->n_atoms);
  for (int k = 0; k < L->n_nodes; k++) {
    if (k == 0) {
      origin[0] = conf[0];
      origin[1] = conf[1];
      origin[2] = conf[2];
      memcpy(q, conf + 3, sizeof(float) * 4); /* set_orientation does not normalize (tree.h:53-56) */
      axis[0] = axis[1] = axis[2] = 0;
    } else if (L->parent[k] == -2) { /* first_segment::set_conf (tree.h:272-277): fixed origin and axis */
      memcpy(origin + 3 * k, L->rel_origin + 3 * k, sizeof(float) * 3);
      memcpy(axis + 3 * k, L->rel_axis + 3 * k, sizeof(float) * 3);
      angle_to_quat(axis + 3 * k, conf[7 + (k - 1)], q + 4 * k);
    } else {
      int p = L->parent[k];
      float t[3];
      mat_vec(M + 9 * p, L->rel_origin + 3 * k, t); /* local_to_lab: origin + M * local (tree.h:34-38) */
      origin[3 * k + 0] = origin[3 * p + 0] + t[0];
      origin[3 * k + 1] = origin[3 * p + 1] + t[1];
      origin[3 * k + 2] = origin[3 * p + 2] + t[2];
      mat_vec(M + 9 * p, L->rel_axis + 3 * k, axis + 3 * k);
      float rq[4];
      angle_to_quat(axis + 3 * k, conf[7 + (k - 1)], rq);
      quat_mul(rq, q + 4 * p, q + 4 * k);
      quat_normalize_approx(q + 4 * k);
    }
    quat_to_r3(q + 4 * k, M + 9 * k);
    for (int i = L->abeg[k]; i < L->aend[k]; i++) {
      float t[3];
      mat_vec(M + 9 * k, L->local_xyz + 3 * i, t);
      coords[3 * i + 0] = origin[3 * k + 0] + t[0];
      coords[3 * i + 1] = origin[3 * k + 1] + t[1];
      coords[3 * i + 2] = origin[3 * k + 2] + t[2];
    }
  }
  free(q);
  free(M);
}

static void cross(const float *a, const float *b, float *o) { /* common.h:198-200 */
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* tree<segment>::derivative (tree.h:328-338) / branches_derivative (:301-311): returns the
 * (force, torque) of node k's subtree about node k's origin, writes the torsion derivative. */
static void node_derivative(const ora_ligand *L, int k, const float *coords, const float *forces,
                            const float *origin, const float *axis, float *change, float *ft) {
  float f[3] = {0, 0, 0}, tq[3] = {0, 0, 0};
  for (int i = L->abeg[k]; i < L->aend[k]; i++) { /* sum_force_and_torque, tree.h:133-140 */
    float r[3] = {coords[3 * i] - origin[3 * k], coords[3 * i + 1] - origin[3 * k + 1],
                  coords[3 * i + 2] - origin[3 * k + 2]};
    float c[3];
    cross(r, forces + 3 * i, c);
    f[0] += forces[3 * i];
    f[1] += forces[3 * i + 1];
    f[2] += forces[3 * i + 2];
    tq[0] += c[0];
    tq[1] += c[1];
    tq[2] += c[2];
  }
  for (int c = k + 1; c < L->n_nodes; c++) {
    if (L->parent[c] != k) continue;
    float cft[6];
    node_derivative(L, c, coords, forces, origin, axis, change, cft);
    f[0] += cft[0];
    f[1] += cft[1];
    f[2] += cft[2];
    float r[3] = {origin[3 * c] - origin[3 * k], origin[3 * c + 1] - origin[3 * k + 1],
                  origin[3 * c + 2] - origin[3 * k + 2]};
    float cr[3];
    cross(r, cft, cr);
    tq[0] += cr[0] + cft[3];
    tq[1] += cr[1] + cft[4];
    tq[2] += cr[2] + cft[5];
  }
  if (k == 0) { /* rigid_body::set_derivative, tree.h:160-164 */
    change[0] = f[0];
    change[1] = f[1];
    change[2] = f[2];
    change[3] = tq[0];
    change[4] = tq[1];
    change[5] = tq[2];
  } else { /* axis_frame::set_derivative: torque . axis, tree.h:188-190 */
    change[6 + (k - 1)] = tq[0] * axis[3 * k] + tq[1] * axis[3 * k + 1] + tq[2] * axis[3 * k + 2];
  }
  ft[0] = f[0];
  ft[1] = f[1];
  ft[2] = f[2];
  ft[3] = tq[0];
  ft[4] = tq[1];
  ft[5] = tq[2];
}

/* ligands.derivative + flex.derivative (model.cu:219-221): the ligand's tree from node 0, every residue's from its
 * first_segment (its torque . axis is the torsion derivative; nothing propagates above it, tree.h:374-393) */
static void all_derivatives(const ora_ligand *L, const float *coords, const float *forces, const float *origin,
                            const float *axis, float *change) {
  float ft[6];
  node_derivative(L, 0, coords, forces, origin, axis, change, ft);
  for (int k = 1; k < L->n_nodes; k++)
    if (L->parent[k] == -2) node_derivative(L, k, coords, forces, origin, axis, change, ft);
}

/* one pass over the interacting pairs: model::other_pairs (kind 1, cap v[2]) are summed first, the ligand's own
 * (kind 0, cap v[0]) second, each on its own accumulator (model.cu:209-216: ie += other; ie += ligand) */
#define PAIR_CAP(L, p, v) (((L)->pair_kind && (L)->pair_kind[p]) ? (v)[2] : (v)[0])

/* model::eval_deriv (model.cu:202-225) with ig = cache (cache.cpp:65-83):
 *   set(c); e = sum over movable heavy atoms of grid::evaluate on its type's grid (v[1]);
 *   e += eval_interacting_pairs_deriv(ligand pairs, v[0]) (model.cu:38-60);
 *   ligands.derivative -> change.
 * grids[t] = data pointer for smina type t (NULL when absent -> atom skipped like t >= nat).
 * change may be NULL (energy only still uses the deriv-aware code path of the reference's
 * eval_deriv; for model::eval see ora_vina_eval).  coords_out/forces_out optional. */
float ora_vina_model_eval_deriv(const ora_vina_tables *T, const ora_grid_dims *gd, const float *const *grids,
                                float slope, const ora_ligand *L, const float *conf, const float *v, float *change,
                                float *coords_out, float *forces_out) {
  const int n = L->n_atoms;
  float *coords = (float *)malloc(sizeof(float) * 3 * n);
  float *forces = (float *)calloc(3 * (size_t)n, sizeof(float));
  float *origin = (float *)malloc(sizeof(float) * 3 * L->n_nodes);
  float *axis = (float *)malloc(sizeof(float) * 3 * L->n_nodes);
  ora_vina_set_conf(L, conf, coords, origin, axis);
  float e = 0;
  for (int i = 0; i < n_movable_of(L); i++) {
    int t = L->smt[i];
    if (is_hydrogen(t) || !grids[t]) continue; /* minus_forces[i] = 0 */
    float d[3];
    e += ora_vina_grid_evaluate(gd, grids[t], coords + 3 * i, slope, v[1], d);
    forces[3 * i] = d[0];
    forces[3 * i + 1] = d[1];
    forces[3 * i + 2] = d[2];
  }
  float ie = 0;
  for (int kind = 1; kind >= 0; kind--) {
    float sum = 0;
    for (int p = 0; p < L->n_pairs; p++) {
      if ((L->pair_kind ? L->pair_kind[p] : 0) != kind) continue;
      int a = L->pairs[2 * p], b = L->pairs[2 * p + 1];
      float r[3] = {coords[3 * b] - coords[3 * a], coords[3 * b + 1] - coords[3 * a + 1],
                    coords[3 * b + 2] - coords[3 * a + 2]};
      float r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      if (r2 < T->cutoff_sqr) {
        float pe, dor;
        ora_vina_table_eval_deriv(T, L->smt[a], L->smt[b], r2, &pe, &dor);
        float force[3] = {dor * r[0], dor * r[1], dor * r[2]};
        curl3(&pe, force, PAIR_CAP(L, p, v));
        sum += pe;
        for (int k = 0; k < 3; k++) {
          forces[3 * a + k] -= force[k];
          forces[3 * b + k] += force[k];
        }
      }
    }
    ie += sum;
  }
  e += ie;
  if (change) all_derivatives(L, coords, forces, origin, axis, change);
  if (coords_out) memcpy(coords_out, coords, sizeof(float) * 3 * n);
  if (forces_out) memcpy(forces_out, forces, sizeof(float) * 3 * n);
  free(coords);
  free(forces);
  free(origin);
  free(axis);
  return e;
}

/* ---------------------------------------------------------------------------------------------
 * conf increment (conf.h:54-59,103-118; quaternion.cu:32-62,96-100)
 * ------------------------------------------------------------------------------------------- */
static float normalized_angle(float x) {
  g_normalize_angle(&x);
  return x;
}

void ora_vina_conf_increment(float *conf, const float *p, float alpha, int n_tors) {
  conf[0] += alpha * p[0];
  conf[1] += alpha * p[1];
  conf[2] += alpha * p[2];
  float rot[3] = {alpha * p[3], alpha * p[4], alpha * p[5]};
  float angle = sqrtf(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2]);
  float rq[4] = {1, 0, 0, 0};
  if (angle > V_EPS) {
    float inv = 1 / angle;
    float ax[3] = {inv * rot[0], inv * rot[1], inv * rot[2]};
    angle_to_quat(ax, angle, rq);
  }
  float nq[4];
  quat_mul(rq, conf + 3, nq);
  quat_normalize_approx(nq);
  memcpy(conf + 3, nq, sizeof(nq));
  for (int i = 0; i < n_tors; i++) {
    conf[7 + i] += normalized_angle(alpha * p[6 + i]);
    g_normalize_angle(&conf[7 + i]);
  }
}

/* ---------------------------------------------------------------------------------------------
 * BFGS with Vina's fast line search (bfgs.h:34-91,357-502; quasi_newton.cpp:49-83)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const ora_vina_tables *T;
  const ora_grid_dims *gd;
  const float *const *grids;
  float slope;
  const ora_ligand *L;
  const float *v;
  long evals;
  /* direct (non_cache) evaluation instead of the cache grids */
  int direct;
  const float *rec_xyz;
  const int32_t *rec_smt;
  int n_rec;
  /* any other igrid (non_cache_cnn: the CNN loss + box penalties): the objective comes from a callback */
  float (*cb)(const float *conf, float *change, void *user);
  void *user;
  /* the conformation of the most recent evaluation = what `model` holds afterwards (model::eval_deriv starts with
   * set(c)); optional, [7+T] */
  float *model_conf;
} bfgs_ctx;

float ora_vina_noncache_eval(const ora_vina_tables *T, const float *w, int exact, const ora_grid_dims *gd, float slope,
                             const float *rec_xyz, const int32_t *rec_smt, int n_rec, const ora_ligand *L,
                             const float *conf, const float *v, int deriv, float *change, float *inter_out,
                             float *intra_out);

static float fx(bfgs_ctx *c, const float *conf, float *g) {
  c->evals++;
  if (c->model_conf) memcpy(c->model_conf, conf, sizeof(float) * (size_t)(7 + c->L->n_nodes - 1));
  if (c->cb) return c->cb(conf, g, c->user);
  if (c->direct)
    return ora_vina_noncache_eval(c->T, NULL, 0, c->gd, c->slope, c->rec_xyz, c->rec_smt, c->n_rec, c->L, conf, c->v, 1,
                                  g, NULL, NULL);
  return ora_vina_model_eval_deriv(c->T, c->gd, c->grids, c->slope, c->L, conf, c->v, g, NULL, NULL);
}

static float bfgs_run(bfgs_ctx *ctxp, float *conf, int max_iters, float *g_out);

static inline int hidx(int i, int j) { return i <= j ? i + j * (j + 1) / 2 : j + i * (i + 1) / 2; }

static float dotn(const float *a, const float *b, int n) {
  float t = 0;
  for (int i = 0; i < n; i++) t += a[i] * b[i];
  return t;
}

/* returns the final energy; conf is updated in place; change g receives the final gradient */
float ora_vina_bfgs(const ora_vina_tables *T, const ora_grid_dims *gd, const float *const *grids, float slope,
                    const ora_ligand *L, float *conf, const float *v, int max_iters, float *g_out, long *evals_out) {
  bfgs_ctx ctx = {T, gd, grids, slope, L, v, 0, 0, NULL, NULL, 0, NULL, NULL, NULL};
  float f = bfgs_run(&ctx, conf, max_iters, g_out);
  if (evals_out) *evals_out = ctx.evals;
  return f;
}

/* the same, also reporting the conformation `model` is left in (the last evaluated one; differs from the returned
 * conformation when bfgs reverts to its starting point, bfgs.h:490-494) */
static float bfgs_model(const ora_vina_tables *T, const ora_grid_dims *gd, const float *const *grids, float slope,
                        const ora_ligand *L, float *conf, const float *v, int max_iters, float *model_conf, long *evals) {
  bfgs_ctx ctx = {T, gd, grids, slope, L, v, 0, 0, NULL, NULL, 0, NULL, NULL, model_conf};
  float f = bfgs_run(&ctx, conf, max_iters, NULL);
  *evals += ctx.evals;
  return f;
}

/* quasi_newton on a caller-supplied objective (conf -> energy, change): the same bfgs<> + fast_line_search,
 * used by the tests to restate refinement on non_cache_cnn (non_cache_cnn.cpp:79-169), whose energy is the
 * CNN loss and lives in Python (oracle/cnn_ref.py). */
typedef float (*ora_fx_cb)(const float *conf, float *change, void *user);
float ora_vina_bfgs_cb(const ora_ligand *L, float *conf, int max_iters, ora_fx_cb cb, void *user, float *g_out,
                       long *evals_out) {
  bfgs_ctx ctx = {NULL, NULL, NULL, 0, L, NULL, 0, 0, NULL, NULL, 0, cb, user, NULL};
  float f = bfgs_run(&ctx, conf, max_iters, g_out);
  if (evals_out) *evals_out = ctx.evals;
  return f;
}

/* ligands.derivative(coords, minus_forces, g.ligands) (model.cu:223; tree.h:293-401) on given forces:
 * sets the conformation, folds per-atom forces into change[6 + T]; coords_out optional */
void ora_vina_forces_to_change(const ora_ligand *L, const float *conf, const float *forces, float *change,
                               float *coords_out) {
  const int n = L->n_atoms;
  float *coords = (float *)malloc(sizeof(float) * 3 * n);
  float *origin = (float *)malloc(sizeof(float) * 3 * L->n_nodes);
  float *axis = (float *)malloc(sizeof(float) * 3 * L->n_nodes);
  ora_vina_set_conf(L, conf, coords, origin, axis);
  if (forces && change) all_derivatives(L, coords, forces, origin, axis, change);
  if (coords_out) memcpy(coords_out, coords, sizeof(float) * 3 * n);
  free(coords);
  free(origin);
  free(axis);
}

/* refine_structure (main.cpp:131-171): BFGS on non_cache with the out-of-box slope raised 10x per try
 * until every heavy atom is inside the box (at most 5 tries); energy = max_fl if it never gets in. */
float ora_vina_refine(const ora_vina_tables *T, const ora_grid_dims *gd, const float *rec_xyz, const int32_t *rec_smt,
                      int n_rec, const ora_ligand *L, float *conf, const float *v, int max_iters, int *tries_out) {
  int ora_vina_within(const ora_grid_dims *gd, const ora_ligand *L, const float *conf);
  float slope = 10, e = 0;
  int p = 0;
  for (; p < 5; p++) {
    bfgs_ctx ctx = {T, gd, NULL, slope, L, v, 0, 1, rec_xyz, rec_smt, n_rec, NULL, NULL, NULL};
    e = bfgs_run(&ctx, conf, max_iters, NULL);
    if (ora_vina_within(gd, L, conf)) break;
    slope *= 10;
  }
  if (tries_out) *tries_out = p < 5 ? p + 1 : 5;
  if (!ora_vina_within(gd, L, conf)) e = V_MAXFL;
  return e;
}

/* instrumentation for the design notes: how many trials the line searches take (index = trials used, 0..10) */
static __thread long g_trial_hist[11]; /* per thread: bench.py runs chains on many threads */
static __thread long g_mc_stats[4]; /* steps, accepted, accepted with a second BFGS, rotation mutations */
void ora_vina_mc_stats(long *out, int reset) {
  for (int i = 0; i < 4; i++) {
    out[i] = g_mc_stats[i];
    if (reset) g_mc_stats[i] = 0;
  }
}
void ora_vina_trial_hist(long *out, int reset) {
  for (int i = 0; i < 11; i++) {
    out[i] = g_trial_hist[i];
    if (reset) g_trial_hist[i] = 0;
  }
}

/* --accurate_line_search (minimization_params::BFGSAccurateLineSearch): process-wide switch of this test library */
static int g_accurate_ls, g_simple; /* kind 2: minimization_params::Simple (simple_gradient_ascent, bfgs.h:234-355) */
void ora_vina_set_line_search(int kind) {
  g_accurate_ls = kind >= 1;
  g_simple = kind == 2;
}

/* quaternion_to_angle (quaternion.cu:46-62) */
static void quat_to_angle(const float *q, float *ang) {
  ang[0] = ang[1] = ang[2] = 0;
  const float c = q[0];
  if (c > -1 && c < 1) {
    float angle = 2 * acosf(c);
    if (angle > V_PI) angle -= 2 * V_PI;
    float sn = sinf(angle / 2);
    if (fabsf(sn) < V_EPS) return;
    for (int k = 0; k < 3; k++) ang[k] = q[1 + k] * (angle / sn);
  }
}

static float bfgs_run(bfgs_ctx *ctxp, float *conf, int max_iters, float *g_out) {
  const ora_ligand *L = ctxp->L;
  const int nt = L->n_nodes - 1, n = 6 + nt, nc = 7 + nt;
#define ctx (*ctxp)
  float *h = (float *)calloc((size_t)n * (n + 1) / 2, sizeof(float));
  for (int i = 0; i < n; i++) h[hidx(i, i)] = 1;
  float *g = (float *)malloc(sizeof(float) * n), *g_new = (float *)malloc(sizeof(float) * n);
  float *p = (float *)malloc(sizeof(float) * n), *y = (float *)malloc(sizeof(float) * n);
  float *mhy = (float *)malloc(sizeof(float) * n), *g_orig = (float *)malloc(sizeof(float) * n);
  float *x_new = (float *)malloc(sizeof(float) * nc), *x_orig = (float *)malloc(sizeof(float) * nc);
  float f0 = fx(&ctx, conf, g);
  const float f_orig = f0;
  memcpy(g_orig, g, sizeof(float) * n);
  memcpy(x_orig, conf, sizeof(float) * nc);
  memcpy(g_new, g, sizeof(float) * n);
  for (int step = 0; step < max_iters; step++) {
    if (g_simple) {
      for (int i = 0; i < n; i++) p[i] = -g[i]; /* set_to_neg, bfgs.h:262 */
    } else
    for (int i = 0; i < n; i++) { /* minus_mat_vec_product, bfgs.h:34-43 */
      float sum = 0;
      for (int j = 0; j < n; j++) sum += h[hidx(i, j)] * g[j];
      p[i] = -sum;
    }
    float f1 = 0, alpha = 1;
    const float pg = dotn(p, g, n);
    if (g_accurate_ls) { /* accurate_line_search, bfgs.h:104-180 (fl arithmetic, the literals 2.0 / 3.0 are double) */
      const float slope = pg;
      alpha = 0;
      if (!(slope >= 0)) {
        float ang[3], test = 0;
        quat_to_angle(conf + 3, ang);
        for (int i = 0; i < n; i++) { /* compute_lambdamin, bfgs.h:93-102; conf(i) in change indexing, conf.h:459-490 */
          float xi = i < 3 ? conf[i] : i < 6 ? ang[i - 3] : conf[i + 1];
          float t = fabsf(p[i]) / fmaxf(fabsf(xi), 1.0f);
          if (t > test) test = t;
        }
        const float alamin = V_EPS / test;
        float a_ = 1.0f, alpha2 = 0, f2 = 0;
        for (;;) {
          memcpy(x_new, conf, sizeof(float) * nc);
          ora_vina_conf_increment(x_new, p, a_, nt);
          f1 = fx(&ctx, x_new, g_new);
          if (a_ < alamin || !isfinite(a_)) break; /* alpha stays 0 */
          if (f1 <= f0 + 1.0e-4f * a_ * slope) {
            alpha = a_;
            break;
          }
          float tmplam;
          if (a_ == 1.0f) {
            tmplam = (float)(-(double)slope / (2.0 * (double)(f1 - f0 - slope)));
          } else {
            float rhs1 = f1 - f0 - a_ * slope, rhs2 = f2 - f0 - alpha2 * slope;
            float ca = (rhs1 / (a_ * a_) - rhs2 / (alpha2 * alpha2)) / (a_ - alpha2);
            float cb = (-alpha2 * rhs1 / (a_ * a_) + a_ * rhs2 / (alpha2 * alpha2)) / (a_ - alpha2);
            if (ca == 0.0f) {
              tmplam = (float)(-(double)slope / (2.0 * (double)cb));
            } else {
              float disc = (float)((double)(cb * cb) - 3.0 * (double)ca * (double)slope);
              if (disc < 0) tmplam = 0.5f * a_;
              else if (cb <= 0) tmplam = (float)((double)(-cb + sqrtf(disc)) / (3.0 * (double)ca));
              else tmplam = -slope / (cb + sqrtf(disc));
            }
            if (tmplam > 0.5f * a_) tmplam = 0.5f * a_;
          }
          alpha2 = a_;
          f2 = f1;
          a_ = fmaxf(tmplam, 0.1f * a_);
        }
      }
    } else
    /* fast_line_search, bfgs.h:73-91 */
    for (unsigned trial = 0; trial < 10; trial++) {
      memcpy(x_new, conf, sizeof(float) * nc);
      ora_vina_conf_increment(x_new, p, alpha, nt);
      f1 = fx(&ctx, x_new, g_new);
      if (f1 - f0 < 0.0001f * alpha * pg) {
        g_trial_hist[trial + 1]++;
        break;
      }
      alpha *= 0.5f;
    }
    if (alpha < 0.001f) g_trial_hist[0]++; /* never accepted: 10 trials */
    if (alpha == 0) break;
    for (int i = 0; i < n; i++) y[i] = g_new[i] - g[i];
    f0 = f1;
    memcpy(conf, x_new, sizeof(float) * nc);
    memcpy(g, g_new, sizeof(float) * n);
    float gradnormsq = dotn(g, g, n);
    if (!(gradnormsq >= 1e-4f)) break;
    if (g_simple) continue; /* no Hessian estimate */
    if (step == 0) {
      const float yy = dotn(y, y, n);
      if (fabsf(yy) > V_EPS) {
        float dgl = alpha * dotn(y, p, n) / yy;
        for (int i = 0; i < n; i++) h[hidx(i, i)] = dgl;
      }
    }
    { /* bfgs_update, bfgs.h:52-66 */
      const float yp = dotn(y, p, n);
      if (!(alpha * yp < V_EPS)) {
        for (int i = 0; i < n; i++) {
          float sum = 0;
          for (int j = 0; j < n; j++) sum += h[hidx(i, j)] * y[j];
          mhy[i] = -sum;
        }
        const float yhy = -dotn(y, mhy, n);
        const float r = 1 / (alpha * yp);
        for (int i = 0; i < n; i++)
          for (int j = i; j < n; j++)
            h[hidx(i, j)] += alpha * r * (mhy[i] * p[j] + mhy[j] * p[i]) + alpha * alpha * (r * r * yhy + r) * p[i] * p[j];
      }
    }
  }
  if (!(f0 <= f_orig)) {
    f0 = f_orig;
    memcpy(conf, x_orig, sizeof(float) * nc);
    memcpy(g, g_orig, sizeof(float) * n);
  }
  if (g_out) memcpy(g_out, g, sizeof(float) * n);
#undef ctx
  free(h);
  free(g);
  free(g_new);
  free(p);
  free(y);
  free(mhy);
  free(g_orig);
  free(x_new);
  free(x_orig);
  return f0;
}

/* model::eval's own user-grid term (model.cu:125-134): every atom of the ligand, hydrogens included, at slope 1000,
 * added one by one after the pair terms -- this is how the user grid reaches the final energies (eval_adjusted). */
static float model_eval_user_term(const ora_ligand *L, const float *coords, float e) {
  if (!g_ug.on) return e;
  const int b = L->lig_end > L->lig_begin ? L->lig_begin : 0, en = L->lig_end > L->lig_begin ? L->lig_end : L->n_atoms;
  for (int i = b; i < en; i++) e += user_grid_eval(coords + 3 * i, 1000.0f, NULL);
  return e;
}

/* model::eval with ig = cache (cache.cpp:52-63 + model.cu:22-36): energy only (score_only / final energies). */
float ora_vina_eval(const ora_vina_tables *T, const ora_grid_dims *gd, const float *const *grids, float slope,
                    const ora_ligand *L, const float *conf, const float *v) {
  const int n = L->n_atoms;
  float *coords = (float *)malloc(sizeof(float) * 3 * n);
  float *origin = (float *)malloc(sizeof(float) * 3 * L->n_nodes);
  float *axis = (float *)malloc(sizeof(float) * 3 * L->n_nodes);
  ora_vina_set_conf(L, conf, coords, origin, axis);
  float e = 0;
  for (int i = 0; i < n_movable_of(L); i++) {
    int t = L->smt[i];
    if (is_hydrogen(t) || !grids[t]) continue;
    e += ora_vina_grid_evaluate(gd, grids[t], coords + 3 * i, slope, v[1], NULL);
  }
  /* eval_interacting_pairs sums on its own; model::evale adds other_pairs, model::eval then the ligand's
   * (model.cu:22-36,112-123) */
  for (int kind = 1; kind >= 0; kind--) {
    float ie = 0;
    for (int p = 0; p < L->n_pairs; p++) {
      if ((L->pair_kind ? L->pair_kind[p] : 0) != kind) continue;
      int a = L->pairs[2 * p], b = L->pairs[2 * p + 1];
      float dx = coords[3 * a] - coords[3 * b], dy = coords[3 * a + 1] - coords[3 * b + 1],
            dz = coords[3 * a + 2] - coords[3 * b + 2];
      float r2 = dx * dx + dy * dy + dz * dz;
      if (r2 < T->cutoff_sqr) {
        /* p.eval = eval_fast(...) in the reference (precalculate.h:67-70): midpoint table */
        float pe = ora_vina_eval_fast(T, L->smt[a], L->smt[b], r2);
        curl1(&pe, PAIR_CAP(L, p, v));
        ie += pe;
      }
    }
    e += ie;
  }
  e = model_eval_user_term(L, coords, e);
  free(coords);
  free(origin);
  free(axis);
  return e;
}

/* cache::eval (cache.cpp:52-63): receptor-grid term only, energy only -- what update_energy
 * (monte_carlo.cpp:44-47) feeds the Metropolis criterion. */
float ora_vina_cache_eval(const ora_grid_dims *gd, const float *const *grids, float slope, const ora_ligand *L,
                          const float *conf, float v1) {
  const int n = L->n_atoms;
  float *coords = (float *)malloc(sizeof(float) * 3 * n);
  float *origin = (float *)malloc(sizeof(float) * 3 * L->n_nodes);
  float *axis = (float *)malloc(sizeof(float) * 3 * L->n_nodes);
  ora_vina_set_conf(L, conf, coords, origin, axis);
  float e = 0;
  for (int i = 0; i < n_movable_of(L); i++) {
    int t = L->smt[i];
    if (is_hydrogen(t) || !grids[t]) continue;
    e += ora_vina_grid_evaluate(gd, grids[t], coords + 3 * i, slope, v1, NULL);
  }
  free(coords);
  free(origin);
  free(axis);
  return e;
}

/* ---------------------------------------------------------------------------------------------
 * Monte-Carlo chain: monte_carlo::operator() (monte_carlo.cpp:99-148), mutate_conf (mutate.cpp:35-73),
 * metropolis_accept (monte_carlo.cpp:38-42), conf randomize (conf.h:119-122,189-192),
 * add_to_output_container (coords.cpp:25-56).
 *
 * The chain draws from the reference's stream (rng kind 1 below: mt19937 + Boost's distributions restated, verified
 * draw by draw against oracle/_ref) and keeps the reference's `model` state: gyration_radius and the Metropolis energy
 * are taken at the last conformation bfgs evaluated (model_conf), as monte_carlo.cpp does.  tests/test_ref_vina.py
 * compares whole chains with the reference bit for bit.
 * ------------------------------------------------------------------------------------------- */
/* Two generators behind one interface.
 * kind 0: the counter-based splitmix64 stream shared with the HIP kernel (statistical parity with the reference).
 * kind 1: boost::mt19937 (= the standard MT19937) under restatements of Boost.Random's distributions, exactly
 *         what oracle/ref_shims/boost/random.hpp gives the reference code in oracle/_ref: uniform_real =
 *         u32 / 2^32 * (b - a) + a, redrawn when it reaches b; uniform_int = equal buckets with rejection;
 *         normal_distribution = Box-Muller on two fresh uniforms (random_normal builds a new distribution object per
 *         call, random.cpp:38-42, so the cached second value is never used).  With kind 1 this chain follows the
 *         reference's monte_carlo.cpp step for step (tests/test_ref_vina.py). */
typedef struct {
  int kind;
  uint64_t s;
  uint32_t mt[624];
  int mti;
} ora_rng;

static void rng_seed(ora_rng *r, int kind, uint64_t seed) {
  r->kind = kind;
  r->s = seed;
  r->mt[0] = (uint32_t)seed;
  for (int i = 1; i < 624; i++) r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
  r->mti = 624;
}
static uint32_t rng_u32(ora_rng *r) {
  if (r->kind == 0) { /* splitmix64, high half */
    r->s += 0x9E3779B97F4A7C15ull;
    uint64_t z = r->s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
  }
  if (r->mti >= 624) {
    for (int i = 0; i < 624; i++) {
      uint32_t y = (r->mt[i] & 0x80000000u) | (r->mt[(i + 1) % 624] & 0x7fffffffu);
      r->mt[i] = r->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    r->mti = 0;
  }
  uint32_t y = r->mt[r->mti++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
static float rng_u01(ora_rng *r) { return (float)(rng_u32(r) >> 8) * (1.0f / 16777216.0f); }
static float rng_fl(ora_rng *r, float a, float b) { /* random_fl, random.cpp:27-35 */
  if (r->kind == 0) return a + (b - a) * rng_u01(r);
  for (;;) {
    float numerator = (float)rng_u32(r), divisor = 4294967296.0f;
    float result = numerator / divisor * (b - a) + a;
    if (result < b) return result;
  }
}
static int rng_int(ora_rng *r, int a, int b) { /* random_int, random.cpp:44-52 */
  if (r->kind == 0) return a + (int)(rng_u32(r) % (uint32_t)(b - a + 1));
  const uint32_t range = (uint32_t)b - (uint32_t)a, brange = 0xffffffffu;
  if (range == 0) return a;
  uint32_t bucket = brange / (range + 1);
  if (brange % (range + 1) == range) ++bucket;
  for (;;) {
    uint32_t q = rng_u32(r) / bucket;
    if (q <= range) return (int)(q + (uint32_t)a);
  }
}
static float rng_normal(ora_rng *r) { /* random_normal(0, 1), random.cpp:37-42 */
  if (r->kind == 0) { /* Box-Muller */
    float u1 = ((float)(rng_u32(r) >> 8) + 1.0f) * (1.0f / 16777216.0f);
    float u2 = rng_u01(r);
    return sqrtf(-2.0f * logf(u1)) * cosf(2.0f * V_PI * u2);
  }
  float r1 = rng_fl(r, 0, 1), r2 = rng_fl(r, 0, 1);
  float rho = sqrtf(-2.0f * logf(1.0f - r2));
  return rho * cosf(2.0f * 3.14159265358979323846f * r1) * 1.0f + 0.0f;
}
static void rng_inside_sphere(ora_rng *r, float *o) { /* random.cpp:66-75 */
  for (;;) {
    o[0] = rng_fl(r, -1, 1);
    o[1] = rng_fl(r, -1, 1);
    o[2] = rng_fl(r, -1, 1);
    if (o[0] * o[0] + o[1] * o[1] + o[2] * o[2] < 1) return;
  }
}
/* raw draws for the tests that check the restated distributions against oracle/_ref's */
void ora_vina_random_stream(int kind, uint64_t seed, int n, float *uniform01, int32_t *ints_0_9, float *normals) {
  ora_rng r;
  rng_seed(&r, kind, seed);
  for (int i = 0; i < n; i++) uniform01[i] = rng_fl(&r, 0, 1);
  for (int i = 0; i < n; i++) ints_0_9[i] = rng_int(&r, 0, 9);
  for (int i = 0; i < n; i++) normals[i] = rng_normal(&r);
}

typedef struct {
  int n_steps, max_iters, num_saved;
  float temperature, mutation_amplitude, min_rmsd;
  float hunt_cap[3], authentic_v[3];
} ora_mc_params;

/* mutate_conf (mutate.cpp:35-73) of cand; `model_conf` = the conformation `model` currently holds, which is what
 * model::gyration_radius reads (coords and the root node's origin, model.cpp:1002-1014). */
static void mc_mutate(const ora_ligand *L, ora_rng *rng, float amplitude, float *cand, const float *model_conf,
                      float *coords, float *origin, float *axis) {
  const int nt = L->n_nodes - 1, na = L->n_atoms;
  int which = rng_int(rng, 0, 2 + nt - 1);
  if (which == 0) {
    float d[3];
    rng_inside_sphere(rng, d);
    for (int k = 0; k < 3; k++) cand[k] += amplitude * d[k];
  } else if (which == 1) {
    ora_vina_set_conf(L, model_conf, coords, origin, axis);
    float acc = 0;
    int cnt = 0;
    const int lb = L->lig_end > L->lig_begin ? L->lig_begin : 0, le = L->lig_end > L->lig_begin ? L->lig_end : na;
    for (int i = lb; i < le; i++) /* model::gyration_radius: the ligand's atoms (model.cpp:1002-1014) */
      if (!is_hydrogen(L->smt[i])) {
        float dx = coords[3 * i] - origin[0], dy = coords[3 * i + 1] - origin[1], dz = coords[3 * i + 2] - origin[2];
        acc += dx * dx + dy * dy + dz * dz;
        cnt++;
      }
    float gr = cnt > 0 ? sqrtf(acc / (float)cnt) : 0;
    if (gr > V_EPS) {
      float d[3];
      rng_inside_sphere(rng, d);
      float s = amplitude / gr;
      float rot[6] = {0, 0, 0, s * d[0], s * d[1], s * d[2]};
      ora_vina_conf_increment(cand, rot, 1.0f, 0); /* quaternion_increment(orientation, rotation) */
    }
  } else {
    cand[7 + (which - 2)] = rng_fl(rng, -V_PI, V_PI);
  }
}
void ora_vina_mutate(const ora_ligand *L, int rng_kind, uint64_t seed, float amplitude, float *conf) {
  ora_rng rng;
  rng_seed(&rng, rng_kind, seed);
  float *coords = (float *)malloc(sizeof(float) * 3 * L->n_atoms), *origin = (float *)malloc(sizeof(float) * 3 * L->n_nodes),
        *axis = (float *)malloc(sizeof(float) * 3 * L->n_nodes), *mc = (float *)malloc(sizeof(float) * (7 + L->n_nodes - 1));
  memcpy(mc, conf, sizeof(float) * (7 + L->n_nodes - 1));
  mc_mutate(L, &rng, amplitude, conf, mc, coords, origin, axis);
  free(coords);
  free(origin);
  free(axis);
  free(mc);
}

/* out arrays sized for num_saved entries: out_e[num_saved], out_conf[num_saved][7+T],
 * out_coords[num_saved][n_heavy][3]; returns the number of saved poses (sorted by energy).
 *
 * `model` state: the reference evaluates the Metropolis energy (update_energy, monte_carlo.cpp:44-47) and the
 * gyration radius on whatever coordinates the last evaluation left in `model`.  That is the returned conformation
 * except when bfgs ends by reverting to its starting point (bfgs.h:490-494): then `model` still holds the last
 * line-search trial.  model_conf tracks it; m.set(tmp.c) after an accepted step resets it (monte_carlo.cpp:122,131). */
int ora_vina_mc_chain_rng(const ora_vina_tables *T, const ora_grid_dims *gd, const float *const *grids, float slope,
                          const ora_ligand *L, const float *corner1, const float *corner2, uint64_t seed, int rng_kind,
                          const float *conf0, const ora_mc_params *P, float *out_e, float *out_conf, float *out_coords, long *evals_out) {
  const int nt = L->n_nodes - 1, nc = 7 + nt, na = n_movable_of(L); /* get_heavy_atom_movable_coords: movable atoms */
  int nh = 0;
  for (int i = 0; i < na; i++) nh += !is_hydrogen(L->smt[i]);
  ora_rng rng;
  rng_seed(&rng, rng_kind, seed);
  float *tmp = (float *)malloc(sizeof(float) * nc), *cand = (float *)malloc(sizeof(float) * nc),
        *model_conf = (float *)malloc(sizeof(float) * nc);
  float *coords = (float *)malloc(sizeof(float) * 3 * L->n_atoms), *origin = (float *)malloc(sizeof(float) * 3 * L->n_nodes),
        *axis = (float *)malloc(sizeof(float) * 3 * L->n_nodes), *hc = (float *)malloc(sizeof(float) * 3 * nh);
  long evals = 0;
  /* conf::randomize */
  for (int k = 0; k < 3; k++) tmp[k] = rng_fl(&rng, corner1[k], corner2[k]);
  for (;;) {
    float q[4];
    for (int k = 0; k < 4; k++) q[k] = rng_normal(&rng); /* a0..a3 in order (quaternion.cu:81-86) */
    /* abs(qt) (quaternion.h:169-190): scaled by the largest component */
    float maxim = 0, nrm = 0;
    for (int k = 0; k < 4; k++)
      if (fabsf(q[k]) > maxim) maxim = fabsf(q[k]);
    if (maxim != 0) {
      float mixam = (float)(1.0 / maxim), sum = 0;
      for (int k = 0; k < 4; k++) {
        float val = q[k] * mixam;
        sum += val * val;
      }
      nrm = maxim * sqrtf(sum);
    }
    if (nrm > V_EPS) {
      for (int k = 0; k < 4; k++) tmp[3 + k] = q[k] / nrm;
      break;
    }
  }
  for (int k = 0; k < nt; k++) tmp[7 + k] = rng_fl(&rng, -V_PI, V_PI);
  /* before the first evaluation `model` holds the input pose; the first mutation of the orientation reads it */
  if (conf0) {
    memcpy(model_conf, conf0, sizeof(float) * nc);
  } else { /* any rigid placement of the input torsions gives the same radius */
    memset(model_conf, 0, sizeof(float) * nc);
    model_conf[3] = 1.0f;
  }
  float tmp_e = 0, best_e = V_MAXFL;
  int n_out = 0;
  for (int step = 0; step < P->n_steps; step++) {
    memcpy(cand, tmp, sizeof(float) * nc);
    mc_mutate(L, &rng, P->mutation_amplitude, cand, model_conf, coords, origin, axis);
    bfgs_model(T, gd, grids, slope, L, cand, P->hunt_cap, P->max_iters, model_conf, &evals);
    float cand_e = ora_vina_cache_eval(gd, grids, slope, L, model_conf, P->authentic_v[1]);
    int accept = step == 0 || cand_e < tmp_e;
    if (!accept) { /* metropolis_accept */
      float prob = expf((tmp_e - cand_e) / P->temperature);
      accept = rng_fl(&rng, 0, 1) < prob;
    }
    g_mc_stats[0]++;
    if (accept) {
      g_mc_stats[1]++;
      memcpy(tmp, cand, sizeof(float) * nc);
      tmp_e = cand_e;
      memcpy(model_conf, tmp, sizeof(float) * nc); /* m.set(tmp.c) */
      if (tmp_e < best_e || n_out < P->num_saved) {
        g_mc_stats[2]++;
        bfgs_model(T, gd, grids, slope, L, tmp, P->authentic_v, P->max_iters, model_conf, &evals);
        tmp_e = ora_vina_cache_eval(gd, grids, slope, L, model_conf, P->authentic_v[1]);
        memcpy(model_conf, tmp, sizeof(float) * nc); /* m.set(tmp.c) */
        ora_vina_set_conf(L, tmp, coords, origin, axis);
        int h = 0;
        for (int i = 0; i < na; i++)
          if (!is_hydrogen(L->smt[i])) {
            hc[3 * h] = coords[3 * i];
            hc[3 * h + 1] = coords[3 * i + 1];
            hc[3 * h + 2] = coords[3 * i + 2];
            h++;
          }
        /* add_to_output_container (coords.cpp:43-56) */
        int closest = n_out;
        float closest_rmsd = V_MAXFL;
        for (int o = 0; o < n_out; o++) {
          float acc = 0;
          for (int i = 0; i < nh; i++) { /* rmsd_upper_bound: sum of vec_distance_sqr per atom (coords.cpp:25-32) */
            float dx = hc[3 * i] - out_coords[(size_t)o * 3 * nh + 3 * i], dy = hc[3 * i + 1] - out_coords[(size_t)o * 3 * nh + 3 * i + 1],
                  dz = hc[3 * i + 2] - out_coords[(size_t)o * 3 * nh + 3 * i + 2];
            acc += dx * dx + dy * dy + dz * dz;
          }
          float res = nh > 0 ? sqrtf(acc / (float)nh) : 0;
          if (o == 0 || res < closest_rmsd) {
            closest = o;
            closest_rmsd = res;
          }
        }
        int slot = -1;
        if (closest < n_out && closest_rmsd < P->min_rmsd) {
          if (tmp_e < out_e[closest]) slot = closest;
        } else if (n_out < P->num_saved) {
          slot = n_out++;
        } else if (n_out > 0 && tmp_e < out_e[n_out - 1]) {
          slot = n_out - 1;
        }
        if (slot >= 0) {
          out_e[slot] = tmp_e;
          memcpy(out_conf + (size_t)slot * nc, tmp, sizeof(float) * nc);
          memcpy(out_coords + (size_t)slot * 3 * nh, hc, sizeof(float) * 3 * nh);
          /* out.sort(): insertion keeps the container ordered by energy */
          for (int o = slot; o > 0 && out_e[o] < out_e[o - 1]; o--) {
            float te = out_e[o];
            out_e[o] = out_e[o - 1];
            out_e[o - 1] = te;
            for (int k = 0; k < nc; k++) {
              float t = out_conf[(size_t)o * nc + k];
              out_conf[(size_t)o * nc + k] = out_conf[(size_t)(o - 1) * nc + k];
              out_conf[(size_t)(o - 1) * nc + k] = t;
            }
            for (int k = 0; k < 3 * nh; k++) {
              float t = out_coords[(size_t)o * 3 * nh + k];
              out_coords[(size_t)o * 3 * nh + k] = out_coords[(size_t)(o - 1) * 3 * nh + k];
              out_coords[(size_t)(o - 1) * 3 * nh + k] = t;
            }
          }
          for (int o = slot; o + 1 < n_out && out_e[o + 1] < out_e[o]; o++) {
            float te = out_e[o];
            out_e[o] = out_e[o + 1];
            out_e[o + 1] = te;
            for (int k = 0; k < nc; k++) {
              float t = out_conf[(size_t)o * nc + k];
              out_conf[(size_t)o * nc + k] = out_conf[(size_t)(o + 1) * nc + k];
              out_conf[(size_t)(o + 1) * nc + k] = t;
            }
            for (int k = 0; k < 3 * nh; k++) {
              float t = out_coords[(size_t)o * 3 * nh + k];
              out_coords[(size_t)o * 3 * nh + k] = out_coords[(size_t)(o + 1) * 3 * nh + k];
              out_coords[(size_t)(o + 1) * 3 * nh + k] = t;
            }
          }
        }
        if (tmp_e < best_e) best_e = tmp_e;
      }
    }
  }
  if (evals_out) *evals_out = evals;
  free(tmp);
  free(cand);
  free(model_conf);
  free(coords);
  free(origin);
  free(axis);
  free(hc);
  return n_out;
}

int ora_vina_mc_chain(const ora_vina_tables *T, const ora_grid_dims *gd, const float *const *grids, float slope,
                      const ora_ligand *L, const float *corner1, const float *corner2, uint64_t seed,
                      const ora_mc_params *P, float *out_e, float *out_conf, float *out_coords, long *evals_out) {
  return ora_vina_mc_chain_rng(T, gd, grids, slope, L, corner1, corner2, seed, 0, NULL, P, out_e, out_conf, out_coords,
                               evals_out);
}

/* ---------------------------------------------------------------------------------------------
 * Direct (grid-free) receptor term and exact pair functions: rows "Direct pair evaluators" /
 * a18 of SURVEY 8a -- non_cache::eval (non_cache.cpp:52-83), non_cache::eval_deriv (:125-179),
 * check_bounds(_deriv) (:32-50,102-123), within (:84-101), precalculate_exact (precalculate.h:452-494),
 * refine_structure (main.cpp:131-171), conf-independent num_tors_div (everything.h:796-814).
 * prec_mode 0 = precalculate_linear tables, 1 = precalculate_exact.
 * ------------------------------------------------------------------------------------------- */
static float prec_eval(const ora_vina_tables *T, const float *w, int exact, int t1, int t2, float r2) {
  if (exact) return ora_vina_pair_energy(w, t1, t2, sqrtf(r2)); /* precalculate_exact::eval_fast */
  return ora_vina_eval_fast(T, t1, t2, r2);
}

static void prec_eval_deriv(const ora_vina_tables *T, const float *w, int exact, int t1, int t2, float r2, float *e,
                            float *dor) {
  if (!exact) {
    ora_vina_table_eval_deriv(T, t1, t2, r2, e, dor);
    return;
  }
  const float delta = 0.000005f; /* precalculate.h:457 */
  float r = sqrtf(r2);
  float X = ora_vina_pair_energy(w, t1, t2, r);
  float rhi = r + delta, rlo = r - delta;
  if (rlo < 0) rlo = 0;
  float W = ora_vina_pair_energy(w, t1, t2, rlo), Y = ora_vina_pair_energy(w, t1, t2, rhi);
  float dx = (Y - W) / (rhi - rlo);
  *e = X;
  *dor = dx / r;
}

/* model::eval_deriv / model::eval with ig = non_cache.  deriv != 0 -> eval_deriv semantics (interpolated
 * tables / numeric exact derivative, change written); deriv == 0 -> eval semantics (p.eval = eval_fast). */
float ora_vina_noncache_eval(const ora_vina_tables *T, const float *w, int exact, const ora_grid_dims *gd, float slope,
                             const float *rec_xyz, const int32_t *rec_smt, int n_rec, const ora_ligand *L,
                             const float *conf, const float *v, int deriv, float *change, float *inter_out,
                             float *intra_out) {
  const int n = L->n_atoms;
  float *coords = (float *)malloc(sizeof(float) * 3 * n), *forces = (float *)calloc(3 * (size_t)n, sizeof(float));
  float *origin = (float *)malloc(sizeof(float) * 3 * L->n_nodes), *axis = (float *)malloc(sizeof(float) * 3 * L->n_nodes);
  ora_vina_set_conf(L, conf, coords, origin, axis);
  float e = 0;
  for (int i = 0; i < n_movable_of(L); i++) {
    int t1 = L->smt[i];
    if (is_hydrogen(t1)) continue;
    float adj[3], oob_d[3] = {0, 0, 0}, oob = 0;
    for (int k = 0; k < 3; k++) { /* check_bounds_deriv */
      float c = coords[3 * i + k];
      adj[k] = c;
      if (gd->n[k] > 0) {
        if (c < gd->begin[k]) {
          adj[k] = gd->begin[k];
          oob_d[k] = -1;
          oob += fabsf(c - gd->begin[k]);
        } else if (c > gd->end[k]) {
          adj[k] = gd->end[k];
          oob_d[k] = 1;
          oob += fabsf(c - gd->end[k]);
        }
      }
    }
    oob *= slope;
    float this_e = 0, d[3] = {0, 0, 0};
    for (int j = 0; j < n_rec; j++) {
      if (is_hydrogen(rec_smt[j])) continue; /* szv_grid possibilities hold no hydrogens */
      float rx = adj[0] - rec_xyz[3 * j], ry = adj[1] - rec_xyz[3 * j + 1], rz = adj[2] - rec_xyz[3 * j + 2];
      float r2 = rx * rx + ry * ry + rz * rz;
      if (r2 < T->cutoff_sqr) {
        if (deriv) {
          float pe, dor;
          prec_eval_deriv(T, w, exact, t1, rec_smt[j], r2, &pe, &dor);
          this_e += pe;
          d[0] += dor * rx;
          d[1] += dor * ry;
          d[2] += dor * rz;
        } else {
          this_e += prec_eval(T, w, exact, t1, rec_smt[j], r2);
        }
      }
    }
    if (deriv) {
      if (g_ug.on) { /* non_cache.cpp:168-173 */
        float ud[3] = {0, 0, 0};
        this_e += user_grid_eval(coords + 3 * i, slope, ud);
        for (int k = 0; k < 3; k++) d[k] += ud[k];
      }
      curl3(&this_e, d, v[1]);
      for (int k = 0; k < 3; k++) forces[3 * i + k] = d[k] + slope * oob_d[k];
    } else {
      curl1(&this_e, v[1]);
    }
    e += this_e + oob;
  }
  float ie = 0;
  for (int kind = 1; kind >= 0; kind--) {
    float sum = 0;
    for (int p = 0; p < L->n_pairs; p++) {
      if ((L->pair_kind ? L->pair_kind[p] : 0) != kind) continue;
      int a = L->pairs[2 * p], b = L->pairs[2 * p + 1];
      float r[3] = {coords[3 * b] - coords[3 * a], coords[3 * b + 1] - coords[3 * a + 1],
                    coords[3 * b + 2] - coords[3 * a + 2]};
      float r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      if (r2 < T->cutoff_sqr) {
        if (deriv) {
          float pe, dor;
          prec_eval_deriv(T, w, exact, L->smt[a], L->smt[b], r2, &pe, &dor);
          float force[3] = {dor * r[0], dor * r[1], dor * r[2]};
          curl3(&pe, force, PAIR_CAP(L, p, v));
          sum += pe;
          for (int k = 0; k < 3; k++) {
            forces[3 * a + k] -= force[k];
            forces[3 * b + k] += force[k];
          }
        } else {
          float pe = prec_eval(T, w, exact, L->smt[a], L->smt[b], r2);
          curl1(&pe, PAIR_CAP(L, p, v));
          sum += pe;
        }
      }
    }
    ie += sum;
  }
  if (deriv && change) all_derivatives(L, coords, forces, origin, axis, change);
  if (inter_out) *inter_out = e;
  if (intra_out) *intra_out = ie;
  float total = e + ie;
  if (!deriv) total = model_eval_user_term(L, coords, total);
  free(coords);
  free(forces);
  free(origin);
  free(axis);
  return total;
}

/* non_cache::within (non_cache.cpp:84-101), margin 0.0001 */
int ora_vina_within(const ora_grid_dims *gd, const ora_ligand *L, const float *conf) {
  const int n = L->n_atoms;
  float *coords = (float *)malloc(sizeof(float) * 3 * n);
  float *origin = (float *)malloc(sizeof(float) * 3 * L->n_nodes), *axis = (float *)malloc(sizeof(float) * 3 * L->n_nodes);
  ora_vina_set_conf(L, conf, coords, origin, axis);
  int ok = 1;
  for (int i = 0; i < n_movable_of(L) && ok; i++) {
    if (is_hydrogen(L->smt[i])) continue;
    for (int k = 0; k < 3; k++)
      if (gd->n[k] > 0 && (coords[3 * i + k] < gd->begin[k] - 0.0001f || coords[3 * i + k] > gd->end[k] + 0.0001f)) ok = 0;
  }
  free(coords);
  free(origin);
  free(axis);
  return ok;
}

/* do_search's reported energies for the general model (flexible residues included), main.cpp:339-344:
 *   intramolecular = model::eval_intramolecular(exact_prec, v, c)            (model.cu:352-399)
 *                  = ligand pairs (v[0]) + flexible heavy atoms x rigid receptor heavy atoms, every PAIR curled with v[1]
 *                    + those other_pairs that touch no ligand atom (v[2])
 *   e = model::eval(exact_prec, non_cache, v, c) = non_cache::eval (its own LINEAR tables) + other_pairs + ligand pairs
 *       (+ the user-grid sum), model.cu:112-136
 * returns e (before conf_independent); *intra_out = intramolecular.  Sums in the reference's order. */
float ora_vina_model_energies(const ora_vina_tables *T, const float *w, const ora_grid_dims *gd, float slope,
                              const float *rec_xyz, const int32_t *rec_smt, int n_rec, const ora_ligand *L,
                              const float *conf, const float *v, float *intra_out) {
  const int n = L->n_atoms;
  float *coords = (float *)malloc(sizeof(float) * 3 * n);
  float *origin = (float *)malloc(sizeof(float) * 3 * L->n_nodes), *axis = (float *)malloc(sizeof(float) * 3 * L->n_nodes);
  ora_vina_set_conf(L, conf, coords, origin, axis);
  const int lb = L->lig_end > L->lig_begin ? L->lig_begin : 0, le = L->lig_end > L->lig_begin ? L->lig_end : n;
  float pair_sum[2] = {0, 0}, flexflex = 0; /* eval_interacting_pairs sums on its own (model.cu:22-36) */
  for (int kind = 0; kind < 2; kind++)
    for (int p = 0; p < L->n_pairs; p++) {
      if ((L->pair_kind ? L->pair_kind[p] : 0) != kind) continue;
      int a = L->pairs[2 * p], b = L->pairs[2 * p + 1];
      float dx = coords[3 * a] - coords[3 * b], dy = coords[3 * a + 1] - coords[3 * b + 1],
            dz = coords[3 * a + 2] - coords[3 * b + 2];
      float r2 = dx * dx + dy * dy + dz * dz;
      if (!(r2 < T->cutoff_sqr)) continue;
      float pe = prec_eval(T, w, 1, L->smt[a], L->smt[b], r2);
      curl1(&pe, PAIR_CAP(L, p, v));
      pair_sum[kind] += pe;
    }
  float intra = 0;
  intra += pair_sum[0];
  for (int i = 0; i < n_movable_of(L); i++) { /* flex-rigid */
    if (i >= lb && i < le) continue;
    int t1 = L->smt[i];
    if (is_hydrogen(t1)) continue;
    for (int j = 0; j < n_rec; j++) {
      if (is_hydrogen(rec_smt[j])) continue;
      float dx = coords[3 * i] - rec_xyz[3 * j], dy = coords[3 * i + 1] - rec_xyz[3 * j + 1],
            dz = coords[3 * i + 2] - rec_xyz[3 * j + 2];
      float r2 = dx * dx + dy * dy + dz * dz;
      if (r2 < T->cutoff_sqr) {
        float pe = prec_eval(T, w, 1, t1, rec_smt[j], r2);
        curl1(&pe, v[1]);
        intra += pe;
      }
    }
  }
  for (int p = 0; p < L->n_pairs; p++) { /* flex-flex: other_pairs with no ligand atom, added one by one */
    if (!(L->pair_kind && L->pair_kind[p])) continue;
    int a = L->pairs[2 * p], b = L->pairs[2 * p + 1];
    if ((a >= lb && a < le) || (b >= lb && b < le)) continue;
    float dx = coords[3 * a] - coords[3 * b], dy = coords[3 * a + 1] - coords[3 * b + 1],
          dz = coords[3 * a + 2] - coords[3 * b + 2];
    float r2 = dx * dx + dy * dy + dz * dz;
    if (r2 < T->cutoff_sqr) {
      float pe = prec_eval(T, w, 1, L->smt[a], L->smt[b], r2);
      curl1(&pe, v[2]);
      intra += pe;
    }
  }
  (void)flexflex;
  float inter = 0;
  (void)ora_vina_noncache_eval(T, w, 0, gd, slope, rec_xyz, rec_smt, n_rec, L, conf, v, 0, NULL, &inter, NULL);
  float e = inter;
  e += pair_sum[1];
  e += pair_sum[0];
  e = model_eval_user_term(L, coords, e);
  if (intra_out) *intra_out = intra;
  free(coords);
  free(origin);
  free(axis);
  return e;
}

/* num_tors_div (everything.h:796-814) with the default weight 5*0.05846/0.1 - 1 (main.cpp:1329):
 * e / (1 + w * num_tors / 5), w = 0.1 * (weight + 1) */
float ora_vina_conf_independent(float e, float num_tors) {
  const float weight = (float)(5 * 0.05846 / 0.1 - 1);
  float w = 0.1f * (weight + 1);
  float y = 1 + w * num_tors / 5.0f;
  if (fabsf(e) < V_EPS) return 0;
  if (fabsf(y) < V_EPS) return (e * y > 0) ? V_MAXFL : -V_MAXFL;
  return e / y;
}
