/*
 * oracle/voxel_ref.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar fp32, no FMA contraction) of the atom -> density
 * voxelization that gnina's CNN scoring path delegates to libmolgrid, plus the
 * smina-type -> (channel, radius) typer.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the shipped library never does.
 *
 * The arithmetic is NOT in /root/reference: it lives in libmolgrid
 * (github.com/gnina/libmolgrid, fetched unpinned at git HEAD by
 * /root/reference/CMakeLists.txt:145-160).  This file restates libmolgrid's published
 * algorithm (GridMaker::forward / calc_point / backward, FileMappedGninaTyper,
 * CoordinateSet::center) as summarised in SURVEY.md App. A and anchors it on the
 * reference's own call sites and goldens:
 *   call sites : gninasrc/lib/torch_model.cpp:108 (gmaker.initialize(res, dim, binary=false, rscale)),
 *                :112-113 (FileMappedGninaTyper), :120-142 (make_coordset), :163-168
 *                (lig.center(), CoordinateSet(rec,lig)), :175-181 (zeros + gmaker.forward),
 *                :200-206 (gmaker.backward); gninasrc/gninagrid/molgridder.cpp:48,86-94,100-138.
 *   goldens    : test/gninagrid/files/ccdx_0_{rec,lig}_*.dx, ccmap_0_*.map, ccbin_0_*.dx,
 *                ccgrid_0.25.29.binmap (tests/test_oracle_voxel_golden.py pins all of them, 1e-4,
 *                the tolerance of test/gninagrid/compare_dx.py:24).
 * Parity status: density formula, geometry, layout, summation, binary mode = PINNED by goldens.
 *   Hydrogen handling in center(), the exact float op order of d^2, and backward() = restated
 *   from the published libmolgrid source, NOT pinned by any in-tree vector ("parity unpinned").
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <ctype.h>

#define ORA_NUM_SMINA_TYPES 28

/* smina type names and xs_radius, restated from gninasrc/lib/atom_constants.h:45-75 (enum)
 * and :101-133 (default_data: column smina_name and column xs_radius). */
static const char *const ora_smina_names[ORA_NUM_SMINA_TYPES] = {
    "Hydrogen", "PolarHydrogen", "AliphaticCarbonXSHydrophobe", "AliphaticCarbonXSNonHydrophobe",
    "AromaticCarbonXSHydrophobe", "AromaticCarbonXSNonHydrophobe", "Nitrogen", "NitrogenXSDonor",
    "NitrogenXSDonorAcceptor", "NitrogenXSAcceptor", "Oxygen", "OxygenXSDonor",
    "OxygenXSDonorAcceptor", "OxygenXSAcceptor", "Sulfur", "SulfurAcceptor", "Phosphorus",
    "Fluorine", "Chlorine", "Bromine", "Iodine", "Magnesium", "Manganese", "Zinc", "Calcium",
    "Iron", "GenericMetal", "Boron"};

static const float ora_xs_radius[ORA_NUM_SMINA_TYPES] = {
    0.37f, 0.37f, 1.9f, 1.9f, 1.9f, 1.9f, 1.8f, 1.8f, 1.8f, 1.8f, 1.7f, 1.7f, 1.7f, 1.7f,
    2.0f,  2.0f,  2.1f, 1.5f, 1.8f, 2.0f, 2.2f, 1.2f, 1.2f, 1.2f, 1.2f, 1.2f, 1.2f, 1.92f};

const char *ora_smina_type_name(int smt) {
  return (smt >= 0 && smt < ORA_NUM_SMINA_TYPES) ? ora_smina_names[smt] : NULL;
}

float ora_smina_xs_radius(int smt) {
  return (smt >= 0 && smt < ORA_NUM_SMINA_TYPES) ? ora_xs_radius[smt] : 0.0f;
}

/* FileMappedGninaTyper (libmolgrid), as used at torch_model.cpp:110-113 and
 * molgridder.cpp:25-32: each line of the map text is a whitespace-separated set of smina
 * type NAMES sharing one channel; channel index = index of the (non-empty) line; names not
 * listed map to -1.  Returns the number of channels, or -1 on an unknown name.
 * chan_of_smt[28] receives the mapping. */
int ora_typer_parse(const char *map_text, int32_t *chan_of_smt) {
  for (int i = 0; i < ORA_NUM_SMINA_TYPES; i++) chan_of_smt[i] = -1;
  int nchan = 0;
  const char *p = map_text;
  while (*p) {
    const char *eol = strchr(p, '\n');
    size_t len = eol ? (size_t)(eol - p) : strlen(p);
    int found_on_line = 0;
    size_t i = 0;
    while (i < len) {
      while (i < len && isspace((unsigned char)p[i])) i++;
      size_t s = i;
      while (i < len && !isspace((unsigned char)p[i])) i++;
      if (i > s) {
        int hit = -1;
        for (int t = 0; t < ORA_NUM_SMINA_TYPES; t++) {
          if (strlen(ora_smina_names[t]) == i - s && strncmp(ora_smina_names[t], p + s, i - s) == 0) {
            hit = t;
            break;
          }
        }
        if (hit < 0) return -1;
        chan_of_smt[hit] = nchan;
        found_on_line = 1;
      }
    }
    if (found_on_line) nchan++;
    p += len;
    if (*p == '\n') p++;
  }
  return nchan;
}

/* make_coordset (torch_model.cpp:120-142): smt -> (channel or -1, xs_radius of the ORIGINAL type). */
void ora_type_atoms(const int32_t *smt, int n, const int32_t *chan_of_smt, int32_t *chan, float *radius) {
  for (int i = 0; i < n; i++) {
    int t = smt[i];
    if (t < 0 || t >= ORA_NUM_SMINA_TYPES) {
      chan[i] = -1;
      radius[i] = 0.0f;
    } else {
      chan[i] = chan_of_smt[t];
      radius[i] = ora_xs_radius[t];
    }
  }
}

/* CoordinateSet::center() (libmolgrid) as called at torch_model.cpp:165: arithmetic mean over
 * ALL rows of the ligand coordinate set (typed or not -- see SURVEY App. A.3, [R]); fp32
 * accumulation in index order, then one divide per axis.  only_typed != 0 selects the
 * alternative reading (rows with chan >= 0 only) kept switchable because no golden pins it. */
void ora_center(const float *xyz, const int32_t *chan, int n, int only_typed, float *center) {
  float sx = 0.0f, sy = 0.0f, sz = 0.0f;
  int cnt = 0;
  for (int i = 0; i < n; i++) {
    if (only_typed && chan && chan[i] < 0) continue;
    sx = sx + xyz[3 * i + 0];
    sy = sy + xyz[3 * i + 1];
    sz = sz + xyz[3 * i + 2];
    cnt++;
  }
  float fn = (float)(cnt > 0 ? cnt : 1);
  center[0] = sx / fn;
  center[1] = sy / fn;
  center[2] = sz / fn;
}

/* Grid points per side: N = round(dimension / resolution) + 1 (molgridder.cpp:48;
 * GridMaker::get_first_dim at torch_model.cpp:176).  23.5 / 0.5 -> 48. */
int ora_grid_points(float resolution, float dimension) {
  return (int)roundf(dimension / resolution) + 1;
}

/* Density of one atom at one grid point: libmolgrid GridMaker::calc_point with
 * gaussian_radius_multiple G = 1 -> final_radius_multiple = (1 + 2 G^2) / (2 G) = 1.5,
 * A = 4 e^-2, B = -12 e^-2, C = 9 e^-2  (pinned by the ccdx goldens).
 * (ar is radius * radius_scale.) */
typedef struct {
  float A, B, C;
  float final_mult;
  float gauss_mult;
} ora_density_consts;

static ora_density_consts ora_consts(void) {
  ora_density_consts k;
  const float G = 1.0f;
  k.gauss_mult = G;
  k.final_mult = (1.0f + 2.0f * G * G) / (2.0f * G);
  float e = expf(-2.0f * G * G);
  k.A = e * 4.0f * G * G;
  k.B = -e * (4.0f * G + 8.0f * G * G * G);
  k.C = e * (4.0f * G * G * G * G + 4.0f * G * G + 1.0f);
  return k;
}

static inline float ora_calc_point(float ax, float ay, float az, float ar, float gx, float gy, float gz,
                                   int binary, const ora_density_consts *k) {
  float dx = gx - ax;
  float dy = gy - ay;
  float dz = gz - az;
  float rsq = (dx * dx + dy * dy) + dz * dz;
  if (binary) return rsq < ar * ar ? 1.0f : 0.0f;
  float dist = sqrtf(rsq);
  if (dist >= ar * k->final_mult) return 0.0f;
  if (dist <= ar * k->gauss_mult) {
    float ex = (-2.0f * dist * dist) / (ar * ar);
    return expf(ex);
  }
  float dr = dist / ar;
  float q = (k->A * dr + k->B) * dr + k->C;
  return q > 0.0f ? q : 0.0f;
}

/* Inclusive-exclusive index range of grid points an atom can touch along one axis
 * (libmolgrid get_bounds_1D).  Any superset gives identical values because the density is
 * exactly 0 for dist >= 1.5 r; we keep one extra point each side. */
static inline void ora_bounds_1d(float origin, float coord, float maxr, float res, int N, int *lo, int *hi) {
  float l = (coord - maxr - origin) / res;
  float h = (coord + maxr - origin) / res;
  int il = (int)floorf(l) - 1;
  int ih = (int)ceilf(h) + 2;
  if (il < 0) il = 0;
  if (ih > N) ih = N;
  if (ih < il) ih = il;
  *lo = il;
  *hi = ih;
}

/* GridMaker::forward(center, CoordinateSet, Grid4f) (called torch_model.cpp:181).
 * out: float[n_channels][N][N][N], x slowest / z fastest (pinned), MUST be
 * zero-filled by the caller for Gaussian mode exactly as torch_model.cpp:179 does; this
 * function ADDS densities (atoms in one channel sum; binary mode SETS 1).
 * grid point (i,j,k) = origin + res*(i,j,k), origin = center - dimension/2.
 * Atoms with chan < 0 are skipped.  chan is the index in the COMBINED set (ligand channels
 * already offset by n_rec_channels, torch_model.cpp:168). */
void ora_grid_forward(const float *center, const float *xyz, const int32_t *chan, const float *radius, int n_atoms,
                      int n_channels, float resolution, float dimension, float radius_scale, int binary,
                      float *out) {
  const int N = ora_grid_points(resolution, dimension);
  const ora_density_consts k = ora_consts();
  const float half = dimension / 2.0f;
  const float ox = center[0] - half, oy = center[1] - half, oz = center[2] - half;
  for (int a = 0; a < n_atoms; a++) {
    int c = chan[a];
    if (c < 0 || c >= n_channels) continue;
    float ax = xyz[3 * a + 0], ay = xyz[3 * a + 1], az = xyz[3 * a + 2];
    float ar = radius[a] * radius_scale;
    float maxr = ar * k.final_mult;
    int x0, x1, y0, y1, z0, z1;
    ora_bounds_1d(ox, ax, maxr, resolution, N, &x0, &x1);
    ora_bounds_1d(oy, ay, maxr, resolution, N, &y0, &y1);
    ora_bounds_1d(oz, az, maxr, resolution, N, &z0, &z1);
    float *g = out + (size_t)c * N * N * N;
    for (int i = x0; i < x1; i++) {
      float gx = ox + (float)i * resolution;
      for (int j = y0; j < y1; j++) {
        float gy = oy + (float)j * resolution;
        for (int kk = z0; kk < z1; kk++) {
          float gz = oz + (float)kk * resolution;
          float v = ora_calc_point(ax, ay, az, ar, gx, gy, gz, binary, &k);
          size_t idx = ((size_t)i * N + j) * N + kk;
          if (binary) {
            if (v != 0.0f) g[idx] = 1.0f;
          } else {
            g[idx] = g[idx] + v;
          }
        }
      }
    }
  }
}

/* GridMaker::backward(center, coords, gridgrad, atomgrad) (called torch_model.cpp:203):
 * dL/dx_a = sum_ijk g[c][ijk] * rho'(d) * (x_a - p_ijk)/d  with rho' from SURVEY App. A.4:
 *   d <= r        : -4 d / r^2 * exp(-2 d^2 / r^2)
 *   r < d < 1.5 r : e^-2 (8 d / r^2 - 12 / r)       [= (2 A d/r + B)/r]
 *   else / d == 0 : 0
 * atomgrad: float[n_atoms][3], overwritten (zeros for untyped atoms). "parity unpinned". */
void ora_grid_backward(const float *center, const float *xyz, const int32_t *chan, const float *radius, int n_atoms,
                       int n_channels, float resolution, float dimension, float radius_scale,
                       const float *gridgrad, float *atomgrad) {
  const int N = ora_grid_points(resolution, dimension);
  const ora_density_consts k = ora_consts();
  const float half = dimension / 2.0f;
  const float ox = center[0] - half, oy = center[1] - half, oz = center[2] - half;
  for (int a = 0; a < n_atoms; a++) {
    atomgrad[3 * a + 0] = atomgrad[3 * a + 1] = atomgrad[3 * a + 2] = 0.0f;
    int c = chan[a];
    if (c < 0 || c >= n_channels) continue;
    float ax = xyz[3 * a + 0], ay = xyz[3 * a + 1], az = xyz[3 * a + 2];
    float ar = radius[a] * radius_scale;
    float maxr = ar * k.final_mult;
    int x0, x1, y0, y1, z0, z1;
    ora_bounds_1d(ox, ax, maxr, resolution, N, &x0, &x1);
    ora_bounds_1d(oy, ay, maxr, resolution, N, &y0, &y1);
    ora_bounds_1d(oz, az, maxr, resolution, N, &z0, &z1);
    const float *g = gridgrad + (size_t)c * N * N * N;
    float gx_acc = 0.0f, gy_acc = 0.0f, gz_acc = 0.0f;
    for (int i = x0; i < x1; i++) {
      float px = ox + (float)i * resolution;
      for (int j = y0; j < y1; j++) {
        float py = oy + (float)j * resolution;
        for (int kk = z0; kk < z1; kk++) {
          float pz = oz + (float)kk * resolution;
          float dx = ax - px, dy = ay - py, dz = az - pz;
          float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
          if (dist >= maxr || dist == 0.0f) continue;
          float d;
          if (dist <= ar * k.gauss_mult) {
            float ex = (-2.0f * dist * dist) / (ar * ar);
            d = (-4.0f * dist / (ar * ar)) * expf(ex);
          } else {
            d = (2.0f * k.A * (dist / ar) + k.B) / ar;
          }
          float gval = g[((size_t)i * N + j) * N + kk] * d / dist;
          gx_acc = gx_acc + gval * dx;
          gy_acc = gy_acc + gval * dy;
          gz_acc = gz_acc + gval * dz;
        }
      }
    }
    atomgrad[3 * a + 0] = gx_acc;
    atomgrad[3 * a + 1] = gy_acc;
    atomgrad[3 * a + 2] = gz_acc;
  }
}

/* TorchModel::forward's voxelization front half for ONE pose (torch_model.cpp:153-181):
 * type both sets, centre on the ligand (or the given centre when finite), build the combined
 * channel index and voxelize.  Convenience used by the parity tests and the CPU baseline.
 * center_in may be NULL / NaN -> ligand mean.  Returns N (grid points per side). */
int ora_voxelize_pose(const float *rec_xyz, const int32_t *rec_smt, int n_rec, const float *lig_xyz,
                      const int32_t *lig_smt, int n_lig, const int32_t *rec_chan_of_smt, int n_rec_ch,
                      const int32_t *lig_chan_of_smt, int n_lig_ch, const float *center_in, float resolution,
                      float dimension, float radius_scale, float *center_out, float *out) {
  const int N = ora_grid_points(resolution, dimension);
  const int C = n_rec_ch + n_lig_ch;
  int n = n_rec + n_lig;
  float *xyz = (float *)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
  int32_t *chan = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  float *rad = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  memcpy(xyz, rec_xyz, sizeof(float) * 3 * (size_t)n_rec);
  memcpy(xyz + 3 * (size_t)n_rec, lig_xyz, sizeof(float) * 3 * (size_t)n_lig);
  ora_type_atoms(rec_smt, n_rec, rec_chan_of_smt, chan, rad);
  ora_type_atoms(lig_smt, n_lig, lig_chan_of_smt, chan + n_rec, rad + n_rec);
  for (int i = 0; i < n_lig; i++)
    if (chan[n_rec + i] >= 0) chan[n_rec + i] += n_rec_ch;
  float center[3];
  if (center_in && isfinite(center_in[0])) {
    center[0] = center_in[0];
    center[1] = center_in[1];
    center[2] = center_in[2];
  } else {
    ora_center(lig_xyz, NULL, n_lig, 0, center);
  }
  if (center_out) {
    center_out[0] = center[0];
    center_out[1] = center[1];
    center_out[2] = center[2];
  }
  memset(out, 0, sizeof(float) * (size_t)C * N * N * N);
  ora_grid_forward(center, xyz, chan, rad, n, C, resolution, dimension, radius_scale, 0, out);
  free(xyz);
  free(chan);
  free(rad);
  return N;
}
