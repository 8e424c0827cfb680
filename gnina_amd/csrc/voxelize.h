// voxelize.h -- argument blocks for the voxelization kernels (voxelize.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mig {

// One typed atom, 32 bytes, with the density constants of its smina type (typer.h DensityConsts).
struct AtomRec {
  float x, y, z, ar;
  float t2, g2, kexp, inv_ar;
};

struct LigConsts {
  float ar, t2, g2, kexp, inv_ar;
};

struct GatherArgs {
  const AtomRec *rec;      // typed receptor atoms, sorted by channel (stable)
  const int *rec_chan;     // their channel in the combined set
  int n_rec;
  // flexible-residue atoms (DLScorer::setReceptor refreshes their coordinates on every call,
  // dl_scorer.cpp:181-192): per-pose coordinates replace the stored ones
  const int *rec_flex_slot;  // [n_rec] index into the flex rows, -1 = rigid; nullptr = no flexible atoms
  const float *flex_xyz;     // [B][n_flex][3]
  int n_flex;
  const float *lig_xyz;    // [B][L][3]
  int L;
  const int *lig_perm;     // [n_lig] index into the L ligand rows (typed atoms, sorted by channel)
  const LigConsts *lig_consts;  // [n_lig]
  const int *lig_chan;     // [n_lig] channel in the combined set (already offset by n_rec_channels)
  int n_lig;
  const unsigned char *lig_typed;  // [L] 1 if the row has a channel (for the typed-only centre switch)
  // ragged batches (virtual screening: every pose may belong to a different ligand): the four ligand arrays
  // above are then per pose, [B][L], and these give the per-pose counts
  const int *pose_rows;     // [B] real ligand rows of pose b (<= L), or nullptr = one ligand for the whole batch
  const int *pose_n_lig;    // [B] typed atoms of pose b
  const float *centers_in;  // [B][3] or nullptr; non-finite x -> ligand mean
  // Transform(gcenter, 0, rotate) (torch_model.cpp:170-173): unit quaternions [B][4] (a, b, c, d), every atom becomes
  // R(q)(x - centre) + centre before it is placed on the grid; nullptr = no rotation
  const float *rot;
  int center_typed_only;
  float half_dim;
  float *centers_out;  // [B][3]
  // Candidate lists, one per x-slab of 8 voxels (= one per tile index tx): the atoms whose support can reach
  // the slab, order preserving.  A tile wavefront scans only its slab's list (~2.4/6 of the pose's candidates
  // at 48^3) with the full sphere/box test; the slab test is the x part of that test, so nothing is lost.
  AtomRec *cand;       // [B][n_slab][cap]
  int *cand_chan;      // [B][n_slab][cap]
  int *cand_n;         // [B][n_slab]
  int cap;
  int n_slab;          // tiles per axis (<= kMaxSlabs), or 1 = a single list for the whole grid
  float res;
};

constexpr int kMaxSlabs = 16;

struct VoxArgs {
  const AtomRec *cand;
  const int *cand_chan;
  const int *cand_n;
  int cap;
  int n_slab;            // as in GatherArgs
  const float *centers;  // [B][3]
  int N;                 // grid points per side
  int tiles_per_axis;    // ceil(ceil(N/2) / 4)
  int C;                 // channels
  int Cp;                // channel stride of the pooled channels-last output (multiple of 4)
  float res, half_dim;
  float qa, qb, qc;      // quadratic tail coefficients (4 e^-2, -12 e^-2, 9 e^-2)
  float *out;
  unsigned char *argmax_out;  // pooled max mode: arg-max voxel (x*4+y*2+z) per (cell, channel), or nullptr
  // pooled modes: write the grid in the split-fp16 kernels' tensor format (conv3d.h ConvArgs::in_split) -- Cp a multiple of
  // 8, [pose][octet][cell][h0..h7 | l0..l7] fp16 with h = RN_f16(v), l = RN_f16(v - h) -- so that the first convolution
  // stages it by LDS-DMA.  overflow: the scorer's sticky range flag (a density beyond 65504, or a NaN).
  int split;
  unsigned *overflow;
  // split mode: occupancy bytes for the first convolution's zero skipping, [pose][tiles_per_axis]^3[8] -- byte w of a tile
  // (= a 4 x 4 x 4-cell block of the pooled grid) is non-zero iff window (octet) w of the tile holds a non-zero value
  unsigned char *occ;
  // timing experiments only (MI_VOX_DBG in a -DMI_VOX_TIMING build; wrong results): 1 = hits are found but not evaluated, 2 = flushes do no transpose /
  // pooling / staging, 4 = windows are not stored, 8 = no hit test (no hits)
  int dbg;
  // -DMI_VOX_TRAP builds only (tools/experiments/vox_stress.py; nullptr otherwise and ignored by the product build): a ring
  // of 16-dword records -- [0] = records written so far, record r at dwords 16 (r + 1) ... -- into which the tile kernel
  // reports what must never happen: a hit whose channel is below the current one, a record that differs between the scalar
  // and the vector load path, a window index outside the staged window, a damaged LDS canary
  unsigned *trap;
};

struct VoxBackArgs {
  const float *lig_xyz;  // [B][L][3]
  int L;
  const int *lig_perm;   // typed ligand atoms (sorted by channel)
  const LigConsts *lig_consts;
  const int *lig_chan;
  int n_lig;
  const float *centers;  // [B][3]
  const float *grad_pooled;     // [B][N/2]^3[Cp] gradient w.r.t. the pooled grid
  const unsigned char *argmax;  // [B][N/2]^3[Cp] (max pooling) or nullptr
  int N, Cp;
  float res, half_dim, qa, qb;
  float *lig_grad;  // [B][L][3]
  float scale;      // e.g. 1 / n_models
  int accumulate;   // add into lig_grad instead of overwriting
  const float *rot; // [B][4] rotation of the forward pass (see GatherArgs); the gradient is rotated back
};

void launch_gather(const GatherArgs &g, int B, hipStream_t s);
// mode 0: full grid [B][C][N][N][N] (out must be pre-zeroed); 1: max-pooled; 2: avg-pooled
// ([B][N/2]^3[Cp], fully written).
void launch_voxelize(const VoxArgs &v, int B, int mode, hipStream_t s);
void launch_voxel_backward(const VoxBackArgs &a, int B, int pool_mode, hipStream_t s);
// diagnostic: dwords of got[] that differ from want[]: log[0] += their number, log[1] += 1 if any, and the first
// (cap - 1) of them as {iter, index, got, want} from log[4] on
void launch_dword_compare(const unsigned *got, const unsigned *want, size_t n, int iter, int *log, int cap, hipStream_t s);

}  // namespace mig
