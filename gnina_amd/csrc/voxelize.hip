// voxelize.hip -- atom -> density voxelization for gfx950 (MI355X).
//
// Replaces libmolgrid's GridMaker::forward as called by gnina at
// gninasrc/lib/torch_model.cpp:175-181 (zeros + gmaker.forward) together with the host work the
// reference repeats per pose: make_coordset (:120-142), lig.center() (:163-166) and
// CoordinateSet(rec, lig) (:168).  Semantics follow SURVEY.md App. A (pinned by the gninagrid
// goldens through the CPU restatement under oracle/ -- test infrastructure, never linked here).
//
// Design (MI355X-first, not a translation of libmolgrid's CUDA kernel):
//  * The receptor is typed, filtered and channel-sorted ONCE (engine.cpp) and stays in HBM as
//    32-byte records carrying per-type density constants, so no pose ever re-types it.
//  * gather_pose_atoms: one workgroup per pose computes the grid centre (sequential fp32 mean,
//    bit-identical to the oracle) and compacts, order preserving, the atoms whose support
//    overlaps the pose's grid box -> a channel-sorted candidate list.
//  * voxelize_tiles: one 64-lane wavefront per 8x8x8-voxel tile. Lanes cooperatively test
//    64 candidates at a time against the tile (sphere/box), ballot the hits and broadcast each
//    hit with v_readlane, so the atom data is wave-uniform (SGPR) and each lane owns eight
//    voxels in registers: one per 4x4x4 sub-block of the tile while atoms are accumulated (an
//    instruction's 64 evaluations are then one compact cube, skipped as a whole when the atom
//    does not reach it), handed over through LDS to 2x2x2-cell ownership for the fused pooling.
//    No atomics: per voxel the sum runs in atom order, which makes
//    the result deterministic and equal to the oracle's up to the exp/sqrt approximations.
//  * Output is either the reference layout [B][C][N][N][N] (export path, mi_voxelize_batch) or,
//    for the CNN, the 2x2x2-pooled grid written straight from registers: every shipped network
//    starts with Max/AvgPool3d(2), so the full-resolution grid (15.5 MB/pose)
//    never touches HBM.  Two pooled formats: channels-last fp32 [B][N/2][N/2][N/2][Cp] (gradient,
//    fp32-MFMA and bf16 programs) and -- the default forward program -- the split-fp16 kernels'
//    octet-major tensor format [B][Cp/8][N/2][N/2][N/2][h0..h7 | l0..l7] (VoxArgs::split,
//    conv3d.h ConvArgs::in_split), which the first convolution stages by LDS-DMA.  Pooled tiles
//    are staged in LDS one window of channels (split: one octet) at a time and stored as 16-byte
//    coalesced runs.
//  * In/out decisions (which voxels are non-zero, gaussian vs quadratic zone) use thresholds on
//    the squared distance precomputed per atom type on the host (typer.cpp) so they are
//    bit-identical to the sqrtf-based reference arithmetic.
#include "voxelize.h"
#include "options.h"

#include <cstdlib>

#include "conv3d.h"  // xcd_contiguous_id

namespace mig {

constexpr int kVoxSplitWin = 8;  // channels per staged window of the split-format output: one octet (the format is octet-major)
// transpose buffer of voxelize_tiles (sub-block ownership -> cell ownership): dwords between the rows of two sub-blocks
// (64 lanes + the largest bank offset, 21) and its size
constexpr int kVoxTrStride = 96, kVoxTrFloats = 8 * kVoxTrStride;

__device__ __forceinline__ float rl_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// The candidate lists seen through the CONSTANT address space: they were written by gather_pose_atoms, an earlier
// launch, so inside voxelize_tiles they are read-only -- which the compiler cannot prove for a plain pointer (the output
// might alias it).  A uniform-address load through these pointers is an s_load into SGPRs (scalar data cache): the
// record of a hit then costs no VGPRs (nine v_readlane from a per-lane copy otherwise, which has to stay live through the
// hit loop -- the difference between 72 and 64 VGPRs, i.e. 7 and 8 waves per SIMD).
static_assert(sizeof(AtomRec) == 32, "AtomRec is read as one 8-dword scalar load");
typedef float f32x8 __attribute__((ext_vector_type(8)));  // one AtomRec: x y z ar t2 g2 kexp inv_ar
typedef const __attribute__((address_space(4))) f32x8 *ConstRecPtr;
typedef const __attribute__((address_space(4))) int *ConstIntPtr;

// Rotation of libmolgrid's Transform (rotate about the grid centre, no translation): rows of R(q) for a unit
// quaternion q = a + bi + cj + dk.
struct Rot3 {
  float m[3][3];
};
__device__ __forceinline__ Rot3 rot_of_quat(const float *q) {
  const float a = q[0], b = q[1], c = q[2], d = q[3];
  Rot3 r;
  r.m[0][0] = a * a + b * b - c * c - d * d, r.m[0][1] = 2.f * (b * c - a * d), r.m[0][2] = 2.f * (b * d + a * c);
  r.m[1][0] = 2.f * (b * c + a * d), r.m[1][1] = a * a - b * b + c * c - d * d, r.m[1][2] = 2.f * (c * d - a * b);
  r.m[2][0] = 2.f * (b * d - a * c), r.m[2][1] = 2.f * (c * d + a * b), r.m[2][2] = a * a - b * b - c * c + d * d;
  return r;
}
__device__ __forceinline__ void rot_about(const Rot3 &r, float cx, float cy, float cz, float &x, float &y, float &z) {
  const float dx = x - cx, dy = y - cy, dz = z - cz;
  x = (r.m[0][0] * dx + r.m[0][1] * dy + r.m[0][2] * dz) + cx;
  y = (r.m[1][0] * dx + r.m[1][1] * dy + r.m[1][2] * dz) + cy;
  z = (r.m[2][0] * dx + r.m[2][1] * dy + r.m[2][2] * dz) + cz;
}

// ---------------------------------------------------------------------------------------------
// gather_pose_atoms
// ---------------------------------------------------------------------------------------------
// (Two block sizes: 256 threads per pose for batches -- 0.050 ms per 1,024 poses, against 0.099 with 1,024 threads -- and
// 1,024 for small calls, where a pose's ~2,500 candidate atoms then take three passes of the block instead of ten, each
// three barriers and a round trip to L2 that a B = 1 call waits for: 31 -> 26 us.  The order of the lists is the atoms'
// index order whatever the block size.)
template <int kGatherThreads>
__global__ __launch_bounds__(kGatherThreads) void gather_pose_atoms(GatherArgs g) {
  constexpr int kGatherWaves = kGatherThreads / 64;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  __shared__ float s_center[3];
  __shared__ int s_wave_cnt[kMaxSlabs][kGatherWaves];
  __shared__ int s_base[kMaxSlabs];

  const float *lig = g.lig_xyz + (size_t)b * g.L * 3;
  const size_t po = g.pose_rows ? (size_t)b * g.L : 0;  // offset of this pose's ligand description
  const int rows = g.pose_rows ? g.pose_rows[b] : g.L;
  const int n_lig = g.pose_rows ? g.pose_n_lig[b] : g.n_lig;
  // The pose's coordinates go through LDS (one coalesced round trip): thread 0's sequential mean below read them from global
  // memory one dependent L1 hit at a time -- ~8 us of the 23 us this kernel takes in a per-pose call, where nothing hides it.
  constexpr int kLigLds = 3 * 512;
  __shared__ float s_lig[kLigLds];
  const bool lig_in_lds = 3 * rows <= kLigLds;
  if (lig_in_lds) {
    for (int i = tid; i < 3 * rows; i += kGatherThreads) s_lig[i] = lig[i];
    __syncthreads();
  }
  const int total = g.n_rec + n_lig;
  // atom i's record and channel (receptor atoms first, then the ligand's in voxelization order)
  auto fetch = [&](int i, AtomRec &a, int &ch) __attribute__((always_inline)) {
    ch = -1;
    if (i < total) {
      if (i < g.n_rec) {
        a = g.rec[i];
        ch = g.rec_chan[i];
        if (g.rec_flex_slot) {
          const int fs = g.rec_flex_slot[i];
          if (fs >= 0) {
            const float *f = g.flex_xyz + ((size_t)b * g.n_flex + fs) * 3;
            a.x = f[0], a.y = f[1], a.z = f[2];
          }
        }
      } else {
        int j = i - g.n_rec;
        int src = g.lig_perm[po + j];
        const LigConsts lc = g.lig_consts[po + j];
        if (lig_in_lds) a.x = s_lig[3 * src + 0], a.y = s_lig[3 * src + 1], a.z = s_lig[3 * src + 2];
        else a.x = lig[3 * src + 0], a.y = lig[3 * src + 1], a.z = lig[3 * src + 2];
        a.ar = lc.ar;
        a.t2 = lc.t2;
        a.g2 = lc.g2;
        a.kexp = lc.kexp;
        a.inv_ar = lc.inv_ar;
        ch = g.lig_chan[po + j];
      }
    }
  };
  // (the first pass's loads go out in front of thread 0's sequential mean)
  AtomRec a_next{};
  int ch_next;
  fetch(tid, a_next, ch_next);
  if (tid == 0) {
    float cx, cy, cz;
    bool given = false;
    if (g.centers_in) {
      cx = g.centers_in[3 * b + 0];
      cy = g.centers_in[3 * b + 1];
      cz = g.centers_in[3 * b + 2];
      given = isfinite(cx);
    }
    if (!given) {
      // CoordinateSet::center(): fp32 sum in index order, one divide per axis (oracle ora_center)
      float sx = 0.f, sy = 0.f, sz = 0.f;
      int cnt = 0;
      if (lig_in_lds && !g.center_typed_only) {  // (same additions in the same order)
        for (int i = 0; i < rows; i++) {
          sx = sx + s_lig[3 * i + 0];
          sy = sy + s_lig[3 * i + 1];
          sz = sz + s_lig[3 * i + 2];
        }
        cnt = rows;
      } else {
        for (int i = 0; i < rows; i++) {
          if (g.center_typed_only && g.lig_typed[po + i] == 0) continue;
          sx = sx + lig[3 * i + 0];
          sy = sy + lig[3 * i + 1];
          sz = sz + lig[3 * i + 2];
          cnt++;
        }
      }
      float fn = (float)(cnt > 0 ? cnt : 1);
      cx = sx / fn;
      cy = sy / fn;
      cz = sz / fn;
    }
    s_center[0] = cx;
    s_center[1] = cy;
    s_center[2] = cz;
    g.centers_out[3 * b + 0] = cx;
    g.centers_out[3 * b + 1] = cy;
    g.centers_out[3 * b + 2] = cz;
  }
  if (tid < kMaxSlabs) s_base[tid] = 0;
  __syncthreads();
  const float cx = s_center[0], cy = s_center[1], cz = s_center[2];
  Rot3 R;
  if (g.rot) R = rot_of_quat(g.rot + 4 * (size_t)b);

  // (the next pass's loads are issued before this pass's barriers: in a per-pose call nothing else hides their round trip)
  for (int base = 0; base < total; base += kGatherThreads) {
    int i = base + tid;
    bool keep = false;
    AtomRec a = a_next;
    int ch = ch_next;
    if (base + kGatherThreads < total) fetch(i + kGatherThreads, a_next, ch_next);
    if (i < total) {
      if (g.rot) rot_about(R, cx, cy, cz, a.x, a.y, a.z);
      float reach = g.half_dim + a.ar * 1.5f + 0.01f;
      keep = fabsf(a.x - cx) <= reach && fabsf(a.y - cy) <= reach && fabsf(a.z - cz) <= reach;
    }
    // per slab: does the atom's support reach it?  (the x part of voxelize_tiles' sphere/box test, same floats)
    unsigned slab_mask = 0u;
    if (keep) {
      if (g.n_slab == 1) {
        slab_mask = 1u;
      } else {
        const float ox = cx - g.half_dim, lim = a.t2 * 1.0001f + 1e-4f;
        for (int sl = 0; sl < g.n_slab; sl++) {
          const float tlox = ox + (float)(8 * sl) * g.res, thix = ox + (float)(8 * sl + 7) * g.res;
          const float ddx = fmaxf(0.f, fmaxf(tlox - a.x, a.x - thix));
          if (ddx * ddx <= lim) slab_mask |= 1u << sl;
        }
      }
    }
    int prefix[kMaxSlabs];
#pragma unroll 1
    for (int sl = 0; sl < g.n_slab; sl++) {
      const unsigned long long m = __ballot((slab_mask >> sl) & 1u);
      prefix[sl] = __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (lane == 0) s_wave_cnt[sl][wid] = __builtin_popcountll(m);
    }
    __syncthreads();
    for (int sl = 0; sl < g.n_slab; sl++) {
      if (!((slab_mask >> sl) & 1u)) continue;
      int off = s_base[sl];
      for (int w = 0; w < wid; w++) off += s_wave_cnt[sl][w];
      const size_t o = ((size_t)b * g.n_slab + sl) * g.cap + off + prefix[sl];
      g.cand[o] = a;
      g.cand_chan[o] = ch;
    }
    __syncthreads();
    if (tid < g.n_slab) {
      int n = 0;
      for (int w = 0; w < kGatherWaves; w++) n += s_wave_cnt[tid][w];
      s_base[tid] += n;
    }
    __syncthreads();
  }
  if (tid < g.n_slab) g.cand_n[(size_t)b * g.n_slab + tid] = s_base[tid];
}

// ---------------------------------------------------------------------------------------------
// voxelize_tiles
// ---------------------------------------------------------------------------------------------
// Correctly rounded sqrt of a normal float: v_sqrt_f32 is within one ulp, the two fma residuals pick the neighbour.
__device__ __forceinline__ float sqrt_rn(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
  const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
  const float r = r_dn <= 0.f ? s_dn : s;
  return r_up > 0.f ? s_up : r;
}
// Correctly rounded a / b given y = RN(1 / b) (typer.cpp computes inv_ar with an IEEE divide): two Markstein steps --
// the first makes the quotient faithful, the second rounds it correctly.
__device__ __forceinline__ float div_rn(float a, float b, float y) {
  const float q0 = a * y;
  const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, b, a), y, q0);
  return __builtin_fmaf(__builtin_fmaf(-q1, b, a), y, q1);
}

// acc += density of one atom (wave-uniform constants) at squared distance rsq; SURVEY App. A.2.  The accumulation sits
// INSIDE the zone branches: a lane outside the atom's support executes nothing after the range test -- written as
// "v = 0; if (...) v = ...; acc += v" the compiler emits a zero-initialisation and an add for every sub-block an atom
// does not reach (five of eight on average), on a kernel whose time is its VALU instruction count.  Adding nothing and
// adding +0 give the same bits (the accumulators are sums of non-negative terms and start at +0).
#ifdef MI_VOX_TRAP
#define VOX_TRAP_VIOL_PARAM , bool &viol
#define VOX_TRAP_VIOL_ARG , viol
#else
#define VOX_TRAP_VIOL_PARAM
#define VOX_TRAP_VIOL_ARG
#endif
// MI_VOX_FIX: how the hit loop forms its eight squared distances.  1 (the product): scalar fp32 instructions.  0: the round-5
// form, packed fp32 (v_pk_add_f32 / v_pk_mul_f32 on float2 values: half the instructions) -- which, built WITH packed-fp32
// instructions (tools/experiments/build_variant.sh: the product's flags switch them off), gives wrong distances in lanes
// 32-63 of a wavefront in ~2 % of the launches that run next to a second scorer's Dense conv kernels (round 6, DESIGN
// "concurrency").  2 .. 5: the builds that told what it was NOT (wait states behind the packed results, behind the compare,
// SGPR operands of the packed instructions, a scalar load in flight): all four deviate like 0.
#ifndef MI_VOX_FIX
#define MI_VOX_FIX 1
#endif
__device__ __forceinline__ void density_add(float &acc, float rsq, float t2, float g2, float kexp, float ar, float inv_ar,
                                            float qa, float qb, float qc VOX_TRAP_VIOL_PARAM) {
  // (round 5: straight-line inside the support test.  The kernel issues MORE scalar than vector instructions -- the
  // exec-mask bookkeeping of nested zone branches was 15 SALU instructions per evaluated sub-block next to 13 VALU -- and a
  // sub-block that an atom reaches almost always has lanes in both zones, so both sides ran anyway.  What is left: the
  // support test and the thin-shell redo.  Same operations per lane, same bits: a lane whose tail value is <= 0 adds +0.)
  // (a sub-block nobody reaches -- five of eight -- costs v_cmp + s_cbranch_vccz this way; "if (rsq < t2)" alone compiles to
  // s_and_saveexec, s_cbranch_execz and an s_or at the join)
  if (__builtin_amdgcn_ballot_w64(rsq < t2) == 0ull) return;
#if MI_VOX_FIX == 3
  asm volatile("s_nop 7");  // (experiment: wait states between the v_cmp and the s_and_saveexec that reads its mask)
#else
  asm volatile("");
#endif
  if (rsq < t2) {
#ifdef MI_VOX_TRAP
    {  // a lane in here whose own distance is outside the support: the exec mask is not the compare's result
      float rsq_again = rsq;
      asm volatile("s_nop 4\n\tv_mov_b32 %0, %0" : "+v"(rsq_again));
      viol |= !(rsq_again < t2);
    }
#endif
    const bool gauss = rsq <= g2;
    // (the exponential only where a lane of the sub-block is in the Gaussian zone -- the shell between r and 1.5 r is 70 %
    // of an atom's support, and v_exp_f32 is a quarter-rate instruction; the empty asm keeps the compiler from turning
    // the wave-uniform branch back into a select)
    float gv = 0.f;
    if (__builtin_amdgcn_ballot_w64(gauss) != 0ull) {
      gv = __builtin_amdgcn_exp2f(rsq * kexp);
      asm volatile("" : "+v"(gv));
    }
    float dr = __builtin_amdgcn_sqrtf(rsq) * inv_ar;
    float q = (qa * dr + qb) * dr + qc;
    if (!gauss && q < 4e-6f) {
      // thin shell next to the 1.5 r cut-off, where the tail is ~1e-7 and its SIGN decides whether
      // the voxel is non-zero: redo it with the reference's exact operation sequence
      // (correctly rounded sqrtf and divide, unfused polynomial) so the support set is bit-exact.
      // (sqrt_rn / div_rn: the same correctly rounded results in 14 instructions instead of the 28 of the generic
      // sequences, which also handle denormals and scale their inputs -- rsq is 1..12 A^2 here.  Checked exhaustively
      // against sqrtf and '/' for every float in [0.25, 64) x 414 radii: tools/microbench/exact_sqrt_div_check.hip)
      float dre = div_rn(sqrt_rn(rsq), ar, inv_ar);
      q = (qa * dre + qb) * dre + qc;
      // (only here can the value be <= 0 -- everywhere else it is >= 4e-6: the reference's "if (q > 0)" as a clamp, +0 added)
      asm("v_max_f32 %0, 0, %0" : "+v"(q));
    }
    acc = acc + (gauss ? gv : q);
  }
}

// one pooled value in the split-fp16 tensor format: h | l << 16, h = RN_f16(x), l = RN_f16(x - h) (x - h is exact in fp32;
// v_fma_mix_f32 reads h as fp16).  Densities are sums of at most a few hundred terms <= 1: no clamp, but the caller flags
// anything beyond the fp16 range.
typedef _Float16 vox_f16x2 __attribute__((ext_vector_type(2)));
typedef float vox_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned vox_split1(float x) {
  const vox_f32x2 a = {x, 0.f};
  const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(a, vox_f16x2));
  float r;
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(x));
  const vox_f32x2 b = {x, r};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(b, vox_f16x2));
}

// The MI_VOX_DBG timing switches (wrong results; LAB.md §3.10's breakdown) exist only in a build with -DMI_VOX_TIMING: the
// tile kernel is bound by its scalar instruction count, and four tests of a kernel argument per hit / flush / window are
// part of it.
#ifdef MI_VOX_TIMING
#define VOX_DBG(bit) (v.dbg & (bit))
#else
#define VOX_DBG(bit) false
#endif

// -DMI_VOX_TRAP (a diagnostic build, tools/experiments/vox_stress.py): the tile kernel reports what must never happen into
// VoxArgs::trap.  Record: kind, pose, tile, six words of detail, HW_ID, XCC_ID, s_memtime.
#ifdef MI_VOX_TRAP
constexpr int kVoxTrapPad = 64;  // canary dwords in front of and behind the kernel's LDS
__device__ __forceinline__ void vox_trap(unsigned *ring, unsigned kind, unsigned pose, unsigned tile, unsigned a, unsigned b,
                                         unsigned c, unsigned d, unsigned e, unsigned f) {
  if (!ring || threadIdx.x != 0) return;
  const unsigned slot = atomicAdd(ring, 1u);
  if (slot >= 1023u) return;
  unsigned *r = ring + 16 * (slot + 1);
  const unsigned long long t = __builtin_amdgcn_s_memtime();
  r[0] = kind, r[1] = pose, r[2] = tile, r[3] = a, r[4] = b, r[5] = c, r[6] = d, r[7] = e, r[8] = f;
  r[9] = __builtin_amdgcn_s_getreg(0xF804);   // HW_REG_HW_ID: wave / SIMD / CU / SH / SE ...
  r[10] = __builtin_amdgcn_s_getreg(0xF814);  // HW_REG_XCC_ID
  r[11] = (unsigned)t, r[12] = (unsigned)(t >> 32);
}
#else
constexpr int kVoxTrapPad = 0;
#endif

// MODE 0: full grid [B][C][N][N][N]; 1: max-pooled, 2: avg-pooled, channels last.  SPLIT (pooled modes): VoxArgs::split.
template <int MODE, bool SPLIT = false>
__global__ __launch_bounds__(64, 8) void voxelize_tiles(VoxArgs v) {
  static_assert(!SPLIT || MODE != 0, "the split format is a pooled, channels-last format");
  const int lane = threadIdx.x;
  const int ntile = v.tiles_per_axis;
  // 1-D grid of B * tiles workgroups, re-numbered so that an XCD (private L2) sees whole poses: the 216 tile
  // wavefronts of a pose all read that pose's candidate list
  const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int b = wg / (ntile * ntile * ntile);
  const int tile_id = wg - b * (ntile * ntile * ntile);
  const int tz = tile_id % ntile, ty = (tile_id / ntile) % ntile, tx = tile_id / (ntile * ntile);
  const int tile_occ = (tx * ntile + ty) * ntile + tz;  // index of this tile's occupancy bytes (VoxArgs::occ)
  // Two lane <-> voxel maps.  While atoms are accumulated a lane owns the voxel at local position (lx, ly, lz) of each of
  // the tile's eight 4x4x4 SUB-BLOCKS (acc[k], k = sub-block): the 64 evaluations of one instruction then are one compact
  // 2 A cube, and an atom that touches the tile reaches only 3.2 of the 8 cubes on average -- the others are skipped by
  // the wave-uniform exec-mask branches around density_add().  For pooling / output a lane owns the 2x2x2 CELL (cx, cy, cz);
  // the accumulators change hands through a 2 KB LDS transpose per flushed channel (s_tr).
  const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;

  // MODE != 0: the pooled tile is staged for 16-byte stores one WINDOW of kWin channels at a time ([64 cells][kWin]); the
  // candidate list is sorted by channel, so a window is complete when the first atom of a later window arrives.  (The
  // whole [64][Cp] tile was 9 KB: 14 single-wave workgroups per CU.  The kernel is latency-bound -- its time is inversely
  // proportional to the waves in flight, measured by padding the LDS request -- and 5 KB lets the 68-VGPR limit of 7
  // waves per SIMD decide.)
  // (split format: a window is whole octets -- one: 4 KB of LDS with the transpose buffer, like the fp32 windows' 5 KB)
  constexpr int kWin = SPLIT ? kVoxSplitWin : 12;
  extern __shared__ __attribute__((aligned(16))) float s_lds[];  // [64][kWin] (+ transpose buffer, arg-max bytes)
  float *const s_stage = s_lds + kVoxTrapPad;
  const int Cp = v.Cp;
  const int nwin = (Cp + kWin - 1) / kWin;
  float *s_tr = s_stage + 64 * kWin;                                            // [8 sub-blocks][kTrStride] (see tr_put below)
  unsigned char *s_arg = reinterpret_cast<unsigned char *>(s_tr + kVoxTrFloats); // [64][kWin], only if argmax_out
#ifdef MI_VOX_TRAP
  // canaries: 64 dwords in front of the stage and 64 behind the transpose buffer (split builds: no arg-max bytes)
  unsigned *const s_can0 = reinterpret_cast<unsigned *>(s_lds), *const s_can1 = reinterpret_cast<unsigned *>(s_tr + kVoxTrFloats);
  if constexpr (SPLIT) {
    s_can0[lane] = 0xC0DE0000u | (unsigned)lane;
    s_can1[lane] = 0xCAFE0000u | (unsigned)lane;
  }
  auto check_canaries = [&](unsigned where) {
    if constexpr (SPLIT) {
      const unsigned a0 = s_can0[lane], a1 = s_can1[lane];
      const unsigned long long bad0 = __ballot(a0 != (0xC0DE0000u | (unsigned)lane)), bad1 = __ballot(a1 != (0xCAFE0000u | (unsigned)lane));
      if (bad0 | bad1) {
        const int l0 = bad0 ? __builtin_ctzll(bad0) : 0, l1 = bad1 ? __builtin_ctzll(bad1) : 0;
        vox_trap(v.trap, 3u, blockIdx.x, where, (unsigned)bad0, (unsigned)(bad0 >> 32), (unsigned)bad1, (unsigned)(bad1 >> 32),
                 (unsigned)__builtin_amdgcn_readlane((int)a0, l0), (unsigned)__builtin_amdgcn_readlane((int)a1, l1));
      }
    }
  };
#endif
  // The transpose buffer, bank-conflict free in both directions (round 5; the [lane][8] layout cost 4 LDS cycles per
  // ds_read_b32 lane group instead of 1 -- SQ_LDS_BANK_CONFLICT was 0.68 of the kernel's LDS-active cycles, next to a VALU
  // 76 % busy).  Writer lane L = lx + 4 ly + 16 lz puts acc[k] at dword  kVoxTrStride k + F(k, L) + L,
  // F = (k & 1) + 4 ((k >> 1) & 1) + 16 (L >> 5): for one k the 32 lanes of a ds_write_b32 lane group hit 32 consecutive
  // dwords.  The r-th voxel (x*4 + y*2 + z) of this lane's pooling cell (cx, cy, cz) is (2 cx + rx, 2 cy + ry, 2 cz + rz),
  // written by lane L' = (x & 3) + 4 (y & 3) + 16 (z & 3) as its sub-block k' = (x >> 2, y >> 2, z >> 2): the reading lane's
  // bits (b5 .. b0) = (cx, cy, cz) give L' = (2 b4 + 8 b2 + 32 b0) + (rx + 4 ry + 16 rz) and k' = 4 b5 + 2 b3 + b1, so the
  // 32 lanes of a ds_read_b32 group (b5 fixed) read dwords that are 2 b4 + 8 b2 + b1 + 4 b3 + 16 b0 + const modulo 32 --
  // all 32 banks -- from one per-lane base plus a compile-time offset per r (an immediate of the ds_read).
  const int tw_base = lane + 16 * (lane >> 5);  // writer: + kVoxTrStride k + (k & 1) + 4 ((k >> 1) & 1), immediates
  const int tbase = (4 * ((lane >> 5) & 1) + 2 * ((lane >> 3) & 1) + ((lane >> 1) & 1)) * kVoxTrStride + ((lane >> 1) & 1) +
                    4 * ((lane >> 3) & 1) + 16 * (lane & 1) + 2 * ((lane >> 4) & 1) + 8 * ((lane >> 2) & 1) + 32 * (lane & 1);
  // pooled output, tile-invariant per lane: item i = lane + 64 k of a window is float4 `part` of cell i / (kWin / 4)
  // (staged at s_stage + 4 i); eo[k] = its float offset inside the pose without the window's channel base, part in
  // the two low bits (the offset is a multiple of 4), all ones = outside the grid
  unsigned eo[kWin / 4];
  if (MODE != 0) {
    const int S = v.N / 2;
#pragma unroll
    for (int k = 0; k < kWin / 4; k++) {
      const int i = lane + 64 * k;
      const int cell = i / (kWin / 4), part = i - cell * (kWin / 4);  // cell = cx * 16 + cy * 4 + cz, like the owner lane
      const int rx = tx * 4 + (cell >> 4), ry = ty * 4 + ((cell >> 2) & 3), rz = tz * 4 + (cell & 3);
      // (split format: [octet][cell][8 words], a window = one octet: the cell's stride is 8 words, the window's S^3 * 8)
      eo[k] = (rx < S && ry < S && rz < S) ? ((unsigned)(((rx * S + ry) * S + rz) * (SPLIT ? kWin : Cp) + 4 * part) | (unsigned)part) : 0xffffffffu;
    }
  }
  auto clear_window = [&]() {
#pragma unroll
    for (int k = 0; k < kWin / 4; k++) *reinterpret_cast<float4 *>(s_stage + (lane * (kWin / 4) + k) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 1 && v.argmax_out)
#pragma unroll
      for (int k = 0; k < kWin / 4; k++) reinterpret_cast<unsigned *>(s_arg)[lane * (kWin / 4) + k] = 0u;
  };
  if (MODE != 0) clear_window();

  const float ctrx = v.centers[3 * b + 0], ctry = v.centers[3 * b + 1], ctrz = v.centers[3 * b + 2];
  const float ox = ctrx - v.half_dim, oy = ctry - v.half_dim, oz = ctrz - v.half_dim;
  // grid point (i,j,k) = origin + (float)i * res   (oracle ora_grid_forward)
  float gx[2], gy[2], gz[2];
#pragma unroll
  for (int d = 0; d < 2; d++) {
    gx[d] = ox + (float)(8 * tx + 4 * d + lx) * v.res;
    gy[d] = oy + (float)(8 * ty + 4 * d + ly) * v.res;
    gz[d] = oz + (float)(8 * tz + 4 * d + lz) * v.res;
  }
  // tile bounding box in space (first/last voxel of the 8x8x8 tile)
  const float tlox = ox + (float)(8 * tx) * v.res, thix = ox + (float)(8 * tx + 7) * v.res;
  const float tloy = oy + (float)(8 * ty) * v.res, thiy = oy + (float)(8 * ty + 7) * v.res;
  const float tloz = oz + (float)(8 * tz) * v.res, thiz = oz + (float)(8 * tz + 7) * v.res;

  const int slab = v.n_slab > 1 ? tx : 0;
  const AtomRec *cand = v.cand + ((size_t)b * v.n_slab + slab) * v.cap;
  const int *cand_chan = v.cand_chan + ((size_t)b * v.n_slab + slab) * v.cap;
  const int n = v.cand_n[(size_t)b * v.n_slab + slab];
  const ConstRecPtr candc = (ConstRecPtr)(const void *)cand;
  const ConstIntPtr chanc = (ConstIntPtr)(const void *)cand_chan;

  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = 0.f;
  int cur = -1;

  int cur_w = 0;
  // window cur_w of the staged tile -> out[b][cx][cy][cz][cur_w * kWin ...]: 64 cells x kWin / 4 float4, then cleared
  auto emit_window = [&]() {
    if (VOX_DBG(4)) {  // (timing only)
      cur_w++;
      return;
    }
    __builtin_amdgcn_s_waitcnt(0);  // wave-synchronous: the LDS writes of flush() complete before the reads
    __builtin_amdgcn_wave_barrier();
#ifdef MI_VOX_TRAP
    check_canaries(0x100u + (unsigned)cur_w);
#endif
    const int S = v.N / 2;
    const int c0 = cur_w * kWin;
    // (uniform 64-bit base of the pose + a 32-bit offset inside it -- a pooled pose is at most 48^3 x 36 floats: the
    // store's address is then an SGPR base and one VGPR offset instead of 64-bit multiplies per lane and window; this
    // kernel's time is its VALU instruction count)
    const size_t pose_off = (size_t)b * S * S * S * Cp;
    float *out_pose = v.out + pose_off;
    unsigned any_bits = 0u;
#pragma unroll
    for (int k = 0; k < kWin / 4; k++) {
      const int part = (int)(eo[k] & 3u);
      if (eo[k] != 0xffffffffu && c0 + 4 * part < Cp) {
        const unsigned o = (eo[k] & ~3u) + (SPLIT ? (unsigned)(cur_w * S * S * S * kWin) : (unsigned)c0);
        float4 item;
        if constexpr (SPLIT) {  // item i = lane + 64 k is dwords 4 part .. 4 part + 3 of cell i >> 1 (transposed stage: stage_put)
          const int cell = (lane + 64 * k) >> 1;  // (part == i & 1)
          const float *src = s_stage + 256 * part + ((cell + 16 * part) & 63);
          item = make_float4(src[0], src[64], src[128], src[192]);
        } else {
          // (cell * kWin + 4 * part == 4 * i: the staged tile is read back in item order)
          item = *reinterpret_cast<const float4 *>(s_stage + 4 * (lane + 64 * k));
        }
        if constexpr (SPLIT) any_bits |= __float_as_uint(item.x) | __float_as_uint(item.y) | __float_as_uint(item.z) | __float_as_uint(item.w);
        *reinterpret_cast<float4 *>(out_pose + o) = item;
        if (MODE == 1 && v.argmax_out)
          *reinterpret_cast<unsigned *>(v.argmax_out + pose_off + o) = *reinterpret_cast<const unsigned *>(s_arg + 4 * (lane + 64 * k));
      }
    }
    if constexpr (SPLIT) {  // occupancy byte of this (tile, octet): the first convolution stages only what has content
      if (v.occ && cur_w < 8) {
        const bool nz = __ballot(any_bits != 0u) != 0ull;
        if (lane == 0) v.occ[((size_t)b * ntile * ntile * ntile + tile_occ) * 8 + cur_w] = nz ? 1 : 0;
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    clear_window();
    cur_w++;
  };
  // channel cw of the window under construction, this lane's cell: an fp32 word, or (SPLIT) the h and l halves at their
  // places in the cell's [octet][h8 | l8] row
  bool ovf = false;
  auto stage_put = [&](int cw, float val) {
    if constexpr (SPLIT) {
      // The staged octet is kept TRANSPOSED -- dword d of cell c's 32-byte record [h0..h7 | l0..l7] at dword
      // 64 d + ((c + 16 (d >> 2)) & 63) -- so that the 32 lanes of a store group hit 32 different banks (as [cell][8 dwords]
      // the two 16-bit stores of every flush were 8-way conflicts) and so do the 32 lanes of emit_window's reads.
      ovf |= !(val <= 65504.f);
      const unsigned hl = vox_split1(val);
      unsigned short *st16 = reinterpret_cast<unsigned short *>(s_stage);
      const int d = cw >> 1, hf = cw & 1;  // (wave-uniform)
      st16[(64 * d + lane) * 2 + hf] = (unsigned short)(hl & 0xffffu);
      st16[(64 * (4 + d) + ((lane + 16) & 63)) * 2 + hf] = (unsigned short)(hl >> 16);
    } else {
      s_stage[lane * kWin + cw] = val;
    }
  };
  auto flush = [&](int c) {
    if (c < 0) return;
    if (MODE == 0) {
      float *o = v.out + ((size_t)b * v.C + c) * v.N * v.N * v.N;
#pragma unroll
      for (int dx = 0; dx < 2; dx++)
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
          for (int dz = 0; dz < 2; dz++) {
            int i = 8 * tx + 4 * dx + lx, j = 8 * ty + 4 * dy + ly, k = 8 * tz + 4 * dz + lz;
            if (i < v.N && j < v.N && k < v.N) o[((size_t)i * v.N + j) * v.N + k] = acc[dx * 4 + dy * 2 + dz];
          }
      return;
    }
    if (VOX_DBG(2)) {  // (timing only)
      while (cur_w < c / kWin) emit_window();
      return;
    }
    // sub-block ownership -> cell ownership (wave-synchronous LDS round trip; LDS executes a wave's accesses in order)
#pragma unroll
    for (int k = 0; k < 8; k++) s_tr[tw_base + k * kVoxTrStride + (k & 1) + 4 * ((k >> 1) & 1)] = acc[k];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    float cv[8];
#pragma unroll
    for (int r = 0; r < 8; r++) cv[r] = s_tr[tbase + (r >> 2) + 4 * ((r >> 1) & 1) + 16 * (r & 1)];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (MODE == 1) {
      float m;
      int am = 0;
      if (v.argmax_out) {  // gradient program: the arg-max travels with the value
        m = cv[0];
#pragma unroll
        for (int i = 1; i < 8; i++)
          if (cv[i] > m) {  // first maximum in (kd,kh,kw) scan order, like max_pool3d
            m = cv[i];
            am = i;
          }
      } else {
        // forward only: the same maximum from three v_max3_f32 (densities are finite and non-negative; the compare
        // chain above costs a v_cmp, a v_cndmask and the SGPR hazard slots between them per voxel)
        float m1, m2;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(cv[0]), "v"(cv[1]), "v"(cv[2]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m2) : "v"(cv[3]), "v"(cv[4]), "v"(cv[5]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(m1), "v"(cv[6]), "v"(cv[7]));
        asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m1), "v"(m2));
      }
      while (cur_w < c / kWin) emit_window();
#ifdef MI_VOX_TRAP
      if (c - cur_w * kWin < 0 || c - cur_w * kWin >= kWin) vox_trap(v.trap, 5u, blockIdx.x, (unsigned)tile_id, (unsigned)c, (unsigned)cur_w, (unsigned)cur, 0u, 0u, 0u);
#endif
      stage_put(c - cur_w * kWin, m);
      if (!SPLIT && v.argmax_out) s_arg[lane * kWin + (c - cur_w * kWin)] = (unsigned char)am;
    } else {
      float s = cv[0];
#pragma unroll
      for (int i = 1; i < 8; i++) s = s + cv[i];  // (kd,kh,kw) order, then /8 like avg_pool3d
      while (cur_w < c / kWin) emit_window();
      stage_put(c - cur_w * kWin, s * 0.125f);
    }
  };

  for (int base = 0; base < n; base += 64) {
    const int idx = base + lane;
    bool hit = false;
    if (idx < n) {
      const AtomRec a = cand[idx];
      float ddx = fmaxf(0.f, fmaxf(tlox - a.x, a.x - thix));
      float ddy = fmaxf(0.f, fmaxf(tloy - a.y, a.y - thiy));
      float ddz = fmaxf(0.f, fmaxf(tloz - a.z, a.z - thiz));
      float d2 = ddx * ddx + ddy * ddy + ddz * ddz;
      hit = d2 <= a.t2 * 1.0001f + 1e-4f;  // conservative superset of "some voxel has rsq < t2"
      if (VOX_DBG(8)) hit = false;  // (timing only)
    }
    unsigned long long mask = __ballot(hit);
    // the record of a hit comes back through the scalar cache, the next hit's record is requested before this one is
    // evaluated
    f32x8 rec_n = {0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f};
    int c_n = -1;
#ifdef MI_VOX_TRAP
    unsigned i_n = 0u;
#endif
    // (byte offsets in 32 bits: s_load with an SGPR offset instead of two 64-bit shift-and-add sequences per hit)
    typedef const __attribute__((address_space(4))) char *ConstBytePtr;
    auto fetch = [&](int src) {
      const unsigned i = (unsigned)(base + src);
      rec_n = *(ConstRecPtr)((ConstBytePtr)candc + (i << 5));
      c_n = *(ConstIntPtr)((ConstBytePtr)chanc + (i << 2));
#ifdef MI_VOX_TRAP
      i_n = i;
#endif
    };
    if (mask) fetch(__builtin_ctzll(mask));
    while (mask) {
      mask &= mask - 1;
      const f32x8 rec = rec_n;
      const int c = c_n;
#ifdef MI_VOX_TRAP
      {
        const unsigned i_cur = i_n;
        // the same record through the vector path, past the non-coherent caches (agent-scope atomic loads: sc1)
        const int c_vec = __hip_atomic_load(cand_chan + i_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned x_vec = __hip_atomic_load(reinterpret_cast<const unsigned *>(cand + i_cur), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned t2_vec = __hip_atomic_load(reinterpret_cast<const unsigned *>(cand + i_cur) + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool differ = c_vec != c || x_vec != __float_as_uint(rec[0]) || t2_vec != __float_as_uint(rec[4]);
        if (__ballot(differ) != 0ull)
          vox_trap(v.trap, 2u, blockIdx.x, (unsigned)tile_id, i_cur, (unsigned)c, (unsigned)__builtin_amdgcn_readfirstlane(c_vec), __float_as_uint(rec[0]),
                   (unsigned)__builtin_amdgcn_readfirstlane((int)x_vec), (unsigned)n);
        if (c < cur) vox_trap(v.trap, 1u, blockIdx.x, (unsigned)tile_id, i_cur, (unsigned)c, (unsigned)cur, (unsigned)cur_w, (unsigned)__builtin_amdgcn_readfirstlane(c_vec), (unsigned)n);
      }
#endif
#if MI_VOX_FIX != 5
      if (mask) fetch(__builtin_ctzll(mask));
#endif
      const float ax = rec[0], ay = rec[1], az = rec[2], ar = rec[3], t2 = rec[4], g2 = rec[5], kexp = rec[6],
                  inv_ar = rec[7];
      if (c != cur) {
        flush(cur);
        cur = c;
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = 0.f;
      }
      if (VOX_DBG(1)) continue;  // (timing only)
#ifdef MI_VOX_TRAP
      bool viol = false;
#endif
#if MI_VOX_FIX == 1
      // squared distances of this lane's eight voxels, (dx^2 + dy^2) + dz^2 like the oracle, in scalar fp32 instructions (see
      // MI_VOX_FIX above: the packed form computed the same bits with half the instructions -- until a neighbour queue's conv
      // kernels shared the SIMD; measured cost of the scalar form: none, 0.872 -> 0.871 ms)
      float dx2[2], dy2[2], dz2[2];
#pragma unroll
      for (int d = 0; d < 2; d++) {
        const float ddx = gx[d] - ax, ddy = gy[d] - ay, ddz = gz[d] - az;
        dx2[d] = ddx * ddx, dy2[d] = ddy * ddy, dz2[d] = ddz * ddz;
      }
#pragma unroll
      for (int dx = 0; dx < 2; dx++)
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
          for (int dz = 0; dz < 2; dz++)
            density_add(acc[dx * 4 + dy * 2 + dz], (dx2[dx] + dy2[dy]) + dz2[dz], t2, g2, kexp, ar, inv_ar, v.qa, v.qb, v.qc VOX_TRAP_VIOL_ARG);
#else
      const vox_f32x2 gx2 = {gx[0], gx[1]}, gy2 = {gy[0], gy[1]}, gz2 = {gz[0], gz[1]};
#if MI_VOX_FIX == 4
      // (experiment: the atom's coordinates in VGPRs -- no packed instruction reads an SGPR)
      float axv, ayv, azv;
      asm volatile("v_mov_b32 %0, %1" : "=v"(axv) : "s"(ax));
      asm volatile("v_mov_b32 %0, %1" : "=v"(ayv) : "s"(ay));
      asm volatile("v_mov_b32 %0, %1" : "=v"(azv) : "s"(az));
      const vox_f32x2 ax2 = {axv, axv}, ay2 = {ayv, ayv}, az2 = {azv, azv};
#else
      const vox_f32x2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az};
#endif
      vox_f32x2 dxx = gx2 - ax2, dyy = gy2 - ay2, dzz = gz2 - az2;
      dxx = dxx * dxx, dyy = dyy * dyy, dzz = dzz * dzz;
      const vox_f32x2 dy0 = {dyy[0], dyy[0]}, dy1 = {dyy[1], dyy[1]}, dz0 = {dzz[0], dzz[0]}, dz1 = {dzz[1], dzz[1]};
      const vox_f32x2 xy0 = dxx + dy0, xy1 = dxx + dy1;          // [dx] for dy = 0 / 1
      vox_f32x2 r00 = xy0 + dz0, r01 = xy0 + dz1, r10 = xy1 + dz0, r11 = xy1 + dz1;  // r<dy><dz>[dx]
#if MI_VOX_FIX == 2
      asm volatile("s_nop 7" : "+v"(r00), "+v"(r01), "+v"(r10), "+v"(r11));  // (experiment: the packed results have landed before any compare reads them)
#endif
#pragma unroll
      for (int dx = 0; dx < 2; dx++) {
        density_add(acc[dx * 4 + 0], r00[dx], t2, g2, kexp, ar, inv_ar, v.qa, v.qb, v.qc VOX_TRAP_VIOL_ARG);
        density_add(acc[dx * 4 + 1], r01[dx], t2, g2, kexp, ar, inv_ar, v.qa, v.qb, v.qc VOX_TRAP_VIOL_ARG);
        density_add(acc[dx * 4 + 2], r10[dx], t2, g2, kexp, ar, inv_ar, v.qa, v.qb, v.qc VOX_TRAP_VIOL_ARG);
        density_add(acc[dx * 4 + 3], r11[dx], t2, g2, kexp, ar, inv_ar, v.qa, v.qb, v.qc VOX_TRAP_VIOL_ARG);
      }
#endif
#if MI_VOX_FIX == 5
      if (mask) fetch(__builtin_ctzll(mask));  // (experiment: no scalar load in flight while the packed instructions run)
#endif
#ifdef MI_VOX_TRAP
      {
        const unsigned long long vm = __ballot(viol);
        if (vm) vox_trap(v.trap, 9u, blockIdx.x, (unsigned)tile_id, (unsigned)vm, (unsigned)(vm >> 32), (unsigned)c, (unsigned)cur_w, 0u, (unsigned)n);
      }
#endif
    }
  }
  flush(cur);

  if (MODE != 0)
    while (cur_w < nwin) emit_window();  // the last window, and windows no atom of this tile belongs to (zeros)
#ifdef MI_VOX_TRAP
  check_canaries(0x200u);
  if (MODE != 0 && cur_w != nwin) vox_trap(v.trap, 8u, blockIdx.x, (unsigned)tile_id, (unsigned)cur_w, (unsigned)nwin, (unsigned)cur, 0u, 0u, 0u);
#endif
  if constexpr (SPLIT)
    if (v.overflow && __ballot(ovf) != 0ull && lane == 0) atomicOr(v.overflow, 1u);
}


// ---------------------------------------------------------------------------------------------
// voxel_backward: GridMaker::backward (called torch_model.cpp:203; semantics SURVEY App. A.4) fused
// with the backward of the network's first 2x2x2 pooling layer.  One wavefront per (pose, typed
// ligand atom): lanes sweep the atom's bounding cube of fine voxels; the gradient of a fine voxel is
// recovered from the pooled-grid gradient (max: only the arg-max voxel of a cell; avg: 1/8 of it).
//   dL/dx_a = sum_v g[c][v] * rho'(d) * (x_a - p_v) / d
//   rho'(d) = -4 d / r^2 exp(-2 d^2 / r^2)  (d <= r);  (2 A d/r + B) / r  (r < d < 1.5 r);  0 otherwise
// ---------------------------------------------------------------------------------------------
template <int POOL>  // 1 max, 2 avg; 0 = no pooling: grad_pooled is a full-resolution gradient [B][C][N][N][N] (Cp = C)
__global__ __launch_bounds__(64) void voxel_backward_kernel(VoxBackArgs a) {
  const int b = blockIdx.y, j = blockIdx.x, lane = threadIdx.x;
  const int src = a.lig_perm[j];
  const LigConsts lc = a.lig_consts[j];
  const int c = a.lig_chan[j];
  float ax = a.lig_xyz[((size_t)b * a.L + src) * 3], ay = a.lig_xyz[((size_t)b * a.L + src) * 3 + 1],
        az = a.lig_xyz[((size_t)b * a.L + src) * 3 + 2];
  Rot3 R;
  if (a.rot) {  // the grid saw the rotated atom
    R = rot_of_quat(a.rot + 4 * (size_t)b);
    rot_about(R, a.centers[3 * b], a.centers[3 * b + 1], a.centers[3 * b + 2], ax, ay, az);
  }
  const float ox = a.centers[3 * b] - a.half_dim, oy = a.centers[3 * b + 1] - a.half_dim,
              oz = a.centers[3 * b + 2] - a.half_dim;
  const float maxr = lc.ar * 1.5f;
  const int N = a.N, S = N / 2, Cp = a.Cp;
  const int i0 = max(0, (int)floorf((ax - maxr - ox) / a.res)), i1 = min(2 * S - 1, (int)ceilf((ax + maxr - ox) / a.res));
  const int j0 = max(0, (int)floorf((ay - maxr - oy) / a.res)), j1 = min(2 * S - 1, (int)ceilf((ay + maxr - oy) / a.res));
  const int k0 = max(0, (int)floorf((az - maxr - oz) / a.res)), k1 = min(2 * S - 1, (int)ceilf((az + maxr - oz) / a.res));
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (i1 >= i0 && j1 >= j0 && k1 >= k0) {
    const int ni = i1 - i0 + 1, nj = j1 - j0 + 1, nk = k1 - k0 + 1;
    const int total = ni * nj * nk;
    const float *G = POOL == 0 ? a.grad_pooled + ((size_t)b * Cp + c) * N * N * N : a.grad_pooled + (size_t)b * S * S * S * Cp;
    const unsigned char *AM = POOL == 1 ? a.argmax + (size_t)b * S * S * S * Cp : nullptr;
    const float inv_ar2 = 1.0f / (lc.ar * lc.ar);
    // The gradient values (and arg-max bytes) of kAhead of this lane's voxels are requested before the first is used: in a
    // per-pose call one wave per atom has nothing to hide a round trip per voxel behind (27 of them for a 12^3 support:
    // this kernel was 26-30 us of a B = 1 gradient call).  Same terms in the same order per lane.
    constexpr int kAhead = 8;
    for (int t0 = lane; t0 < total; t0 += 64 * kAhead) {
      float gq[kAhead];
      unsigned char amq[kAhead];
#pragma unroll
      for (int u = 0; u < kAhead; u++) {
        const int t = t0 + 64 * u;
        gq[u] = 0.f, amq[u] = 0;
        if (t < total) {
          const int k = k0 + t % nk, jj = j0 + (t / nk) % nj, i = i0 + t / (nk * nj);
          const size_t cell = (((size_t)(i >> 1) * S + (jj >> 1)) * S + (k >> 1)) * Cp + c;
          if (POOL == 0) {
            gq[u] = G[((size_t)i * N + jj) * N + k];
          } else {
            gq[u] = G[cell];
            if (POOL == 1) amq[u] = AM[cell];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kAhead; u++) {
        const int t = t0 + 64 * u;
        if (t >= total) break;
        const int k = k0 + t % nk, jj = j0 + (t / nk) % nj, i = i0 + t / (nk * nj);
        float g = gq[u];
        if (POOL == 1) {
          const int r = ((i & 1) << 2) | ((jj & 1) << 1) | (k & 1);
          if (amq[u] != r) g = 0.f;
        } else if (POOL == 2) {
          g = g * 0.125f;
        }
        if (g == 0.f) continue;
        const float px = ox + (float)i * a.res, py = oy + (float)jj * a.res, pz = oz + (float)k * a.res;
        const float dx = ax - px, dy = ay - py, dz = az - pz;
        const float rsq = (dx * dx + dy * dy) + dz * dz;
        if (!(rsq < lc.t2) || rsq == 0.f) continue;
        const float dist = sqrtf(rsq);
        float d;
        if (rsq <= lc.g2)
          d = (-4.0f * dist * inv_ar2) * __expf(-2.0f * rsq * inv_ar2);
        else
          d = (2.0f * a.qa * (dist * lc.inv_ar) + a.qb) * lc.inv_ar;
        const float gv = g * d / dist;
        gx += gv * dx;
        gy += gv * dy;
        gz += gv * dz;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    gx += __shfl_xor(gx, off);
    gy += __shfl_xor(gy, off);
    gz += __shfl_xor(gz, off);
  }
  if (a.rot) {  // Transform::backward(grad, grad, false): the gradient vector goes back through R^T
    const float tx = R.m[0][0] * gx + R.m[1][0] * gy + R.m[2][0] * gz, ty = R.m[0][1] * gx + R.m[1][1] * gy + R.m[2][1] * gz,
                tz = R.m[0][2] * gx + R.m[1][2] * gy + R.m[2][2] * gz;
    gx = tx, gy = ty, gz = tz;
  }
  if (lane == 0) {
    float *o = a.lig_grad + ((size_t)b * a.L + src) * 3;
    if (a.accumulate) {
      o[0] += gx * a.scale;
      o[1] += gy * a.scale;
      o[2] += gz * a.scale;
    } else {
      o[0] = gx * a.scale;
      o[1] = gy * a.scale;
      o[2] = gz * a.scale;
    }
  }
}

void launch_voxel_backward(const VoxBackArgs &a, int B, int pool_mode, hipStream_t s) {
  if (a.n_lig == 0) return;
  dim3 grid(a.n_lig, B), block(64);
  if (pool_mode == 1)
    hipLaunchKernelGGL(voxel_backward_kernel<1>, grid, block, 0, s, a);
  else if (pool_mode == 2)
    hipLaunchKernelGGL(voxel_backward_kernel<2>, grid, block, 0, s, a);
  else
    hipLaunchKernelGGL(voxel_backward_kernel<0>, grid, block, 0, s, a);
}

// diagnostic (mi_debug_vox_stress): see voxelize.h
__global__ void dword_compare_kernel(const unsigned *got, const unsigned *want, size_t n, int iter, int *log, int cap) {
  bool any = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned a = got[i], b = want[i];
    if (a != b) {
      any = true;
      const int k = atomicAdd(log, 1);
      if (k < cap - 1) {
        int *e = log + 4 * (k + 1);
        e[0] = iter, e[1] = (int)i, e[2] = (int)a, e[3] = (int)b;
      }
    }
  }
  if (__syncthreads_or(any) && threadIdx.x == 0) atomicAdd(log + 2, 1);  // workgroups that saw a difference
}
void launch_dword_compare(const unsigned *got, const unsigned *want, size_t n, int iter, int *log, int cap, hipStream_t s) {
  hipLaunchKernelGGL(dword_compare_kernel, dim3(64), dim3(256), 0, s, got, want, n, iter, log, cap);
}

void launch_gather(const GatherArgs &g, int B, hipStream_t s) {
  if (B <= 64)
    hipLaunchKernelGGL(gather_pose_atoms<1024>, dim3(B), dim3(1024), 0, s, g);
  else
    hipLaunchKernelGGL(gather_pose_atoms<256>, dim3(B), dim3(256), 0, s, g);
}

void launch_voxelize(const VoxArgs &v_in, int B, int mode, hipStream_t s) {
  VoxArgs v = v_in;
  v.dbg = option(OPT_MI_VOX_DBG) ? atoi(option(OPT_MI_VOX_DBG)) : 0;
  const int nt = v.tiles_per_axis;
  dim3 grid(nt * nt * nt * B), block(64);
  if (mode == 0) {
    hipLaunchKernelGGL(voxelize_tiles<0>, grid, block, 0, s, v);
  } else {
    size_t lds = (size_t)64 * 12 * sizeof(float) + kVoxTrFloats * sizeof(float) + (v.argmax_out ? (size_t)64 * 12 : 0) + kVoxTrapPad * sizeof(float);  // kWin = 12
    if (option(OPT_MI_VOX_LDS_PAD)) lds += (size_t)atoi(option(OPT_MI_VOX_LDS_PAD)) * 1024;  // occupancy experiment
    if (v.split) {
      lds = (size_t)64 * kVoxSplitWin * sizeof(float) + kVoxTrFloats * sizeof(float) + 2 * kVoxTrapPad * sizeof(float);
      if (mode == 1) hipLaunchKernelGGL((voxelize_tiles<1, true>), grid, block, lds, s, v);
      else hipLaunchKernelGGL((voxelize_tiles<2, true>), grid, block, lds, s, v);
    } else if (mode == 1)
      hipLaunchKernelGGL(voxelize_tiles<1>, grid, block, lds, s, v);
    else
      hipLaunchKernelGGL(voxelize_tiles<2>, grid, block, lds, s, v);
  }
}

}  // namespace mig
