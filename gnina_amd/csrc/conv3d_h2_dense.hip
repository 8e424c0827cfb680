// conv3d_h2_dense.hip -- the Dense family's layers on split-format tensors (round 5).
//
// gnina's default CNN ensemble is two Dense models and one Default2018 (gninasrc/lib/cnn_torch_scorer.cpp:28-35), and a
// Dense model is, by FLOPs, its dense blocks: four 3x3x3 convolutions per block, each BatchNorm(concat so far) -> conv
// (c_in -> 16) -> ReLU -> append to the concat buffer (the TorchScript the reference runs at
// gninasrc/lib/torch_model.cpp:185).  Until round 4 those layers ran on conv3d_h2_16_kernel: fp32 concat buffer, BatchNorm
// and the fp16 split applied while staging (2.3 times per element, halo included), weights fetched by every wave from
// L1 / L2, A-operand reads at 0.48 bank conflicts each.  This file is the first convolution's recipe (LAB.md §3.9)
// applied to them:
//
//   * the concat buffer lives in HBM in the SPLIT FORMAT ([pose][octet][x][y][z][h0..h7 | l0..l7] fp16, conv3d.h
//     ConvArgs::in_split), written by its producers -- the block's own layers (the epilogue below), the model's first
//     convolution (conv3d_h2_kernel, out_split) and the 1x1x1 transition in front of the block (conv3d_h2_k1s_kernel);
//   * the eval BatchNorm is FOLDED: its scale into the packed weights (W'[t][c][n] = scale[c] W[t][c][n], split after the
//     fold), its shift into a bias per border class of the output voxel (ConvArgs::bias_tab: the zero padding is zero AFTER
//     the BatchNorm, so a tap outside the grid contributes no shift) -- the staged operand is the raw activation, staging
//     is a copy, and the copy is LDS-DMA (buffer_load ... lds) into a planar [h plane | l plane] halo tile;
//   * one K chunk = one octet = 27 taps = 7 steps of v_mfma_f32_16x16x32_f16 (lane group l >> 4 feeds one tap of the
//     step; the 28th tap has zero weights); the chunk's B operands (14 KB) go through LDS once per workgroup and, with
//     NP = 2, once per two poses;
//   * an M-tile is 4 x 2 x 2 voxels (row = y * 8 + z * 4 + x) and the taps are PAIRED (kD16TapOrder) so that every
//     ds_read_b128 lane group of the A operands hits sixteen different 16-byte slots modulo 16 with NO pad slots for the
//     tiles used (z-row stride 2 mod 4, x-plane stride 4 mod 8 slots: d16_layout_conflict_free checks the plan with the
//     bank model of MI355X_MICROARCH.md / LAB.md §3.1).
//
// Also here: conv3d_h2_k1s_kernel, the 1x1x1 transitions (96 -> 96 at 24^3, 160 -> 160 at 12^3, + ReLU + max pool) reading
// a split-format concat buffer by LDS-DMA and writing fp32 or split output.
//
// The gradient program keeps fp32 tensors and the round-3 kernels (BatchNorm while staging): a Dense pose scores within
// 2e-6 -- not bit for bit -- with and without its gradient (tests/test_gpu_gradient.py).
#include "common.h"
#include "conv3d.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>

namespace mig {

typedef float d16_f32x4 __attribute__((ext_vector_type(4)));
typedef float d16_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 d16_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 d16_f16x2 __attribute__((ext_vector_type(2)));
typedef float d16_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void *D16LdsPtr;

__device__ __forceinline__ unsigned d16_pk_f16(float a, float b) {
  const d16_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, d16_f16x2));
}
// one value -> h | l << 16 with h = RN_f16(x), l = RN_f16(x - h) (x - h is exact in fp32; v_fma_mix_f32 reads h as fp16);
// clamped to the fp16 range (the caller flags anything beyond it, and the call is repeated on the fp32 kernels)
__device__ __forceinline__ unsigned d16_split1(float x) {
  const float c = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
  const unsigned hp = d16_pk_f16(c, 0.f);
  float r;
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(c));
  return d16_pk_f16(c, r);
}
__device__ __forceinline__ void d16_report_overflow(unsigned *flag, bool lane_overflow) {
  if (flag && __builtin_amdgcn_ballot_w64(lane_overflow) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
// the dword a lane stores for channel `ch` (lanes of channels 2j and 2j + 1 are neighbours): the even lane the two h, the
// odd lane the two l -- dword ((ch & 7) >> 1) + 4 * (ch & 1) of the voxel's 32-byte octet record.  Every lane of the wave
// must call this (the DPP move reads the neighbour).
__device__ __forceinline__ unsigned d16_split_pair_dword(float v, int ch) {
  const unsigned mine = d16_split1(v);
  const unsigned other = (unsigned)__builtin_amdgcn_mov_dpp((int)mine, 0xb1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
  return (ch & 1) ? ((other >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (other << 16));
}

// ---------------------------------------------------------------------------------------------
// conv3d_h2_d16_kernel
// ---------------------------------------------------------------------------------------------
constexpr int kD16NS = 6;      // wave-DMAs per thread and chunk: 2 PL <= kD16NS * 256 slots
constexpr int kD16Steps = 7;   // 28 tap slots / 4 per instruction
constexpr int kD16WBytes = kD16Steps * 2048;  // a chunk's B operands: [step][h | l][lane group][16 couts][8 fp16]

// HONLY (MI_PRECISION_FP16, round 6): the h * h MFMA of every product only, and nothing of the l planes is DMA'd or read
// GRP (round 6, per-pose calls): a workgroup that is alone on its CU has nobody to hide its DMA round trips behind -- one
// exposed L2 round trip (~2 us) per K chunk, 28 of them in a Dense model's 24^3 block and 60 in its 12^3 block at B = 1.
// With GRP the chunks are requested ConvArgs::d16_group at a time into that many [tile | weights] sets (the CU's LDS is
// otherwise idle) and the K loops of the group run back to back: the same MFMAs in the same order, a round trip per group.
template <int TM, int NP, bool HONLY = false, bool GRP = false>
__global__ __launch_bounds__(256, 3) void conv3d_h2_d16_kernel(ConvArgs p) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane >> 4;   // which tap of a step this lane feeds (A rows / B columns: lane & 15)
  const int row = lane & 15;  // A row = voxel y * 8 + z * 4 + x of the M-tile's 4 x 2 x 2 block; B column = output channel

  const int tiles_per_pose = p.ntx * p.nty * p.ntz;
  const int HX = 2 * p.tcx + 2, HY = 2 * p.tcy + 2, HZ = 2 * p.tcz + 2;
  const int SY = HZ + p.h2_pad_y, SX = HY * SY + p.h2_pad_x;  // strides in 16-byte slots
  const int PL = (HX * SX + 31) & ~31;                        // slots per plane; the tile = 2 PL slots = whole wave-DMAs
  const int PLB = PL * 16;

  extern __shared__ __attribute__((aligned(16))) char smem_d16[];
  // set g of a group: [h plane | l plane | kD16WBytes of weights] at smem_d16 + g * GB (one set without GRP)
  const int GB = 2 * PLB + kD16WBytes;
  const int D = GRP ? p.d16_group : 1;

  // this lane's tap of every step: byte offset inside a plane (tap 27, the zero-weight filler, reads tap 0's slot)
  int qo[kD16Steps];
#pragma unroll
  for (int s = 0; s < kD16Steps; s++) {
    int tap = (int)((p.d16_taps[s] >> (8 * kg)) & 0xffu);
    if (tap > 26) tap = 0;
    const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    qo[s] = (dx * SX + dy * SY + dz) * 16;
  }

  // M-tile mt = (cxp, cy, cz): the cells (2 cxp, cy, cz) and (2 cxp + 1, cy, cz), i.e. voxels x = 4 cxp .. 4 cxp + 3
  const int n_mt = (p.tcx >> 1) * p.tcy * p.tcz;
  const int vy_r = row >> 3, vz_r = (row >> 2) & 1, vx_r = row & 3;
  int baseA[TM];
#pragma unroll
  for (int m = 0; m < TM; m++) {
    int mt = wave * TM + m;
    if (mt >= n_mt) mt = 0;  // (an idle M-tile slot reads valid LDS; its results are not stored)
    const int cz = mt % p.tcz, cy = (mt / p.tcz) % p.tcy, cxp = mt / (p.tcz * p.tcy);
    baseA[m] = ((4 * cxp + vx_r) * SX + (2 * cy + vy_r) * SY + (2 * cz + vz_r)) * 16;  // bytes inside a plane
  }

  // staging: slot j = tid + i * 256 of the tile is half j / PL of plane slot j % PL = halo voxel (hx, hy, hz); per lane and
  // slot that is the same for every item: hx | hy << 8 | hz << 16 | half << 24, all ones = a pad slot
  unsigned hpos[kD16NS];
  {
    const unsigned inv_sx = ((1u << 20) + SX - 1) / SX, inv_sy = ((1u << 20) + SY - 1) / SY;  // exact for n < 2^20 / d
#pragma unroll
    for (int i = 0; i < kD16NS; i++) {
      const int j = tid + i * 256;
      const int half = j >= PL ? 1 : 0;
      const int ps = j - half * PL;
      const int hx = (int)(((unsigned)ps * inv_sx) >> 20);
      const int r1 = ps - hx * SX;
      const int hy = (int)(((unsigned)r1 * inv_sy) >> 20), hz = r1 - hy * SY;
      hpos[i] = (j < 2 * PL && hx < HX && hy < HY && hz < HZ) ? ((unsigned)hx | (unsigned)hy << 8 | (unsigned)hz << 16 | (unsigned)half << 24) : 0xffffffffu;
    }
  }

  // PERSISTENT workgroups: the launch has at most (workgroups that fit the chip at once) of them and each walks the
  // (pose pair, tile) items item, item + gridDim.x, ... -- launching a 256-thread workgroup costs ~14 ns of dispatch on this
  // chip (27,648 of them that do nothing but their prologue and barriers take 0.37-0.42 ms: tools/experiments, LAB.md §3.10), a
  // fifth of a layer's time; the item's own set-up (tile coordinates, DMA offsets) is VALU work that overlaps with the other
  // resident workgroups' DMA phases.  gridDim.x is a multiple of 8 (or the item count), so item % 8 is this workgroup's XCD
  // for every item and xcd_contiguous_id keeps a pose's tiles on one XCD's L2.
  bool first = true, ovf = false;
  for (int item = blockIdx.x; item < p.n_items; item += gridDim.x) {
    const int wg = xcd_contiguous_id(item, p.n_items);
    const int b = (wg / tiles_per_pose) * NP;                // first pose of this workgroup
    const int npose = NP == 1 ? 1 : min(NP, p.nposes - b);   // (the last workgroups of an odd batch have one)
    int t = wg - (wg / tiles_per_pose) * tiles_per_pose;
    const int tz = t % p.ntz;
    t /= p.ntz;
    const int ty = t % p.nty, tx = t / p.nty;

    d16_f32x4 acc[NP][TM];
  #pragma unroll
    for (int tp = 0; tp < NP; tp++)
  #pragma unroll
      for (int m = 0; m < TM; m++) acc[tp][m] = {0.f, 0.f, 0.f, 0.f};

    const int S = p.S;
    const int x0 = tx * 2 * p.tcx - 1, y0 = ty * 2 * p.tcy - 1, z0 = tz * 2 * p.tcz - 1;
    const size_t pose_floats = (size_t)S * S * S * p.in_cs;
    const float *in_b = p.in + (size_t)b * pose_floats;

    // this item's DMA sources: voxels outside the grid and pad slots take an out-of-range offset, for which a buffer load
    // returns zeros (the zero padding)
    unsigned voff[kD16NS];
  #pragma unroll
    for (int i = 0; i < kD16NS; i++) {
      const int hx = (int)(hpos[i] & 0xffu), hy = (int)((hpos[i] >> 8) & 0xffu), hz = (int)((hpos[i] >> 16) & 0xffu);
      const int x = x0 + hx, y = y0 + hy, z = z0 + hz;
      const bool ok = hpos[i] != 0xffffffffu && (unsigned)x < (unsigned)S && (unsigned)y < (unsigned)S && (unsigned)z < (unsigned)S;
      voff[i] = ok ? (unsigned)((x * S + y) * S + z) * 32u + (hpos[i] >> 24) * 16u : 0x80000000u;
      if (p.h2_dbg & 32) voff[i] = (unsigned)(((x0 + 1) * S + (y0 + 1)) * S + z0 + 1) * 32u + (unsigned)(tid + i * 256) * 16u;  // (timing only: contiguous sources)
    }
    const int octet_bytes = S * S * S * 32;
    auto issue_tile = [&](int chunk, int tp, int g) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in_b + (size_t)tp * pose_floats), 0, (int)(pose_floats * 4), 0x00020000);
      char *dst = smem_d16 + g * GB + wave * 1024;
  #pragma unroll
      for (int i = 0; i < kD16NS; i++)
        // (MI_PRECISION_FP16: the wave-DMAs that lie in the l plane are not issued -- nothing reads it)
        if ((i * 4 + wave) * 64 < (HONLY ? PL : 2 * PL) && !(p.h2_dbg & 4))  // (wave-uniform; 2 PL is a multiple of 64)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (D16LdsPtr)(dst + i * 4096), 16, voff[i], chunk * octet_bytes, 0, 0);
    };
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wp), 0, p.nchunks * kD16WBytes, 0x00020000);
    auto issue_w = [&](int chunk, int g) {  // piece q = (step, h | l) = 1 KB = one wave-DMA, consecutive bytes
  #pragma unroll
      for (int i = 0; i < 4; i++) {
        const int q = i * 4 + wave;
        if (q < 2 * kD16Steps && !(p.h2_dbg & 8) && !(HONLY && (q & 1)))  // (odd pieces: the l halves)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (D16LdsPtr)(smem_d16 + g * GB + 2 * PLB + q * 1024), 16, (unsigned)lane * 16u, chunk * kD16WBytes + q * 1024, 0, 0);
      }
    };

    // (GRP: a workgroup alone on its CU) the epilogue's border-class bias, fetched in front of the K loops: as a load behind
    // them it was a round trip nothing hid
    float bias_pre[GRP ? TM : 1][4];
    if constexpr (GRP) {
      const int ch = row;
#pragma unroll
      for (int m = 0; m < TM; m++) {
        const int mt = wave * TM + m;
        const int cz = mt % p.tcz, cy = (mt / p.tcz) % p.tcy, cxp = mt / (p.tcz * p.tcy);
        const int gy = 2 * (ty * p.tcy + cy) + (kg >> 1), gz = 2 * (tz * p.tcz + cz) + (kg & 1);
        const int gx0 = 2 * tx * p.tcx + 4 * cxp;
        const int cls_yz = (gy == 0 ? 0 : gy == S - 1 ? 2 : 1) * 3 + (gz == 0 ? 0 : gz == S - 1 ? 2 : 1);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int gx = gx0 + r;
          const int cls = (gx == 0 ? 0 : gx >= S - 1 ? 2 : 1) * 9 + cls_yz;
          bias_pre[m][r] = p.bias_tab[min(cls, 26) * 16 + ch];
        }
      }
    }
    uint4 wh0, wl0, wh1, wl1, ah0[TM], al0[TM], ah1[TM], al1[TM];
    for (int chunk = 0; chunk < p.nchunks; chunk += D) {
      const int ng = GRP ? min(D, p.nchunks - chunk) : 1;  // chunks of this group
      bool w_here = false;  // this group's weights are in LDS
      auto pose_pass = [&](auto tpc) __attribute__((always_inline)) {
        constexpr int tp = decltype(tpc)::value;
        if (!first) __syncthreads();  // every wave is through the previous K loop: tile (and weights) may be overwritten
        first = false;
        for (int g = 0; g < ng; g++) {
          issue_tile(chunk + g, tp, g);
          if (!w_here) issue_w(chunk + g, g);
        }
        w_here = true;
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __syncthreads();
        for (int g = 0; g < ng; g++) {
        const char *const s_tile = smem_d16 + g * GB;
        const char *wl_ = s_tile + 2 * PLB + lane * 16;
        auto load_step = [&](int s, uint4 *ah, uint4 *al, uint4 &wh, uint4 &wl) __attribute__((always_inline)) {
  #pragma unroll
          for (int m = 0; m < TM; m++) {
            const char *a = s_tile + baseA[m] + qo[s];
            ah[m] = *reinterpret_cast<const uint4 *>(a);
            if constexpr (!HONLY) al[m] = *reinterpret_cast<const uint4 *>(a + PLB);
          }
          wh = *reinterpret_cast<const uint4 *>(wl_ + s * 2048);
          if constexpr (!HONLY) wl = *reinterpret_cast<const uint4 *>(wl_ + s * 2048 + 1024);
        };
        auto mfma_step = [&](const uint4 *ah, const uint4 *al, const uint4 &wh, const uint4 &wl) __attribute__((always_inline)) {
          // (three passes over the M-tiles: consecutive MFMAs never wait for each other's accumulator)
          if constexpr (!HONLY) {  // (MI_PRECISION_FP16: the h * h product only)
  #pragma unroll
            for (int m = 0; m < TM; m++)
              acc[tp][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(d16_f16x8, al[m]), __builtin_bit_cast(d16_f16x8, wh), acc[tp][m], 0, 0, 0);
  #pragma unroll
            for (int m = 0; m < TM; m++)
              acc[tp][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(d16_f16x8, ah[m]), __builtin_bit_cast(d16_f16x8, wl), acc[tp][m], 0, 0, 0);
          }
  #pragma unroll
          for (int m = 0; m < TM; m++)
            acc[tp][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(d16_f16x8, ah[m]), __builtin_bit_cast(d16_f16x8, wh), acc[tp][m], 0, 0, 0);
        };
        if (p.h2_dbg & 2) return;  // (timing only: no K loop)
        load_step(0, ah0, al0, wh0, wl0);
  #pragma unroll
        for (int s = 0; s < kD16Steps; s += 2) {
          if (s + 1 < kD16Steps) load_step(s + 1, ah1, al1, wh1, wl1);
          mfma_step(ah0, al0, wh0, wl0);
          if (s + 2 < kD16Steps) load_step(s + 2, ah0, al0, wh0, wl0);
          if (s + 1 < kD16Steps) mfma_step(ah1, al1, wh1, wl1);
        }
        }
      };
      pose_pass(std::integral_constant<int, 0>{});
      if constexpr (NP > 1)
        if (npose > 1) pose_pass(std::integral_constant<int, 1>{});
    }

    // ---- epilogue: accumulator row = 4 * (lane >> 4) + reg = y * 8 + z * 4 + x, column = lane & 15: a lane holds the four
    // x-consecutive voxels of one (y, z) for one output channel.  Un-scale, border-class bias, ReLU, split, store. ----
    const int ch = row;
    const int oct0 = (p.out_c0 >> 3) + (ch >> 3);
    const int ch_sp = ((ch & 7) >> 1) + (ch & 1) * 4;
    const size_t S3 = (size_t)S * S * S;
    auto finish_pose = [&](auto tpc) __attribute__((always_inline)) {
      constexpr int tp = decltype(tpc)::value;
      unsigned *out_u = reinterpret_cast<unsigned *>(p.out + (size_t)(b + tp) * S3 * p.out_cs);
  #pragma unroll
      for (int m = 0; m < TM; m++) {
        const int mt = wave * TM + m;
        const int cz = mt % p.tcz, cy = (mt / p.tcz) % p.tcy, cxp = mt / (p.tcz * p.tcy);
        const int gy = 2 * (ty * p.tcy + cy) + (kg >> 1), gz = 2 * (tz * p.tcz + cz) + (kg & 1);
        const int gx0 = 2 * tx * p.tcx + 4 * cxp;
        const bool ok_yz = mt < n_mt && gy < S && gz < S;
        const int cls_yz = (gy == 0 ? 0 : gy == S - 1 ? 2 : 1) * 3 + (gz == 0 ? 0 : gz == S - 1 ? 2 : 1);
  #pragma unroll
        for (int r = 0; r < 4; r++) {
          const int gx = gx0 + r;
          const int cls = (gx == 0 ? 0 : gx >= S - 1 ? 2 : 1) * 9 + cls_yz;
          float bias_v;
          if constexpr (GRP) bias_v = bias_pre[m][r];
          else bias_v = p.bias_tab[cls * 16 + ch];
          float v = acc[tp][m][r] * p.h2_unscale + bias_v;
          if (p.relu) v = fmaxf(v, 0.f);
          const bool ok = ok_yz && gx < S;
          ovf |= ok && !(fabsf(v) <= 65504.f);
          const unsigned w = d16_split_pair_dword(v, ch);
          if (ok && ch < p.cout) out_u[((size_t)oct0 * S3 + ((size_t)gx * S + gy) * S + gz) * 8 + ch_sp] = w;
        }
      }
    };
    if (!(p.h2_dbg & 64)) {  // (64: timing only, no epilogue)
      finish_pose(std::integral_constant<int, 0>{});
      if constexpr (NP > 1)
        if (npose > 1) finish_pose(std::integral_constant<int, 1>{});
    }
  }
  d16_report_overflow(p.h2_overflow, ovf);
}

// ---------------------------------------------------------------------------------------------
// conv3d_h2_k1s_kernel: a 1x1x1 convolution (+ ReLU, + 2x2x2 pool) whose input is a split-format tensor -- the Dense
// transitions behind a block (96 -> 96 at 24^3, 160 -> 160 at 12^3).  Same tile, K order, packed weights and MFMA sequence
// as conv3d_h2_k1_kernel (conv3d_h2.hip: 4 waves x 1 M-tile of 32 voxels x TN 32-channel groups, [voxel][octet][h | l]
// tile in LDS, one step = an octet pair), but the tile is laid down by LDS-DMA: slot j = voxel * (2 CC8 + 1) + 2 octet +
// half, 64 consecutive slots per wave-instruction, the odd slot count per voxel being the bank padding.  The kernel it
// replaces read 96 fp32 channels per voxel into registers, split them and wrote LDS: 2.66 ms for 5.4 GB at 24^3.
// ---------------------------------------------------------------------------------------------
constexpr int kK1sNS = 7;  // wave-DMAs per thread and chunk: voxels * (2 CC8 + 1) <= kK1sNS * 256 slots
// bytes of one chunk's tile set in LDS (`total` 16-byte slots, whole wave-DMAs, + the finite pad behind the last voxel)
__host__ __device__ inline size_t conv_h2_k1s_set_bytes(int total) { return ((((size_t)total + 63) & ~(size_t)63) * 16 + 64 + 1023) & ~(size_t)1023; }

// GRP (round 6, per-pose calls; see conv3d_h2_d16_kernel): the K chunks' tiles are DMA'd ConvArgs::d16_group at a time into as
// many LDS sets -- one round trip per group instead of one per chunk for a workgroup that is alone on its CU
template <int TN, bool HONLY = false, bool GRP = false>
__global__ __launch_bounds__(256, (TN <= 3 ? 4 : 3)) void conv3d_h2_k1s_kernel(ConvArgs p) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = lane >> 5;
  const int row = lane & 31;
  const int tiles_per_pose = p.ntx * p.nty * p.ntz;
  constexpr int n_base = 0;  // (all output channels in one workgroup: the input is read once)

  const int HX = 2 * p.tcx, HY = 2 * p.tcy, HZ = 2 * p.tcz;
  const int HV = HX * HY * HZ;
  const int CC8 = p.cc4;                // octets per K chunk
  const int NSV = 2 * CC8 + 1;          // 16-byte slots per voxel
  const int total = HV * NSV;
  extern __shared__ __attribute__((aligned(16))) char smem_k1s[];
  char *const s_tile = smem_k1s;
  // (an odd CC8 leaves the second half-wave of a chunk's last step on the voxel's pad slot and the 16 bytes behind it --
  // the next voxel's first slot, or, behind the last voxel, this one: zero weight rows cancel them, but they must be finite)
  const int D = GRP ? p.d16_group : 1;
  const int set_bytes = (int)conv_h2_k1s_set_bytes(total);
  if (tid < 4)
    for (int gi = 0; gi < D; gi++) reinterpret_cast<unsigned *>(s_tile + (size_t)gi * set_bytes + (size_t)((total + 63) & ~63) * 16)[tid] = 0u;

  const int NC = p.tcx * p.tcy * p.tcz;
  const int oz = row & 1, oy = (row >> 1) & 1, ox = (row >> 3) & 1;
  const int cell_in_mt = ((row >> 2) & 1) + 2 * ((row >> 4) & 1);
  auto cell_of = [&](int mt, int cim, int &cx, int &cy, int &cz) -> bool {
    const int cell = mt * 4 + cim;
    cz = cell % p.tcz, cy = (cell / p.tcz) % p.tcy, cx = cell / (p.tcz * p.tcy);
    return cell < NC;
  };
  int baseA;
  {
    int cx, cy, cz;
    if (!cell_of(wave, cell_in_mt, cx, cy, cz)) cx = cy = cz = 0;
    baseA = (((2 * cx + ox) * HY + (2 * cy + oy)) * HZ + (2 * cz + oz)) * NSV * 16;  // bytes
  }
  // bias of this lane's channels, fetched once (the epilogue of a latency-bound workgroup must not start with a round trip)
  float bias_r[TN];
#pragma unroll
  for (int n = 0; n < TN; n++) bias_r[n] = p.bias[n_base + n * 32 + row < p.coutp ? n_base + n * 32 + row : 0];

  const int S = p.S;
  const size_t S3 = (size_t)S * S * S;
  const size_t pose_floats = S3 * p.in_cs;
  const int chunk_bytes = CC8 * (int)S3 * 32;
  const size_t wstride = (size_t)p.coutp * 16;  // fp16 elements per octet row of the packed weights
  const int P = (CC8 + 1) >> 1;                 // steps (octet pairs) per chunk
  const unsigned wlane = ((unsigned)(n_base + row) * 16u + (unsigned)kh * (unsigned)wstride) * 2u;
  const unsigned wstep = 2u * (unsigned)wstride * 2u;  // bytes per octet pair
  const int nsteps = p.nchunks * P;                    // steps of the whole K loop: step g = chunk g / P, pair g % P
  uint4 wh[TN], wl[TN];
  // B operands of step g, straight from L1 / L2 (36 KB of weights per layer, every wave wants all of them) -- ONE register
  // set: the loads of step g + 1 are issued right behind the MFMAs of step g (which read their operands as they issue) and
  // are in flight while those drain and, across a chunk boundary, through the barrier and the DMA wait
  auto load_w = [&](int g) __attribute__((always_inline)) {
    const char *w0 = reinterpret_cast<const char *>(p.wp) + (size_t)g * wstep;
#pragma unroll
    for (int n = 0; n < TN; n++) {
      const char *w = w0 + (wlane + (unsigned)n * 32u * 32u);
      wh[n] = *reinterpret_cast<const uint4 *>(w);
      if constexpr (!HONLY) wl[n] = *reinterpret_cast<const uint4 *>(w + 16);
    }
  };

  // PERSISTENT workgroups (see conv3d_h2_d16_kernel): items item, item + gridDim.x, ...
  bool ovf = false, first = true;
  for (int item = blockIdx.x; item < p.n_items; item += gridDim.x) {
    const int wg = xcd_contiguous_id(item, p.n_items);
    const int b = wg / tiles_per_pose;
    int t = wg - b * tiles_per_pose;
    const int tz = t % p.ntz;
    t /= p.ntz;
    const int ty = t % p.nty, tx = t / p.nty;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in + (size_t)b * pose_floats), 0, (int)(pose_floats * 4), 0x00020000);
    // staging: slot j = tid + i * 256 = voxel j / NSV, piece j % NSV (2 octet + half; the last slot of a voxel is padding);
    // a lane keeps the byte offset of its slots' sources relative to the chunk's first octet array (recomputed per item: a
    // hundred VALU instructions against tens of microseconds of DMA, and seven registers less across the K loop)
    unsigned voff[kK1sNS];
    {
      // (from an OPAQUE copy of the thread index: the slot -> (voxel, piece) decomposition is tile-invariant, and hoisted out
      // of the item loop it was 30 spilled dwords -- stored once, reloaded for every item)
      int tid_i = tid;
      asm volatile("" : "+v"(tid_i));
      const unsigned inv_nsv = ((1u << 20) + NSV - 1) / NSV;  // exact for j < 2^20 / NSV
      const unsigned inv_hz = ((1u << 20) + HZ - 1) / HZ, inv_hy = ((1u << 20) + HY - 1) / HY;
#pragma unroll
      for (int i = 0; i < kK1sNS; i++) {
        const int j = tid_i + i * 256;
        const int v = (int)(((unsigned)j * inv_nsv) >> 20), pp = j - v * NSV;
        const int t1 = (int)(((unsigned)v * inv_hz) >> 20), hz = v - t1 * HZ;
        const int hx = (int)(((unsigned)t1 * inv_hy) >> 20), hy = t1 - hx * HY;
        const int x = tx * HX + hx, y = ty * HY + hy, z = tz * HZ + hz;
        const bool ok = j < total && pp < 2 * CC8 && x < S && y < S && z < S && !(HONLY && (pp & 1));  // (MI_PRECISION_FP16: no l halves)
        voff[i] = ok ? (unsigned)((size_t)(pp >> 1) * S3 + (size_t)((x * S + y) * S + z)) * 32u + (unsigned)(pp & 1) * 16u : 0x80000000u;
      }
    }
    d16_f32x16 acc[TN];
#pragma unroll
    for (int n = 0; n < TN; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[n][r] = 0.f;

    load_w(0);
    int g = 0;
    for (int chunk0 = 0; chunk0 < p.nchunks; chunk0 += D) {
      const int ng = GRP ? min(D, p.nchunks - chunk0) : 1;  // chunks of this group
      if (!first) __syncthreads();  // every wave is through the previous K loop: the tile may be overwritten
      first = false;
      for (int gi = 0; gi < ng; gi++) {
#pragma unroll
        for (int i = 0; i < kK1sNS; i++)
          if ((i * 4 + wave) * 64 < total && !(p.h2_dbg & 4))  // (wave-uniform)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (D16LdsPtr)(s_tile + gi * set_bytes + (i * 4 + wave) * 1024), 16, voff[i], (chunk0 + gi) * chunk_bytes, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      __syncthreads();
      if (p.h2_dbg & 2) {  // (timing only: no K loop)
        g += P * ng;
        continue;
      }
      for (int gi = 0; gi < ng; gi++)
      for (int pr = 0; pr < P; pr++, g++) {
        const char *a = s_tile + gi * set_bytes + baseA + (2 * pr + kh) * 32;
        const uint4 ah = *reinterpret_cast<const uint4 *>(a);
        uint4 al = ah;
        if constexpr (!HONLY) al = *reinterpret_cast<const uint4 *>(a + 16);
        // (three passes over the channel groups: consecutive MFMAs never wait for each other's accumulator; per accumulator
        // the order al*wh, ah*wl, ah*wh of conv3d_h2_k1_kernel is kept -- same bits)
        if constexpr (!HONLY) {  // (MI_PRECISION_FP16: the h * h product only)
#pragma unroll
          for (int n = 0; n < TN; n++)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(d16_f16x8, al), __builtin_bit_cast(d16_f16x8, wh[n]), acc[n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < TN; n++)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(d16_f16x8, ah), __builtin_bit_cast(d16_f16x8, wl[n]), acc[n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < TN; n++)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(d16_f16x8, ah), __builtin_bit_cast(d16_f16x8, wh[n]), acc[n], 0, 0, 0);
        if (g + 1 < nsteps) load_w(g + 1);
      }
    }

    // ---- epilogue (accumulator layout and cell order of conv3d_h2_k1_kernel): un-scale, bias, ReLU, optional 2x2x2 pool;
    // channels-last fp32, or the split format (the lanes of channels 2j / 2j + 1 trade halves: d16_split_pair_dword) ----
    if (p.h2_dbg & 64) continue;  // (timing only: no epilogue)
    // (the epilogue's lane-derived values from an opaque copy of the thread index, for the same reason as voff above: the
    // channel offsets of the TN groups are tile-invariant, and hoisted out of the item loop they are spilled)
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int row = tid_e & 31, kh = (tid_e >> 5) & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid_e >> 6);
    const int So = p.pool ? S / 2 : S;
    const size_t So3 = (size_t)So * So * So;
    float *out_f = p.out + (size_t)b * So3 * p.out_cs + p.out_c0;
    unsigned *out_u = reinterpret_cast<unsigned *>(p.out + (size_t)b * So3 * p.out_cs);
    const int ncx = S / 2;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      int cx, cy, cz;
      const bool in_tile = cell_of(wave, kh + 2 * half, cx, cy, cz);
      const int gcx = tx * p.tcx + cx, gcy = ty * p.tcy + cy, gcz = tz * p.tcz + cz;
      const bool okc = in_tile && gcx < ncx && gcy < ncx && gcz < ncx;
#pragma unroll
      for (int n = 0; n < TN; n++) {
        const int ch = n_base + n * 32 + row;
        const bool okn = okc && ch < p.cout;
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const float tt = acc[n][half * 8 + r] * p.h2_unscale + bias_r[n];
          v[r] = p.relu ? fmaxf(tt, 0.f) : tt;
        }
        auto store = [&](size_t vox, float val) __attribute__((always_inline)) {
          if (p.out_split) {
            ovf |= okn && !(fabsf(val) <= 65504.f);
            const unsigned w = d16_split_pair_dword(val, ch);
            if (okn) out_u[((size_t)((p.out_c0 + ch) >> 3) * So3 + vox) * 8 + ((ch & 7) >> 1) + (ch & 1) * 4] = w;
          } else if (okn) {
            out_f[vox * p.out_cs + ch] = val;
          }
        };
        if (p.pool == 1) {
          // max of the cell's eight voxels (no arg-max in the forward program)
          const float m1 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), m2 = fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]));
          store(((size_t)gcx * So + gcy) * So + gcz, fmaxf(m1, m2));
        } else if (p.pool == 2) {
          float sum = v[0];
#pragma unroll
          for (int r = 1; r < 8; r++) sum = sum + v[r];
          store(((size_t)gcx * So + gcy) * So + gcz, sum * 0.125f);
        } else {
#pragma unroll
          for (int r = 0; r < 8; r++) {
            const int vx = 2 * gcx + (r >> 2), vy = 2 * gcy + ((r >> 1) & 1), vz = 2 * gcz + (r & 1);
            store(((size_t)vx * So + vy) * So + vz, v[r]);
          }
        }
      }
    }
  }
  d16_report_overflow(p.h2_overflow, ovf);
}

size_t conv_h2_k1s_lds_bytes(const ConvArgs &p) {
  const size_t total = (size_t)8 * p.tcx * p.tcy * p.tcz * (2 * p.cc4 + 1);
  return conv_h2_k1s_set_bytes((int)total);
}

// persistent launches: workgroups the chip holds at once for this kernel (occupancy API, cached), a multiple of 8
static int d16_resident_workgroups(const void *kern, size_t lds, int per_cu_cap) {
  static std::mutex mu;
  static std::map<std::tuple<int, const void *, size_t>, int> cache;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_tuple(dev, kern, lds);
  auto it = cache.find(key);
  if (it == cache.end()) {
    int cus = 256, occ = 1;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds) != hipSuccess || occ < 1) occ = 1;
    it = cache.emplace(key, std::max(cus, 8) * 1000 + occ).first;
  }
  const int cus = it->second / 1000, occ = std::min(it->second % 1000, per_cu_cap > 0 ? per_cu_cap : 1000);
  return (cus * occ) & ~7;
}

void launch_conv_h2_k1s(ConvArgs p, int B, hipStream_t s) {
  const int tn = p.coutp / 32;
  const size_t total = (size_t)8 * p.tcx * p.tcy * p.tcz * (2 * p.cc4 + 1);
  if (p.ksize != 1 || !p.in_split || p.in_cs % 8 || p.tcx * p.tcy * p.tcz > 16 || total > (size_t)kK1sNS * 256 || p.bn_scale || p.post_w ||
      (p.in_cs / 8) != p.cc4 * p.nchunks || (p.out_split && ((p.out_c0 % 8) || (p.out_cs % 8) || (p.cout % 8))) || (tn != 3 && tn != 5) ||
      2 * p.tcx > 255 || 2 * p.tcy > 255 || 2 * p.tcz > 255)
    throw Error(2, "launch_conv_h2_k1s: launch outside what the kernel covers");
  size_t lds = conv_h2_k1s_lds_bytes(p);
  p.n_items = B * p.ntx * p.nty * p.ntz;
  // few enough workgroups that each has (a share of) a CU's LDS to itself: the chunk groups of the GRP variants
  const long cap = p.d16_group > 0 ? std::min(6L, (long)p.d16_group) : 6L;
  p.d16_group = 1;
  if (!p.h2_honly && !(p.h2_dbg & 128)) {
    const long per_cu = ((long)p.n_items + 255) / 256;
    const long fit = (long)(160 * 1024) / ((long)lds * per_cu);
    p.d16_group = (int)std::max(1L, std::min({cap, (long)p.nchunks, fit}));
    lds *= (size_t)p.d16_group;
  }
  auto go = [&](auto kern) {
    ensure_max_lds(reinterpret_cast<const void *>(kern), 160 * 1024);
    int grid = p.n_items;
    if (p.h2_persist != 0) grid = std::min(grid, d16_resident_workgroups(reinterpret_cast<const void *>(kern), lds, p.h2_persist));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, p);
  };
  if (p.h2_honly) {
    if (tn == 3) go(conv3d_h2_k1s_kernel<3, true>);
    else go(conv3d_h2_k1s_kernel<5, true>);
    return;
  }
  if (p.d16_group > 1) {
    if (tn == 3) go(conv3d_h2_k1s_kernel<3, false, true>);
    else go(conv3d_h2_k1s_kernel<5, false, true>);
    return;
  }
  if (tn == 3) go(conv3d_h2_k1s_kernel<3>);
  else go(conv3d_h2_k1s_kernel<5>);
}

// the tap fed by lane group g of step s is kD16TapOrder[4 s + g] (27 = the zero-weight filler): pairs (g = 0, 1) and
// (g = 2, 3) chosen by a minimum-weight matching over the bank model so that the A-operand reads are conflict-free
const unsigned char kD16TapOrder[28] = {0, 27, 1, 25, 2, 26, 3, 21, 4, 22, 5, 24, 6, 23, 7, 19, 8, 20, 9, 18, 10, 16, 11, 17, 12, 13, 14, 15};

void conv_d16_tap_order(unsigned taps4[7], unsigned char order[28]) {
  for (int s = 0; s < 7; s++) {
    taps4[s] = 0;
    for (int g = 0; g < 4; g++) taps4[s] |= (unsigned)kD16TapOrder[4 * s + g] << (8 * g);
  }
  if (order) std::copy(kD16TapOrder, kD16TapOrder + 28, order);
}

// Bank model of the kernel's A-operand reads for a halo tile with z-row stride SY and x-plane stride SX (16-byte slots): a
// ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32), a group
// in as many LDS cycles as the largest number of its lanes that hit one slot class modulo 16 at different addresses.
bool conv_d16_layout_conflict_free(int SY, int SX) {
  static const int kGroup[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
  for (int pr = 0; pr < 14; pr++) {
    int off[2];
    for (int g = 0; g < 2; g++) {
      int tap = kD16TapOrder[2 * pr + g];
      if (tap > 26) tap = 0;
      off[g] = (tap / 9) * SX + ((tap / 3) % 3) * SY + tap % 3;
    }
    for (int gi = 0; gi < 2; gi++) {
      int slot[16];
      for (int i = 0; i < 16; i++) {
        const int l = kGroup[gi][i], rw = l & 15;
        slot[i] = (rw & 3) * SX + (rw >> 3) * SY + ((rw >> 2) & 1) + off[l >> 4];
      }
      for (int i = 0; i < 16; i++)
        for (int j = 0; j < i; j++)
          if (slot[i] != slot[j] && (slot[i] - slot[j]) % 16 == 0) return false;
    }
  }
  return true;
}

size_t conv_h2_d16_lds_bytes(const ConvArgs &p) {
  const int HX = 2 * p.tcx + 2, HY = 2 * p.tcy + 2, HZ = 2 * p.tcz + 2;
  const int SY = HZ + p.h2_pad_y, SX = HY * SY + p.h2_pad_x;
  const int PL = (HX * SX + 31) & ~31;
  return (size_t)2 * PL * 16 + kD16WBytes;
}

void launch_conv_h2_d16(ConvArgs p, int B, hipStream_t s) {
  p.nposes = B;
  const int HX = 2 * p.tcx + 2, HY = 2 * p.tcy + 2, HZ = 2 * p.tcz + 2;
  const int SY = HZ + p.h2_pad_y, SX = HY * SY + p.h2_pad_x;
  const int PL = (HX * SX + 31) & ~31;
  const int n_mt = (p.tcx / 2) * p.tcy * p.tcz;
  if (p.tcx % 2 || 2 * PL > kD16NS * 256 || n_mt > 16 || p.cout > 16 || p.ksize != 3 || !p.in_split || !p.out_split || p.in_cs % 8 || p.out_cs % 8 ||
      p.out_c0 % 8 || !p.bias_tab || p.pool || p.bn_scale)
    throw Error(2, "launch_conv_h2_d16: launch outside what the kernel covers");
  size_t lds = conv_h2_d16_lds_bytes(p);
  const int tiles = p.ntx * p.nty * p.ntz;
  const bool two = p.h2_wlds >= 2 && B >= 2;
  // one pose per workgroup and few enough workgroups that each has (a share of) a CU's LDS to itself: the GRP variants, as
  // many chunk sets as that share holds (<= 6: a wave has up to 10 DMAs per chunk in flight and vmcnt counts to 63)
  if (two || p.h2_honly || (p.h2_dbg & 128)) p.d16_group = 1;
  else {
    const long per_cu = ((long)B * tiles + 255) / 256;
    const long fit = (long)(160 * 1024) / ((long)lds * per_cu);
    const long cap = p.d16_group > 0 ? std::min(6L, (long)p.d16_group) : 6L;  // (the caller's cap, MI_GNINA_D16_GROUP_MAX; 0 = none)
    p.d16_group = (int)std::max(1L, std::min({cap, (long)p.nchunks, fit}));
    lds *= (size_t)p.d16_group;
  }
  const bool grp = p.d16_group > 1;
  auto go = [&](auto kern, int np) {
    ensure_max_lds(reinterpret_cast<const void *>(kern), 160 * 1024);
    p.n_items = (B + np - 1) / np * tiles;
    // persistent workgroups: as many as the chip holds at once (occupancy API), a multiple of 8 so that a workgroup stays on
    // one XCD's items; h2_persist = 0: one workgroup per item, > 0: a cap on the workgroups per CU, < 0: no cap
    int grid = p.n_items;
    if (p.h2_persist != 0) grid = std::min(grid, d16_resident_workgroups(reinterpret_cast<const void *>(kern), lds, p.h2_persist));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, p);
  };
  if (p.h2_honly) {  // (MI_PRECISION_FP16: the two-pose variants -- a throughput mode)
    if (n_mt <= 4) two ? go(conv3d_h2_d16_kernel<1, 2, true>, 2) : go(conv3d_h2_d16_kernel<1, 1, true>, 1);
    else if (n_mt <= 8) two ? go(conv3d_h2_d16_kernel<2, 2, true>, 2) : go(conv3d_h2_d16_kernel<2, 1, true>, 1);
    else if (n_mt <= 12) two ? go(conv3d_h2_d16_kernel<3, 2, true>, 2) : go(conv3d_h2_d16_kernel<3, 1, true>, 1);
    else two ? go(conv3d_h2_d16_kernel<4, 2, true>, 2) : go(conv3d_h2_d16_kernel<4, 1, true>, 1);
    return;
  }
  if (grp) {
    if (n_mt <= 4) go(conv3d_h2_d16_kernel<1, 1, false, true>, 1);
    else if (n_mt <= 8) go(conv3d_h2_d16_kernel<2, 1, false, true>, 1);
    else if (n_mt <= 12) go(conv3d_h2_d16_kernel<3, 1, false, true>, 1);
    else go(conv3d_h2_d16_kernel<4, 1, false, true>, 1);
    return;
  }
  if (n_mt <= 4) two ? go(conv3d_h2_d16_kernel<1, 2>, 2) : go(conv3d_h2_d16_kernel<1, 1>, 1);
  else if (n_mt <= 8) two ? go(conv3d_h2_d16_kernel<2, 2>, 2) : go(conv3d_h2_d16_kernel<2, 1>, 1);
  else if (n_mt <= 12) two ? go(conv3d_h2_d16_kernel<3, 2>, 2) : go(conv3d_h2_d16_kernel<3, 1>, 1);
  else two ? go(conv3d_h2_d16_kernel<4, 2>, 2) : go(conv3d_h2_d16_kernel<4, 1>, 1);
}

}  // namespace mig
