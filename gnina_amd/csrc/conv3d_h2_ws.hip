// conv3d_h2_ws.hip -- the first 3x3x3 convolution with STATIONARY WEIGHTS and a RING of halo tiles (round 6).
//
// The layer: Default2017's 35 -> 32 at 24^3 (+ ReLU + max pool), half of the headline step, and the Dense family's
// un-pooled 28 -> 32 (gninasrc/lib/torch_model.cpp:185 runs them inside module.forward).  conv3d_h2_kernel (conv3d_h2.hip)
// runs it as one workgroup per (pose pair, tile): every (K chunk, pose) phase of a workgroup DMAs its 19 KB halo tile AND
// re-fetches the chunk's 28 KB of weights, waits out the round trip (3.4 us under load) and then computes for 0.7-1.1 us;
// three such workgroups per CU (47 KB of LDS each) overlap each other, and the MFMA pipe sits at 0.37 (rounds 4 and 5:
// DESIGN / LAB).  What bounds it is the number of phases in flight per CU -- LDS bytes per phase -- and 60 % of a phase's
// bytes are weights every workgroup of the CU holds a copy of.
//
// Here ONE persistent workgroup per CU (4 waves, one per SIMD, 512 registers each) walks batches of J poses x one tile:
//   * loop order chunk-outer, pose-inner: a chunk's weights are DMA'd into LDS once per J poses (two buffers: the next live
//     chunk's weights arrive while this chunk's poses are computed) -- 1/J-th of the weight traffic, and they leave the
//     per-phase LDS budget;
//   * the accumulators of all J poses stay in registers across the chunks (J x 2 M-tiles x 16 = 256 accumulation registers);
//   * halo tiles go through a ring of R buffers: the tile of round r + R - 1 is requested before the K loop of round r, so
//     R - 1 round trips are in flight under every K loop; s_waitcnt vmcnt(N) with N = the DMA instructions of the YOUNGER
//     tiles lets a wave go on as soon as ITS part of the oldest tile has landed, one barrier per round makes that true for
//     the workgroup;
//   * which (chunk, pose) rounds exist at all is known before anything is staged: the voxelizer's occupancy bytes
//     (ConvArgs::in_occ) -- the round list of a batch is built once, in LDS.
// Tile geometry, LDS image of a tile, packed weights, K order, MFMA order per accumulator and epilogue are conv3d_h2_kernel's
// (<4,1,2,MT = 1,...>): a pose scores the same bits on either kernel (tests/test_gpu_h2.py).
#include "common.h"
#include "conv3d.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>

namespace mig {

typedef float ws_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 ws_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ws_f16x2 __attribute__((ext_vector_type(2)));
typedef float ws_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void *WsLdsPtr;

__device__ __forceinline__ unsigned ws_pk_f16(float a, float b) {
  const ws_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ws_f16x2));
}
// one value -> h | l << 16 (conv3d_h2.hip split1)
__device__ __forceinline__ unsigned ws_split1(float x) {
  const float c = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
  const unsigned hp = ws_pk_f16(c, 0.f);
  float r;
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(c));
  return ws_pk_f16(c, r);
}

constexpr int kWsNT = 5;    // wave-DMAs per wave and tile: a tile buffer = kWsNT * 256 slots of 16 bytes >= 2 PL
// (weights: 28 pieces of 1 KB per chunk, seven wave-DMAs per wave: issue_w spells them out)
constexpr int kWsP = 14;    // steps per chunk: 27 taps in pairs
constexpr int kWsTileB = kWsNT * 256 * 16, kWsWB = kWsP * 2048;

// s_waitcnt vmcnt(n), n a multiple of kWsNT (wave-uniform n: a scalar branch to one of the immediates)
__device__ __forceinline__ void ws_wait_vm(int tiles_younger) {
  switch (tiles_younger) {
    case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;                 // vmcnt(0)
    case 1: __builtin_amdgcn_s_waitcnt(0x0F70 | kWsNT); break;         // vmcnt(5)
    case 2: __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * kWsNT)); break;   // vmcnt(10)
    case 3: __builtin_amdgcn_s_waitcnt(0x0F70 | (3 * kWsNT)); break;   // vmcnt(15)
    default: __builtin_amdgcn_s_waitcnt(0x4F70 | ((4 * kWsNT) & 15)); break;  // vmcnt(20): bits 15:14 = vmcnt[5:4]
  }
}

template <int J, int R>
__global__ __launch_bounds__(256, 1) void conv3d_h2_ws_kernel(ConvArgs p) {
  static_assert(R >= 2 && R <= 5 && J >= 1 && J <= 8, "ring / batch sizes the wait immediates and the round list cover");
  constexpr int TM = 2, NW = 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = lane >> 5, row = lane & 31;

  const int tiles_per_pose = p.ntx * p.nty * p.ntz;
  const int HX = 2 * p.tcx + 2, HY = 2 * p.tcy + 2, HZ = 2 * p.tcz + 2;
  const int SY = HZ + p.h2_pad_y, SX = HY * SY + p.h2_pad_x;  // strides in 16-byte slots
  const int PL = (HX * SX + 31) & ~31;                        // slots per plane (2 PL <= kWsNT * 256: launcher)
  const int PLB = PL * 16;

  extern __shared__ __attribute__((aligned(16))) char smem_ws[];
  char *const s_ring = smem_ws;                                  // [R][kWsTileB]: [h plane | l plane | unused]
  char *const s_w = smem_ws + R * kWsTileB;                      // [2][kWsWB]
  int *const s_qoff = reinterpret_cast<int *>(s_w + 2 * kWsWB);  // [32] byte offset of tap q inside a plane
  int *const s_occ = s_qoff + 32;                                // [8] live-octet mask per pose of the batch
  int *const s_nr = s_occ + 8;                                   // [4] number of rounds (+ pad)
  unsigned char *const s_round = reinterpret_cast<unsigned char *>(s_nr + 4);  // [64] chunk | pose << 3 | first-of-chunk << 6
  unsigned char *const s_nextc = s_round + 64;                   // [64] first-of-chunk rounds: the NEXT live chunk, 0xff = none
  if (tid < 32) {
    const int tap = tid < 27 ? tid : 26;
    const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    s_qoff[tid] = (dx * SX + dy * SY + dz) * 16;
  }

  // M-tile geometry 1 (ConvArgs::mt_x): the four cells of an M-tile are stacked along x
  const int oz = row & 1, oy = (row >> 1) & 1, ox = (row >> 3) & 1;
  const int cell_in_mt = ((row >> 2) & 1) + 2 * ((row >> 4) & 1);
  auto cell_of = [&](int mt, int cim, int &cx, int &cy, int &cz) __attribute__((always_inline)) -> bool {
    cz = mt % p.tcz;
    cy = (mt / p.tcz) % p.tcy;
    cx = 4 * (mt / (p.tcz * p.tcy)) + cim;
    return cx < p.tcx;
  };
  int baseA[TM];
#pragma unroll
  for (int m = 0; m < TM; m++) {
    int cx, cy, cz;
    if (!cell_of(wave * TM + m, cell_in_mt, cx, cy, cz)) cx = cy = cz = 0;
    baseA[m] = ((2 * cx + ox) * SX + (2 * cy + oy) * SY + (2 * cz + oz)) * 16;  // bytes inside a plane
  }

  const int S = p.S;
  const size_t pose_floats = (size_t)S * S * S * p.in_cs;
  const int octet_bytes = S * S * S * 32;
  // packed weights [chunk][step][h | l][half-wave][cout][8 fp16]
  const unsigned wlane = ((unsigned)kh * (unsigned)p.coutp + (unsigned)row) * 16u;
  const unsigned wl_off = 2u * (unsigned)p.coutp * 16u;
  const unsigned wstep = 2u * wl_off;
  // The DMAs are inline asm: the compiler must not know that they write LDS -- it would hold every ds_read of the K loop back
  // until the tiles requested ahead of it have landed (it cannot tell the ring's buffers apart), which is the opposite of what
  // the ring is for.  Completion is counted by hand (ws_wait_vm); the compiler's own waits can only be stricter.
  typedef int ws_i32x4 __attribute__((ext_vector_type(4)));
  auto make_rsrc = [&](const void *base, unsigned bytes) __attribute__((always_inline)) {
    const unsigned long long a = (unsigned long long)base;
    ws_i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull)), rs.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffull));
    rs.z = __builtin_amdgcn_readfirstlane((int)bytes), rs.w = 0x00020000;
    return rs;
  };
  const ws_i32x4 rsrc_w = make_rsrc(p.wp, (unsigned)(p.nchunks * kWsP * wstep));
  auto issue_w = [&](int chunk, int wsel) __attribute__((always_inline)) {
    // piece q = (step, h | l) = 1 KB = one wave-DMA; this wave's pieces q = wave, wave + 4, ...: LDS and source both advance by
    // 4 pieces per instruction
    const unsigned dst = (unsigned)(size_t)(s_w + wsel * kWsWB + wave * 1024);
    int soff = __builtin_amdgcn_readfirstlane(chunk * (int)(kWsP * wstep) + wave * (int)wl_off);
    const int sstep = __builtin_amdgcn_readfirstlane(NW * (int)wl_off);
    unsigned keep;
    if (p.h2_dbg & 8) return;  // (timing only: no weight DMA)
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %2, %4, %1 offen lds\n\ts_add_u32 m0, m0, 0x1000\n\ts_add_u32 %1, %1, %5\n\t"
        "buffer_load_dwordx4 %2, %4, %1 offen lds\n\ts_add_u32 m0, m0, 0x1000\n\ts_add_u32 %1, %1, %5\n\t"
        "buffer_load_dwordx4 %2, %4, %1 offen lds\n\ts_add_u32 m0, m0, 0x1000\n\ts_add_u32 %1, %1, %5\n\t"
        "buffer_load_dwordx4 %2, %4, %1 offen lds\n\ts_add_u32 m0, m0, 0x1000\n\ts_add_u32 %1, %1, %5\n\t"
        "buffer_load_dwordx4 %2, %4, %1 offen lds\n\ts_add_u32 m0, m0, 0x1000\n\ts_add_u32 %1, %1, %5\n\t"
        "buffer_load_dwordx4 %2, %4, %1 offen lds\n\ts_add_u32 m0, m0, 0x1000\n\ts_add_u32 %1, %1, %5\n\t"
        "buffer_load_dwordx4 %2, %4, %1 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep), "+s"(soff)
        : "v"(wlane), "s"(dst), "s"(rsrc_w), "s"(sstep)
        : "memory", "scc");
  };
  const int *lp = s_qoff + kh;
  unsigned n_exec = 0;
  bool ovf_out = false;

  const int ngroups = (p.nposes + J - 1) / J;
  const int n_items = ngroups * tiles_per_pose;
  bool first_batch = true;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int wg = xcd_contiguous_id(item, n_items);
    const int b0 = (wg / tiles_per_pose) * J;
    const int npose = min(J, p.nposes - b0);
    int t = wg - (wg / tiles_per_pose) * tiles_per_pose;
    const int tz = t % p.ntz;
    t /= p.ntz;
    const int ty = t % p.nty, tx = t / p.nty;
    const int x0 = tx * 2 * p.tcx - 1, y0 = ty * 2 * p.tcy - 1, z0 = tz * 2 * p.tcz - 1;
    const float *in_b = p.in + (size_t)b0 * pose_floats;

    if (!first_batch) __syncthreads();  // every wave is through the previous batch's last K loop: LDS may be rewritten
    first_batch = false;

    // ---- the batch's rounds: live (chunk, pose) pairs, chunk-major, from the occupancy bytes of the blocks the halo tile touches ----
    {
      const int bx0 = x0 >> 2, by0 = y0 >> 2, bz0 = z0 >> 2;
      const int nbx = ((x0 + HX - 1) >> 2) - bx0 + 1, nby = ((y0 + HY - 1) >> 2) - by0 + 1, nbz = ((z0 + HZ - 1) >> 2) - bz0 + 1;
      const int nt = p.occ_nt;
#pragma unroll
      for (int k = 0; k < (J + NW - 1) / NW; k++) {
        const int tp = wave + k * NW;  // (wave-uniform)
        uint2 o8 = make_uint2(0u, 0u);
        if (tp < npose && lane < nbx * nby * nbz) {
          const int bz = bz0 + lane % nbz, by = by0 + (lane / nbz) % nby, bx = bx0 + lane / (nbz * nby);
          if ((unsigned)bx < (unsigned)nt && (unsigned)by < (unsigned)nt && (unsigned)bz < (unsigned)nt)
            o8 = *reinterpret_cast<const uint2 *>(p.in_occ + (((size_t)(b0 + tp) * nt + bx) * nt + by) * nt * 8 + (size_t)bz * 8);
        }
        unsigned m = 0u;
#pragma unroll
        for (int o = 0; o < 8; o++) {
          const unsigned byte = ((o < 4 ? o8.x : o8.y) >> (8 * (o & 3))) & 0xffu;
          if (__builtin_amdgcn_ballot_w64(byte != 0u) != 0ull) m |= 1u << o;
        }
        if (p.h2_dbg & 1) m = tp < npose ? (1u << p.nchunks) - 1u : 0u;  // (timing: every chunk live)
        if (tp < J && lane == 0) s_occ[tp] = (int)m;
      }
      __syncthreads();
      if (tid == 0) {
        int n = 0, last_first = -1;
        for (int c = 0; c < p.nchunks; c++) {
          bool first = true;
          for (int tp = 0; tp < J; tp++)
            if ((s_occ[tp] >> c) & 1) {
              if (first && last_first >= 0) s_nextc[last_first] = (unsigned char)c;
              if (first) last_first = n;
              s_nextc[n] = 0xff;
              s_round[n++] = (unsigned char)(c | (tp << 3) | (first ? 64 : 0));
              first = false;
            }
        }
        s_nr[0] = n;
      }
      __syncthreads();
    }
    const int NR = __builtin_amdgcn_readfirstlane(s_nr[0]);

    // ---- this tile's DMA sources: slot j = tid + i * 256 of a buffer is half j / PL of plane slot j % PL; voxels outside
    // the grid, pad slots and the slots behind 2 PL take an out-of-range offset (a buffer load returns zeros) ----
    unsigned voff[kWsNT];
    {
      int tid_i = tid;
      asm volatile("" : "+v"(tid_i));  // (opaque: the decomposition is not hoisted out of the batch loop -- conv3d_h2_dense.hip)
      const unsigned inv_sx = ((1u << 20) + SX - 1) / SX, inv_sy = ((1u << 20) + SY - 1) / SY;  // exact for n < 2^20 / d
#pragma unroll
      for (int i = 0; i < kWsNT; i++) {
        const int j = tid_i + i * 256;
        const int half = j >= PL ? 1 : 0;
        const int ps = j - half * PL;
        const int hx = (int)(((unsigned)ps * inv_sx) >> 20);
        const int r1 = ps - hx * SX;
        const int hy = (int)(((unsigned)r1 * inv_sy) >> 20), hz = r1 - hy * SY;
        const int x = x0 + hx, y = y0 + hy, z = z0 + hz;
        const bool ok = j < 2 * PL && hx < HX && hy < HY && hz < HZ && (unsigned)x < (unsigned)S && (unsigned)y < (unsigned)S && (unsigned)z < (unsigned)S;
        voff[i] = ok ? (unsigned)((x * S + y) * S + z) * 32u + (unsigned)half * 16u : 0x80000000u;
      }
    }
    auto issue_tile = [&](int r) __attribute__((always_inline)) {  // the tile of round r -> ring slot r % R
      const int e = __builtin_amdgcn_readfirstlane((int)s_round[r]);
      const int chunk = e & 7, tp = (e >> 3) & 7;
      const ws_i32x4 rs = make_rsrc(in_b + (size_t)tp * pose_floats, (unsigned)(pose_floats * 4));
      const unsigned dst = (unsigned)(size_t)(s_ring + (r % R) * kWsTileB + wave * 1024);
      const int soff = __builtin_amdgcn_readfirstlane(chunk * octet_bytes);
      unsigned keep;
      if (p.h2_dbg & 4) return;  // (timing only: no tile DMA)
      static_assert(kWsNT == 5, "five wave-DMAs per wave and tile are spelled out");
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
          "buffer_load_dwordx4 %1, %7, %8 offen lds\n\ts_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
          "buffer_load_dwordx4 %2, %7, %8 offen lds\n\ts_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
          "buffer_load_dwordx4 %3, %7, %8 offen lds\n\ts_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
          "buffer_load_dwordx4 %4, %7, %8 offen lds\n\ts_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
          "buffer_load_dwordx4 %5, %7, %8 offen lds\n\ts_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "s"(dst), "s"(rs), "s"(soff)
          : "memory", "scc");
    };

    ws_f32x16 acc[J][TM];
#pragma unroll
    for (int tp = 0; tp < J; tp++)
#pragma unroll
      for (int m = 0; m < TM; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tp][m][r] = 0.f;

    // ---- prologue: the first live chunk's weights, then the first R - 1 tiles ----
    int ci = 0;  // live chunks begun so far: chunk number ci of the batch has its weights in buffer ci & 1
    if (NR > 0) issue_w((int)s_round[0] & 7, 0);
#pragma unroll
    for (int r = 0; r < R - 1; r++)
      if (r < NR) issue_tile(r);
    int w_age = 0;  // tiles requested AFTER the youngest weights DMA (its instructions must have landed before its chunk starts)
    w_age = min(NR, R - 1);

    for (int r = 0; r < NR; r++) {
      const int e = __builtin_amdgcn_readfirstlane((int)s_round[r]);
      const int tp = (e >> 3) & 7;
      const bool first_of_chunk = (e & 64) != 0;
      if (first_of_chunk && r > 0) ci++;
      // tiles younger than round r's that are in flight: rounds r + 1 .. min(r + R - 2, NR - 1); at the first round of a chunk
      // the weights DMA'd for it must be through as well: no more younger tiles than were requested behind it
      int younger = min(R - 2, NR - 1 - r);
      if (first_of_chunk) younger = min(younger, w_age);
      ws_wait_vm(max(younger, 0));
      __syncthreads();  // every wave's part of tile r (and of these weights) is in LDS; every wave is through K loop r - 1
      if (first_of_chunk) {
        const int nc = __builtin_amdgcn_readfirstlane((int)s_nextc[r]);
        if (nc != 0xff) {  // the next live chunk's weights into the buffer the previous chunk's K loops have left
          issue_w(nc, (ci + 1) & 1);
          w_age = 0;
        }
      }
      if (r + R - 1 < NR) {
        issue_tile(r + R - 1);
        w_age++;
      }
      if (p.h2_dbg & 2) continue;  // (timing only: no K loop)

      const char *tile = s_ring + (r % R) * kWsTileB;
      const char *wl_ = s_w + (ci & 1) * kWsWB + lane * 16;
      auto k_loop = [&](auto tpc) __attribute__((always_inline)) {
        constexpr int TP = decltype(tpc)::value;
        if constexpr (TP < J) {
          uint4 wh0, wl0, wh1, wl1, ah0[TM], al0[TM], ah1[TM], al1[TM];
          ws_f32x16 &acc0 = acc[TP][0], &acc1 = acc[TP][1];  // (named here: asm operands alone do not make a generic lambda capture)
          unsigned &n_ex = n_exec;
          int qo_next = lp[0];
          auto load_pair = [&](int pr, uint4 *ah, uint4 *al, uint4 &wh, uint4 &wl) __attribute__((always_inline)) {
            const int qo = qo_next;
#pragma unroll
            for (int m = 0; m < TM; m++) {
              const char *a = tile + baseA[m] + qo;
              ah[m] = *reinterpret_cast<const uint4 *>(a);
              al[m] = *reinterpret_cast<const uint4 *>(a + PLB);
            }
            wh = *reinterpret_cast<const uint4 *>(wl_ + pr * 2048);
            wl = *reinterpret_cast<const uint4 *>(wl_ + pr * 2048 + 1024);
            qo_next = lp[2 * pr + 2];
          };
          // The MFMAs of a step as ONE asm statement.  With one wave per SIMD nothing fills the gaps of this wave's instruction
          // stream: an s_waitcnt or a scalar add that hipcc places between two MFMAs on the SAME accumulator costs ~43 cycles
          // (MI355X_MICROARCH.md), and three MFMAs in a row on one accumulator wait for each other -- the first version of this
          // kernel ran its K loops at a quarter of the MFMA rate.  Here the two M-tiles' chains are interleaved (per accumulator
          // still al * wh, ah * wl, ah * wh: same bits), every operand is waited for before the group, nothing sits inside it.
          auto mfma_pair = [&](const uint4 *ah, const uint4 *al, const uint4 &wh, const uint4 &wl) __attribute__((always_inline)) {
            // all 32 voxels x 16 k of an M-tile's step zero (h = 0 implies l = 0): nothing to add
            const unsigned any0 = ah[0].x | ah[0].y | ah[0].z | ah[0].w, any1 = ah[1].x | ah[1].y | ah[1].z | ah[1].w;
            unsigned long long lv0, lv1;
            asm volatile("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(lv0) : "v"(any0));
            asm volatile("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(lv1) : "v"(any1));
            const ws_f16x8 ah0v = __builtin_bit_cast(ws_f16x8, ah[0]), al0v = __builtin_bit_cast(ws_f16x8, al[0]), ah1v = __builtin_bit_cast(ws_f16x8, ah[1]),
                           al1v = __builtin_bit_cast(ws_f16x8, al[1]), whv = __builtin_bit_cast(ws_f16x8, wh), wlv = __builtin_bit_cast(ws_f16x8, wl);
            // (ONE statement with the three cases behind scalar branches INSIDE it: as three statements in three C++ branches
            // the register allocator gave every case its own accumulator registers and copied 16-64 AGPRs around per step --
            // right behind the MFMAs, whose results had not landed: slow and wrong)
            asm volatile(
                "s_nop 1\n\t"
                "s_cmp_eq_u64 %9, 0\n\ts_cbranch_scc1 .Lws_a%=\n\t"
                "s_cmp_eq_u64 %10, 0\n\ts_cbranch_scc1 .Lws_b%=\n\t"
                "v_mfma_f32_32x32x16_f16 %0, %4, %7, %0\n\tv_mfma_f32_32x32x16_f16 %1, %6, %7, %1\n\t"
                "v_mfma_f32_32x32x16_f16 %0, %3, %8, %0\n\tv_mfma_f32_32x32x16_f16 %1, %5, %8, %1\n\t"
                "v_mfma_f32_32x32x16_f16 %0, %3, %7, %0\n\tv_mfma_f32_32x32x16_f16 %1, %5, %7, %1\n\t"
                "s_add_u32 %2, %2, 2\n\ts_branch .Lws_d%=\n"
                ".Lws_b%=:\n\t"  // M-tile 0 only
                "v_mfma_f32_32x32x16_f16 %0, %4, %7, %0\n\tv_mfma_f32_32x32x16_f16 %0, %3, %8, %0\n\tv_mfma_f32_32x32x16_f16 %0, %3, %7, %0\n\t"
                "s_add_u32 %2, %2, 1\n\ts_branch .Lws_d%=\n"
                ".Lws_a%=:\n\t"  // M-tile 0 dead
                "s_cmp_eq_u64 %10, 0\n\ts_cbranch_scc1 .Lws_d%=\n\t"
                "v_mfma_f32_32x32x16_f16 %1, %6, %7, %1\n\tv_mfma_f32_32x32x16_f16 %1, %5, %8, %1\n\tv_mfma_f32_32x32x16_f16 %1, %5, %7, %1\n\t"
                "s_add_u32 %2, %2, 1\n"
                ".Lws_d%=:"
                : "+a"(acc0), "+a"(acc1), "+s"(n_ex)
                : "v"(ah0v), "v"(al0v), "v"(ah1v), "v"(al1v), "v"(whv), "v"(wlv), "s"(lv0), "s"(lv1)
                : "scc");
          };
          load_pair(0, ah0, al0, wh0, wl0);
#pragma unroll 1
          for (int pr = 0; pr < kWsP; pr += 2) {
            load_pair(pr + 1, ah1, al1, wh1, wl1);
            mfma_pair(ah0, al0, wh0, wl0);
            if (pr + 2 < kWsP) load_pair(pr + 2, ah0, al0, wh0, wl0);
            mfma_pair(ah1, al1, wh1, wl1);
          }
          // (the compiler does not know the asm statements were MFMAs: their results must have landed before any code of its
          // own -- accumulator moves between the poses' K loops, the epilogue -- reads them: 8-pass XDL write -> VALU read)
          asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc0), "+a"(acc1));
        }
      };
      switch (tp) {
        case 0: k_loop(std::integral_constant<int, 0>{}); break;
        case 1: k_loop(std::integral_constant<int, 1>{}); break;
        case 2: k_loop(std::integral_constant<int, 2>{}); break;
        case 3: k_loop(std::integral_constant<int, 3>{}); break;
        case 4: k_loop(std::integral_constant<int, 4>{}); break;
        case 5: k_loop(std::integral_constant<int, 5>{}); break;
        case 6: k_loop(std::integral_constant<int, 6>{}); break;
        default: k_loop(std::integral_constant<int, 7>{}); break;
      }
    }

    // ---- epilogue per pose (conv3d_h2_kernel's): un-scale, bias, ReLU, optional 2x2x2 pool; channels-last fp32 or split ----
    if (p.h2_dbg & 64) continue;  // (timing only)
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int row_e = tid_e & 31, kh_e = (tid_e >> 5) & 1, wave_e = __builtin_amdgcn_readfirstlane(tid_e >> 6);
    const int So = p.pool ? S / 2 : S;
    const int ncx = S / 2;
    const int ch = row_e;
    const int ch_sp = ((ch & 7) >> 1) + (row_e & 1) * 4;
    const size_t oct_sp = (size_t)(ch >> 3) * So * So * So;
    const float bias = ch < p.coutp ? p.bias[ch] : 0.f;
    auto finish_pose = [&](auto tpc) __attribute__((always_inline)) {
      constexpr int TP = decltype(tpc)::value;
      if constexpr (TP < J) {
        if (TP >= npose) return;
        const size_t out_pose = (size_t)(b0 + TP) * So * So * So * p.out_cs + p.out_c0;
        float *out_f = p.out + out_pose;
        auto store = [&](size_t vox, float v) __attribute__((always_inline)) {
          if (p.out_split) {
            ovf_out |= !(fabsf(v) <= 65504.f);
            const unsigned mine = ws_split1(v);
            const unsigned other = (unsigned)__builtin_amdgcn_mov_dpp((int)mine, 0xb1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
            const unsigned w = (row_e & 1) ? ((other >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (other << 16));
            if (ch < p.coutp) reinterpret_cast<unsigned *>(out_f)[(oct_sp + vox) * 8 + ch_sp] = w;
          } else if (ch < p.cout) {
            out_f[vox * p.out_cs + ch] = v;
          }
        };
#pragma unroll
        for (int m = 0; m < TM; m++) {
#pragma unroll
          for (int half = 0; half < 2; half++) {
            int cx, cy, cz;
            if (!cell_of(wave_e * TM + m, kh_e + 2 * half, cx, cy, cz)) continue;
            const int gcx = tx * p.tcx + cx, gcy = ty * p.tcy + cy, gcz = tz * p.tcz + cz;
            if (gcx >= ncx || gcy >= ncx || gcz >= ncx) continue;
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
              const float tt = acc[TP][m][half * 8 + r] * p.h2_unscale + bias;
              v[r] = p.relu ? fmaxf(tt, 0.f) : tt;
            }
            if (p.pool == 1) {
              float mx = v[0];
              int am = 0;
#pragma unroll
              for (int r = 1; r < 8; r++)
                if (v[r] > mx) mx = v[r], am = r;
              const size_t vox = ((size_t)gcx * So + gcy) * So + gcz;
              store(vox, mx);
              if (p.argmax_out && ch < p.cout) p.argmax_out[out_pose + vox * p.out_cs + ch] = (unsigned char)am;
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
    };
    finish_pose(std::integral_constant<int, 0>{});
    finish_pose(std::integral_constant<int, 1>{});
    finish_pose(std::integral_constant<int, 2>{});
    finish_pose(std::integral_constant<int, 3>{});
    finish_pose(std::integral_constant<int, 4>{});
    finish_pose(std::integral_constant<int, 5>{});
    finish_pose(std::integral_constant<int, 6>{});
    finish_pose(std::integral_constant<int, 7>{});
  }
  if (p.mfma_count && lane == 0) atomicAdd(p.mfma_count + (blockIdx.x & (kMfmaCountSlots - 1)), (unsigned long long)n_exec * (3u * 8u));
  if (p.h2_overflow && __builtin_amdgcn_ballot_w64(ovf_out) != 0ull && lane == 0) atomicOr(p.h2_overflow, 1u);
}

size_t conv_h2_ws_lds_bytes(int ring) { return (size_t)ring * kWsTileB + 2 * kWsWB + 32 * 4 + 8 * 4 + 4 * 4 + 128; }

// does the launch fit what the kernel covers?  (the 4 x 4 x 2-cell tile of the first convolutions, M-tiles stacked along x,
// 32 output channels, split-format input with occupancy bytes, no fused 1x1x1 conv, forward only)
bool conv_h2_ws_covers(const ConvArgs &p, int B) {
  int sy, sx, pl;
  conv_h2_planar_geo(p, &sy, &sx, &pl);
  return p.ksize == 3 && p.in_split && p.in_occ && p.sparse && p.nchunks <= 8 && p.coutp == 32 && !p.post_w && p.mt_x == 1 && p.tcx == 4 && p.tcy == 4 &&
         p.tcz == 2 && 2 * pl <= kWsNT * 256 && !p.bn_scale && !p.in_amax && !p.out_amax && !p.out_mask && !p.out_scale && !p.accumulate && p.in_mode == 0 &&
         p.in_cs % 8 == 0 && (!p.out_split || (p.out_cs % 8 == 0 && !p.out_c0 && p.coutp == p.cout && !p.argmax_out)) && B >= 1;
}

void launch_conv_h2_ws(ConvArgs p, int B, int ring, hipStream_t s) {
  if (!conv_h2_ws_covers(p, B)) throw Error(2, "launch_conv_h2_ws: launch outside what the kernel covers");
  p.nposes = B;
  constexpr int J = 8;
  const int n_items = (B + J - 1) / J * p.ntx * p.nty * p.ntz;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  static std::mutex mu;
  static std::map<int, int> cu_cache;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cu_cache.find(dev);
    if (it == cu_cache.end()) {
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      it = cu_cache.emplace(dev, std::max(cus, 8)).first;
    }
    cus = it->second;
  }
  const int grid = std::min(n_items, cus & ~7);
  auto go = [&](auto kern, int r) {
    ensure_max_lds(reinterpret_cast<const void *>(kern), 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), conv_h2_ws_lds_bytes(r), s, p);
  };
  if (ring >= 5) go(conv3d_h2_ws_kernel<J, 5>, 5);
  else if (ring == 4) go(conv3d_h2_ws_kernel<J, 4>, 4);
  else if (ring == 3) go(conv3d_h2_ws_kernel<J, 3>, 3);
  else go(conv3d_h2_ws_kernel<J, 2>, 2);
}

}  // namespace mig
