// conv3d.hip -- 3x3x3 / 1x1x1 stride-1 "same" 3-D convolution as an fp32-MFMA implicit GEMM (gfx950).
//
// Replaces the cuDNN/oneDNN `_convolution` calls inside the TorchScript graphs that gnina runs
// at gninasrc/lib/torch_model.cpp:185 (layer lists: SURVEY.md App. B), with bias, ReLU, eval
// BatchNorm-on-input (Dense family) and the following 2x2x2 Max/AvgPool fused in.
//
// GEMM view:  M = output voxels,  N = output channels,  K = taps x input channels.
//   A[m][k] is gathered on the fly from an LDS-resident halo tile of the channels-last input,
//   B[k][n] streams from a pre-packed weight array (L2 resident, 1.5 MB per network),
//   D accumulates in registers through v_mfma_f32_32x32x2_f32 -- exact fp32 (an fmaf chain),
//   which is what the 1e-4 score parity budget needs; there is no TF32 on gfx950.
//
// Wavefront tiling (64 lanes, not a warp-shaped CUDA tiling):
//   * An M-tile is 32 voxels = four 2x2x2 pooling cells.  Row i of the MFMA maps to
//     (cell = i.bit2 + 2*i.bit4, x = i.bit3, y = i.bit1, z = i.bit0), so that in the 32x32
//     accumulator layout (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) registers 0-7 of a lane are
//     the eight voxels of ONE pooling cell and registers 8-15 those of another: ReLU + 2x2x2
//     max/avg pooling happen entirely in registers, no cross-lane traffic, and the pooled
//     activation (8x smaller) is what goes back to HBM.
//   * K runs over "quads" (one tap x 4 consecutive input channels).  One ds_read_b128 per lane
//     feeds four MFMAs; lanes 0-31 take quad 2p, lanes 32-63 quad 2p+1 (the k=0 / k=1 halves of
//     the 32x32x2 instruction), each with its own LDS address, so K needs no padding beyond a
//     multiple of 4 channels and an even quad count.
//   * A workgroup is WM x WN waves, each wave owning TM M-tiles x TN 32-wide N-tiles.
//
// Zero-skipping (exact: a skipped instruction would only have added +-0 to accumulators that cannot hold -0):
//   1. per tile: channel quads that are all-zero in the whole halo tile are left out of the K-loop list (first conv on
//      the pooled voxel grid, transposed convs on un-pooled gradients: ConvArgs::sparse == 1);
//   2. per MFMA: the lane mask "A component != 0" of every instruction is taken with one v_cmp into an SGPR pair, and
//      the instruction is branched around when the mask is empty (sparse == 1 and == 2; the latter is the mode of
//      ReLU'd activations: every quad listed, so the K order -- channel-major, ConvArgs::korder -- and with it the
//      rounding does not depend on the tile);
//   the N = 16 kernel of the Dense blocks does 2. on x * bn_scale, with the BatchNorm shift folded into a bias table.
#include "common.h"

#include <type_traits>
#include "conv3d.h"

#include <stdexcept>

#ifndef MI_CONV_EXPERIMENT
#define MI_CONV_EXPERIMENT 0  // 1, 2: timing experiments on the conv kernel (wrong results; never in the product build)
#endif

namespace mig {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// "some component is not +0" on the bit patterns: v_or3 + v_or + v_cmp.  (Written as x.x != 0.f || x.y != 0.f || ... the
// compiler materialises four booleans with v_cmp / v_cndmask pairs and merges them with 16-bit shifts: 16 VALU
// instructions and their SGPR hazard slots per staged float4.  -0 counts as non-zero here, as in the K loop's own test;
// either answer is exact: a skipped +-0 operand would only have added +-0 to accumulators that cannot hold -0.)
__device__ __forceinline__ bool any_bits(const float4 &x) {
  return (__float_as_uint(x.x) | __float_as_uint(x.y) | __float_as_uint(x.z) | __float_as_uint(x.w)) != 0u;
}

// Eval BatchNorm of one staged channel quad.  The quad index is wave-uniform, so scale and shift are read through the
// constant address space: two s_load_dwordx4 into SGPRs (the tables are written once at model load).  As plain global
// loads they sat in the vector memory queue BEHIND the batch of activation loads and were waited on with vmcnt(0) per
// quad -- one full L2 latency per staged float4, and the "U loads in flight" batching gone.
typedef float bn_f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) bn_f32x4 *ConstQuadPtr;
struct BnQuad {
  bn_f32x4 sc, sh;
};
__device__ __forceinline__ BnQuad bn_load(const float *scale, const float *shift, int c) {
  return {*(ConstQuadPtr)(const void *)(scale + c), *(ConstQuadPtr)(const void *)(shift + c)};
}
__device__ __forceinline__ void bn_apply(float4 &x, const BnQuad &q) {
  x.x = x.x * q.sc.x + q.sh.x;
  x.y = x.y * q.sc.y + q.sh.y;
  x.z = x.z * q.sc.z + q.sh.z;
  x.w = x.w * q.sc.w + q.sh.w;
}

// (second launch bound = waves per SIMD the register allocation must leave room for: the two-accumulator shapes run
// four workgroups per CU, i.e. <= 128 VGPRs; the TM = 7 shape two)
// SP: 0 dense K loop (every quad, order by ConvArgs::korder); 1 K loop over the per-chunk list of surviving quads
// (channel-major) with the per-MFMA zero test
template <int WM, int WN, int TM, int TN, int SP, bool MTX>
__global__ __launch_bounds__(64 * WM * WN, ((TM * TN <= 2 || (TM == 1 && TN == 3)) ? 4 : TM * TN <= 7 ? 2 : 1)) void conv3d_mfma_kernel(ConvArgs p) {
  constexpr bool SPARSE = SP != 0;
  constexpr int NWAVES = WM * WN;
  constexpr int NTHREADS = 64 * NWAVES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // (readfirstlane: the wave index is uniform, which the compiler cannot see through threadIdx -- with it the M-tile /
  // cell arithmetic below, divisions by run-time tile dimensions included, runs on the scalar unit)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int kh = lane >> 5;   // which k of the 32x32x2 MFMA this lane feeds
  const int row = lane & 31;  // A row / B column

  const int tiles_per_pose = p.ntx * p.nty * p.ntz;
  const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int b = wg / tiles_per_pose;
  int t = wg - b * tiles_per_pose;
  const int tz = t % p.ntz;
  t /= p.ntz;
  const int ty = t % p.nty, tx = t / p.nty;
  const int n_base = (blockIdx.y * WN + wn) * TN * 32;  // first output channel of this wave

  const int halo = p.ksize == 3 ? 1 : 0;
  const int HX = 2 * p.tcx + 2 * halo, HY = 2 * p.tcy + 2 * halo, HZ = 2 * p.tcz + 2 * halo;
  const int HV = HX * HY * HZ;
  const int CC4 = p.cc4, CCs = p.ccs;
  const int taps = p.ksize == 3 ? 27 : 1;
  const int Q = taps * CC4;       // quads per chunk
  const int P = (Q + 1) >> 1;     // quad pairs per chunk

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *s_tile = smem;                                  // [HV][CCs]
  // [Q + 5] K-loop list of this chunk: .x = byte offset of the quad inside the halo tile (tap shift + channel quad),
  // .y = byte offset of its packed weight row.  One ds_read_b64 per quad, fetched one pair ahead of its use; the
  // list ends in five pad entries (zero weights) so that neither an odd quad count nor the read-ahead needs a branch.
  int2 *s_list = reinterpret_cast<int2 *>(smem + (size_t)HV * CCs);
  int *s_flag = reinterpret_cast<int *>(s_list + Q + 5);  // [nchunks][CC4] "this channel quad is non-zero in the tile"
  int *s_vox = s_flag + p.nchunks * CC4;                 // [HV] voxel index of every halo position inside the pose, -1 = padding
  // [27] sparse list building: .x = byte offset of snake tap i inside the halo tile, .y = its tap index (weight row)
  // (in_mode 3 keeps a second [HV] table behind the first: the voxel's cell in the half-resolution gradient)
  int2 *s_tap = reinterpret_cast<int2 *>((reinterpret_cast<size_t>(s_vox + (p.in_mode == 3 ? 2 : 1) * HV) + 7) & ~(size_t)7);  // (8-byte aligned)
  const int wstride_i = p.coutp * 4;                     // floats per quad row of packed weights
  auto list_entry = [&](int q, int wrow) -> int2 {
    const int tap = q / CC4, c4 = q - tap * CC4;
    const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    return make_int2(((p.ksize == 3 ? ((dx * HY + dy) * HZ + dz) * CCs : 0) + c4 * 4) * 4, wrow * wstride_i * 4);
  };

  // sparse == 1: a flag is raised by the staging loops when the quad is non-zero somewhere in the tile.  sparse == 2
  // (ReLU'd activations): every existing quad stays listed -- the K loop runs in the list's channel-major order with
  // its per-MFMA zero test, and that order does not depend on the tile or its contents
  const bool detect = SPARSE && p.sparse == 1;
  if (SPARSE && tid < taps) {  // the tap walk of the per-chunk lists, once per workgroup (the list loop then has no divisions)
    const int tap = taps == 27 ? conv_snake_tap(tid) : tid;
    const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    s_tap[tid] = make_int2(p.ksize == 3 ? ((dx * HY + dy) * HZ + dz) * CCs * 4 : 0, tap);
  }
  for (int i = tid; i < p.nchunks * CC4; i += NTHREADS) s_flag[i] = (SPARSE && p.sparse == 2 && i < p.cin4) ? 1 : 0;
  // Row Q of every chunk's packed weights is all zero (ConvArgs::wrows): the list entry behind the last quad points
  // there, so an odd number of quads needs no special case in the K loop (the idle half-wave multiplies by zeros).
  if (!SPARSE)  // dense: every quad, tap-major -- or (korder 1) in the channel-major order of the listed K loop, so that a
                // layer gives the same bits with and without the per-MFMA test (ConvArgs::korder)
    for (int q = tid; q < Q + 5; q += NTHREADS) {
      int qq = q < Q ? q : Q - 1;
      if (p.korder) {
        const int c4 = qq / taps, i = qq - c4 * taps;
        qq = (taps == 27 ? conv_snake_tap(i) : i) * CC4 + c4;
      }
      s_list[q] = list_entry(qq, q < Q ? qq : Q);
    }

  // A-row geometry of this lane for each of its M-tiles
  const int NC = p.tcx * p.tcy * p.tcz;
  const int oz = row & 1, oy = (row >> 1) & 1, ox = (row >> 3) & 1;
  const int cell_in_mt = ((row >> 2) & 1) + 2 * ((row >> 4) & 1);
  // cell `cim` (0..3) of M-tile `mt` -> cell coordinates inside the workgroup tile (ConvArgs::mt_x)
  auto cell_of = [&](int mt, int cim, int &cx, int &cy, int &cz) -> bool {
    if (MTX) {
      cz = mt % p.tcz;
      cy = (mt / p.tcz) % p.tcy;
      cx = 4 * (mt / (p.tcz * p.tcy)) + cim;
      return cx < p.tcx;
    }
    const int cell = mt * 4 + cim;
    cz = cell % p.tcz, cy = (cell / p.tcz) % p.tcy, cx = cell / (p.tcz * p.tcy);
    return cell < NC;
  };
  int baseA[TM];
#pragma unroll
  for (int m = 0; m < TM; m++) {
    int cx, cy, cz;
    if (!cell_of(wm * TM + m, cell_in_mt, cx, cy, cz)) cx = cy = cz = 0;
    baseA[m] = (((2 * cx + ox) * HY + (2 * cy + oy)) * HZ + (2 * cz + oz)) * CCs * 4;  // bytes
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; m++)
#pragma unroll
    for (int n = 0; n < TN; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

  const int S = p.S;
  const int x0 = tx * 2 * p.tcx - halo, y0 = ty * 2 * p.tcy - halo, z0 = tz * 2 * p.tcz - halo;
  // halo position -> voxel, once per workgroup: the per-chunk staging loops then run without integer divisions
  // hv / HZ and (hv / HZ) / HY by multiply-shift (exact for hv < 2^20 / 16; HV is a few hundred): a generic 32-bit
  // division is ~25 VALU instructions, and this loop runs them per halo voxel
  const unsigned inv_hz = ((1u << 20) + HZ - 1) / HZ, inv_hy = ((1u << 20) + HY - 1) / HY;
  for (int hv = tid; hv < HV; hv += NTHREADS) {
    const int t1 = (int)(((unsigned)hv * inv_hz) >> 20), hz = hv - t1 * HZ;
    const int hx = (int)(((unsigned)t1 * inv_hy) >> 20), hy = t1 - hx * HY;
    const int x = x0 + hx, y = y0 + hy, z = z0 + hz;
    const bool in = (unsigned)x < (unsigned)S && (unsigned)y < (unsigned)S && (unsigned)z < (unsigned)S;
    int v = -1;
    if (in)
      v = p.in_mode == 2 ? (((x >> 1) * (S >> 1) + (y >> 1)) * (S >> 1) + (z >> 1)) | ((((x & 1) << 2) | ((y & 1) << 1) | (z & 1)) << 28)
          : p.in_mode == 0 ? ((x * S + y) * S + z) * p.in_cs  // forward convs: the float offset of the voxel's channel row
                           : (x * S + y) * S + z;
    s_vox[hv] = v;
    if (p.in_mode == 3) s_vox[HV + hv] = in ? ((x >> 1) * (S >> 1) + (y >> 1)) * (S >> 1) + (z >> 1) : -1;
    // forward staging writes the voxels inside the grid only: the zero padding is laid down once, here
    if (p.in_mode == 0 && !in)
      for (int c = 0; c < CCs; c += 4) *reinterpret_cast<float4 *>(s_tile + hv * CCs + c) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const unsigned inv_cc4 = ((1u << 20) + CC4 - 1) / CC4;  // it / CC4 == (it * inv_cc4) >> 20 for it < 2^20 / CC4
  const float *in_b = p.in + (size_t)b * S * S * S * p.in_cs;
  const size_t wstride = (size_t)p.coutp * 4;           // floats per quad row of packed weights

  unsigned n_exec = 0;  // MFMA instructions this wave executed (SP == 1; a scalar counter, read in profile mode only)
  for (int chunk = 0; chunk < p.nchunks; chunk++) {
    __syncthreads();  // previous chunk's reads done (and s_list / s_vox visible on the first pass)
    // ---- stage the halo tile of this channel chunk into LDS (zero padded; BN folded in) ----
    const int c_base = chunk * CC4 * 4;
    // All global loads of a batch of U items per thread are issued before the first one is consumed: a
    // dependent load -> (mask, BN) -> ds_write chain per item would serialise one L2/HBM latency per iteration.
    const int total_items = HV * CC4;
    constexpr int U = 4;
    if (p.in_mode == 0) {
      // Forward convs, voxel-major: a thread takes halo voxels tid, tid + NTHREADS, ... and walks the channel quads of
      // the chunk with immediate offsets -- no per-item index arithmetic (fp32 MFMA and VALU share the SIMD's fp32
      // lanes on gfx950: every VALU instruction here is time taken from the K loops of the co-resident waves).
      const float *src_c = in_b + c_base;
      const int nq = min(CC4, p.cin4 - chunk * CC4);  // quads of this chunk that exist in the input
      // (two copies of the loop, with and without BatchNorm, and full groups of U quads apart from the tail: straight-line
      // code lets the compiler issue the U vector loads and the 2 U scalar loads together and feed the SGPRs to
      // v_pk_mul / v_pk_add directly)
      auto stage = [&](auto has_bn) {
        constexpr bool BN = decltype(has_bn)::value;
        for (int hv = tid; hv < HV; hv += NTHREADS) {
          const int off = s_vox[hv];
          if (off < 0) continue;  // zero padding, already in place
          float *dst = s_tile + hv * CCs;
          const float *src = src_c + off;
          int qb = 0;
          for (; qb + U <= nq; qb += U) {
            float4 val[U];
            BnQuad bn[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
#if MI_CONV_EXPERIMENT == 1  // (tools/conv_experiments.sh: staging without its global loads)
              val[u] = make_float4(1.f, 0.5f, 0.25f, 2.f);
#else
              val[u] = *reinterpret_cast<const float4 *>(src + (qb + u) * 4);
#endif
              if constexpr (BN) bn[u] = bn_load(p.bn_scale, p.bn_shift, c_base + (qb + u) * 4);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
              float4 x = val[u];
              if constexpr (BN) bn_apply(x, bn[u]);  // eval BatchNorm on the conv input; padding stays exactly 0
              *reinterpret_cast<float4 *>(dst + (qb + u) * 4) = x;
              if (detect && any_bits(x)) s_flag[chunk * CC4 + qb + u] = 1;
            }
          }
          if (qb < nq) {  // the tail of 1 .. U - 1 quads, its loads batched as well
            float4 val[U - 1];
#pragma unroll
            for (int u = 0; u < U - 1; u++)
              if (qb + u < nq) {
#if MI_CONV_EXPERIMENT == 1
                val[u] = make_float4(1.f, 0.5f, 0.25f, 2.f);
#else
                val[u] = *reinterpret_cast<const float4 *>(src + (qb + u) * 4);
#endif
              }
#pragma unroll
            for (int u = 0; u < U - 1; u++)
              if (qb + u < nq) {
                float4 x = val[u];
                if constexpr (BN) bn_apply(x, bn_load(p.bn_scale, p.bn_shift, c_base + (qb + u) * 4));
                *reinterpret_cast<float4 *>(dst + (qb + u) * 4) = x;
                if (detect && any_bits(x)) s_flag[chunk * CC4 + qb + u] = 1;
              }
          }
        }
      };
      if (p.bn_scale)
        stage(std::true_type{});
      else
        stage(std::false_type{});
      if (nq < CC4)  // partial last chunk: its missing quads still hold the previous chunk's channels
        for (int it = tid; it < HV * (CC4 - nq); it += NTHREADS) {
          const int hv = it / (CC4 - nq), c4 = nq + it - hv * (CC4 - nq);
          *reinterpret_cast<float4 *>(s_tile + hv * CCs + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else
    for (int base = 0; base < total_items; base += NTHREADS * U) {
      float4 val[U], act[U];
      uchar4 am[U];
      int dst[U], cq[U], rr[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int it = base + u * NTHREADS + tid;
        val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        act[u] = make_float4(1.f, 1.f, 1.f, 1.f);
        am[u] = make_uchar4(0, 0, 0, 0);
        dst[u] = -1;
        cq[u] = -1;
        rr[u] = 0;
        if (it < total_items) {
          const int hv = (int)(((unsigned)it * inv_cc4) >> 20), c4 = it - hv * CC4;
          const int vi = s_vox[hv];
          dst[u] = hv * CCs + c4 * 4;
          if (vi >= 0 && chunk * CC4 + c4 < p.cin4) {
            const int c = c_base + c4 * 4;
            cq[u] = c4;
            if (p.in_mode == 2) {
              // transposed conv fed by a max-pooled gradient: un-pool while staging (no full-size tensor in HBM)
              const int Sh = S >> 1;
              const size_t cell = (size_t)b * Sh * Sh * Sh + (vi & 0x0fffffff);
              rr[u] = vi >> 28;
              val[u] = *reinterpret_cast<const float4 *>(p.in + cell * p.in_cs + c);
              am[u] = *reinterpret_cast<const uchar4 *>(p.in_argmax + cell * p.in_cs + c);
              act[u] = *reinterpret_cast<const float4 *>(p.in_act + cell * p.in_act_cs + c);
            } else if (p.in_mode == 3) {
              // transposed conv fed by an average-pooled gradient: every voxel of a cell receives 1/8 of the cell's
              // gradient, masked by its own forward activation -- un-pooled while staging, like the max pool above
              const int Sh = S >> 1;
              const size_t cell = (size_t)b * Sh * Sh * Sh + s_vox[HV + hv];
              const float4 g = *reinterpret_cast<const float4 *>(p.in + cell * p.in_cs + c);
              val[u] = make_float4(g.x * 0.125f, g.y * 0.125f, g.z * 0.125f, g.w * 0.125f);
              act[u] = *reinterpret_cast<const float4 *>(p.in_act + ((size_t)b * S * S * S + vi) * p.in_act_cs + c);
            } else {
#if MI_CONV_EXPERIMENT == 1  // (tools/conv_experiments.sh: staging without its global loads)
              val[u] = make_float4(1.f, 0.5f, 0.25f, 2.f);
#else
              val[u] = *reinterpret_cast<const float4 *>(in_b + (size_t)vi * p.in_cs + c);
#endif
              if (p.in_mode == 1)  // ReLU backward: gradient passes where the forward activation was > 0
                act[u] = *reinterpret_cast<const float4 *>(p.in_act + ((size_t)b * S * S * S + vi) * p.in_act_cs + c);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (dst[u] < 0) continue;
        float4 v = val[u];
        if (cq[u] >= 0) {
          if (p.in_mode == 2) {
            v.x = (am[u].x == rr[u] && act[u].x > 0.f) ? v.x : 0.f;
            v.y = (am[u].y == rr[u] && act[u].y > 0.f) ? v.y : 0.f;
            v.z = (am[u].z == rr[u] && act[u].z > 0.f) ? v.z : 0.f;
            v.w = (am[u].w == rr[u] && act[u].w > 0.f) ? v.w : 0.f;
          } else {
            if (p.in_mode == 1 || p.in_mode == 3) {
              v.x = act[u].x > 0.f ? v.x : 0.f;
              v.y = act[u].y > 0.f ? v.y : 0.f;
              v.z = act[u].z > 0.f ? v.z : 0.f;
              v.w = act[u].w > 0.f ? v.w : 0.f;
            }
            if (p.bn_scale) {  // eval BatchNorm on the conv input; padding stays exactly 0
              const int c = c_base + cq[u] * 4;
              const float4 sc = *reinterpret_cast<const float4 *>(p.bn_scale + c);
              const float4 sh = *reinterpret_cast<const float4 *>(p.bn_shift + c);
              v.x = v.x * sc.x + sh.x;
              v.y = v.y * sc.y + sh.y;
              v.z = v.z * sc.z + sh.z;
              v.w = v.w * sc.w + sh.w;
            }
          }
        }
        *reinterpret_cast<float4 *>(s_tile + dst[u]) = v;
        if (detect && cq[u] >= 0 && any_bits(v)) s_flag[chunk * CC4 + cq[u]] = 1;
      }
    }
    __syncthreads();

    // ---- input sparsity (the pooled voxel grid is ~12 % dense, SURVEY "Hard parts"): channel quads
    // that are entirely zero inside this halo tile contribute exact zeros, so their 27 taps are
    // skipped.  The surviving quads keep their original order; only which two of them share one two-k MFMA
    // instruction can change (measured effect on scores: none for Default2017/2018, <= 6e-7 for Dense).
    int J = Q;
    if (SPARSE) {
      int n_act = 0;
      for (int c = 0; c < CC4; c++) n_act += s_flag[chunk * CC4 + c];
      if (n_act == 0) continue;
      J = taps * n_act;
      // Channel-major order (all 27 taps of one surviving quad, then the next quad): the two quads of a pair are then
      // the same four channels one tap apart, i.e. nearly the same voxels -- so that the per-M-tile zero test in the K
      // loop below finds BOTH halves of an MFMA's A operand empty about as often as one.
      // (boustrophedon walk of the 3x3x3 taps, conv_snake_tap: consecutive taps are always face neighbours; its tile
      // offsets come from s_tap -- this loop runs on the first J threads while the rest of the workgroup waits)
      for (int j = tid; j < J; j += NTHREADS) {
        const int a = taps == 27 ? (int)(((unsigned)j * 2428u) >> 16) : j;  // j / 27 for j < 4,000
        const int2 tp = s_tap[j - a * taps];
        int c4 = a;
        if (n_act != CC4) {
          int seen = -1;
          for (int c = 0; c < CC4; c++) {
            seen += s_flag[chunk * CC4 + c];
            if (seen == a) {
              c4 = c;
              break;
            }
          }
        }
        const int q = tp.y * CC4 + c4;
        s_list[j] = make_int2(tp.x + c4 * 16, q * wstride_i * 4);
      }
      if (tid < 5) s_list[J + tid] = list_entry(Q - 1, Q);  // the pad entries: zero weights
      __syncthreads();
    }

    // ---- K loop over quad pairs, unrolled by two with two operand register sets (ping-pong): the LDS
    // operands and the L2-resident weight rows of pair pr + 1 are in flight while the eight MFMAs per
    // (M-tile, N-tile) of pair pr run.  A single-set "fetch next, then copy" loop gets folded back by the
    // compiler into load -> s_waitcnt vmcnt(0) -> MFMA, which exposes the L2 latency once per pair.
    // Dense: pair pr = quads (2 pr, 2 pr + 1), affine addressing.  Sparse: pairs of the compacted list
    // of surviving quads, in their original order.
#if MI_CONV_EXPERIMENT == 2  // (tools/conv_experiments.sh: everything but the K loop)
    const int NP = 0;
#else
    const int NP = SPARSE ? (J + 1) >> 1 : P;
#endif
    // the list entry of this half-wave's quad of pair pr; fetched one pair ahead so that the two dependent LDS
    // round trips (entry -> operands) are never both on the critical path of an iteration
    const int2 *lp = s_list + kh;
    int2 e_next = lp[0];
    const char *wbase = reinterpret_cast<const char *>(p.wp) + (size_t)chunk * p.wrows * wstride * 4;  // uniform
    const unsigned wlane = (unsigned)(n_base + row) * 16u;
    auto load_pair = [&](int pr, float4 *aa, float4 *ww) {
      const int2 e = e_next;
#pragma unroll
      for (int m = 0; m < TM; m++) aa[m] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_tile) + baseA[m] + e.x);
#pragma unroll
      for (int n = 0; n < TN; n++) ww[n] = *reinterpret_cast<const float4 *>(wbase + (wlane + (unsigned)e.y + (unsigned)n * 512u));
      e_next = lp[2 * pr + 2];
    };
    auto mfma_pair = [&](const float4 *aa, const float4 *ww) {
      if constexpr (SP == 1) {
        // Third level of zero-skipping (sparse inputs only), per MFMA: the A operand of one instruction is ONE input
        // channel at 32 voxels x the pair's two taps.  If it is zero in every lane the instruction would add exact
        // zeros to the accumulators and is skipped (results bit-identical to executing it).  The channels of the
        // pooled voxel grid are individually sparse (one atom type each): of the operands that survive the tile-level
        // quad list about a third are non-zero.  All eight lane masks are taken first, straight into SGPR pairs
        // (v_cmp_ne_u32 with an SGPR destination; written as asm because the compiler turns `bits != 0` into
        // v_cmp_class -> vcc -> s_cbranch_vccz, one VALU-to-branch round trip per MFMA, and a ballot of that into
        // v_cndmask + v_cmp), then the branches are scalar compares that run beside the other waves' MFMAs.  The
        // instructions of the two M-tiles alternate, so two dependent MFMAs are never adjacent.
        {
          unsigned long long live[TM][4];
#pragma unroll
          for (int m = 0; m < TM; m++) {
            const float ac[4] = {aa[m].x, aa[m].y, aa[m].z, aa[m].w};
#pragma unroll
            for (int j = 0; j < 4; j++) asm("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(live[m][j]) : "v"(ac[j]));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; j++)
#pragma unroll
            for (int m = 0; m < TM; m++) {
              if (live[m][j] == 0ull) continue;
              n_exec += TN;
              const float a = j == 0 ? aa[m].x : j == 1 ? aa[m].y : j == 2 ? aa[m].z : aa[m].w;
#pragma unroll
              for (int n = 0; n < TN; n++) {
                const float wc = j == 0 ? ww[n].x : j == 1 ? ww[n].y : j == 2 ? ww[n].z : ww[n].w;
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wc, acc[m][n], 0, 0, 0);
              }
            }
          return;
        }
      }
      {
#pragma unroll
        for (int m = 0; m < TM; m++)
#pragma unroll
          for (int n = 0; n < TN; n++) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[m].x, ww[n].x, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[m].y, ww[n].y, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[m].z, ww[n].z, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[m].w, ww[n].w, acc[m][n], 0, 0, 0);
          }
      }
    };
#if MI_CONV_EXPERIMENT == 3 || MI_CONV_EXPERIMENT == 4
    // (tools/conv_experiments.sh) What would a K loop cost whose liveness bits come for free (precomputed per chunk
    // from an occupancy bitmap) and which touches live pairs only?  Pseudo-random bits on the scalar unit with the
    // measured statistics -- 35 % of the pairs entirely dead, half of the eight MFMAs of the others live (0.33
    // executed overall); x3 skips dead pairs before their loads, x4 also drops the loads' address adds by fetching
    // a fixed entry.  Wrong results by construction.
    if constexpr (SP == 1) {
      float4 a0[TM], w0[TN];
      for (int pr = 0; pr < NP; pr++) {
        unsigned h = (unsigned)pr * 2654435761u ^ (unsigned)wg * 40503u ^ (unsigned)(chunk * 97 + wave * 13);
        h ^= h >> 13;
        h *= 0x5bd1e995u;
        h ^= h >> 15;
        h = __builtin_amdgcn_readfirstlane(h);
        if ((h & 0xfffffu) < 367001u) continue;  // 35 % of the pairs: nothing live, nothing loaded
        const unsigned bits = (h >> 20) & 0xffu;  // each MFMA live with probability 1/2
        e_next = lp[2 * pr];
        load_pair(pr, a0, w0);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int m = 0; m < TM; m++) {
            if (!((bits >> (2 * j + m)) & 1u)) continue;
            n_exec += TN;
            const float a = j == 0 ? a0[m].x : j == 1 ? a0[m].y : j == 2 ? a0[m].z : a0[m].w;
#pragma unroll
            for (int n = 0; n < TN; n++) {
              const float wc = j == 0 ? w0[n].x : j == 1 ? w0[n].y : j == 2 ? w0[n].z : w0[n].w;
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wc, acc[m][n], 0, 0, 0);
            }
          }
      }
      continue;
    }
#endif
    float4 a0[TM], a1[TM], w0[TN], w1[TN];
    load_pair(0, a0, w0);
    int pr = 0;
    // sched_barrier: the machine scheduler otherwise sinks each fetch down to its consumer (fetch -> wait -> MFMA),
    // which exposes the LDS and L2 latencies once per pair
    for (; pr + 1 < NP; pr += 2) {
      load_pair(pr + 1, a1, w1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_pair(a0, w0);
      __builtin_amdgcn_sched_barrier(0);
      load_pair(pr + 2, a0, w0);  // (past the end: a clamped, unused fetch -- keeps the wait counts static, no branch)
      __builtin_amdgcn_sched_barrier(0);
      mfma_pair(a1, w1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NP & 1) mfma_pair(a0, w0);
  }

  // ---- fused 1x1x1 conv (conv -> ReLU -> conv1 -> ...): the activated tile is transposed through LDS from the
  // accumulator layout (lane = channel, registers = voxels) into A-operand layout (lane = voxel, four
  // consecutive channels), every wave then runs its (M-tile, N-tile) block of the cout x cout GEMM ----
  const float *bias_ptr = p.bias;
  int relu_flag = p.relu;
  constexpr bool kPostCapable = TM <= 3 && TN == 1;  // (compiled out of the register-heavy shapes; engine.cpp checks)
  if (kPostCapable && p.post_w) {
    __syncthreads();  // the halo tile and its index tables are dead: reuse the LDS
    const int stride = p.coutp + 4;  // odd multiple of 16 bytes per voxel row (coutp = 32 or 64)
    float *s_mid = smem;
#pragma unroll
    for (int m = 0; m < TM; m++)
#pragma unroll
      for (int n = 0; n < TN; n++) {
        const int ch = n_base + n * 32 + row;
        const float b1 = p.bias[ch];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * kh;  // accumulator row of register r
          float v = acc[m][n][r] + b1;
          if (p.relu) v = fmaxf(v, 0.f);
          s_mid[((wm * TM + m) * 32 + i) * stride + ch] = v;
          acc[m][n][r] = 0.f;
        }
      }
    __syncthreads();
    // K = coutp channels of the 1x1 conv, in the chunked layout of its own plan (post_cc4 quads per chunk, an even
    // number; post_wrows rows per chunk -- the rows behind the last quad of a chunk are the K loop's zero rows)
    const float *w2 = p.post_w + (size_t)(n_base + row) * 4 + (size_t)kh * wstride;
    const int P2 = p.post_cc4 >> 1;
    for (int c2 = 0; c2 * p.post_cc4 * 4 < p.coutp; c2++)
      for (int pr = 0; pr < P2; pr++) {
        float4 a2[TM], b2[TN];
#pragma unroll
        for (int m = 0; m < TM; m++)
          a2[m] = *reinterpret_cast<const float4 *>(s_mid + ((wm * TM + m) * 32 + row) * stride + (c2 * p.post_cc4 + 2 * pr + kh) * 4);
#pragma unroll
        for (int n = 0; n < TN; n++)
          b2[n] = *reinterpret_cast<const float4 *>(w2 + ((size_t)c2 * p.post_wrows + (size_t)pr * 2) * wstride + (size_t)n * 32 * 4);
#pragma unroll
        for (int m = 0; m < TM; m++)
#pragma unroll
          for (int n = 0; n < TN; n++) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[m].x, b2[n].x, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[m].y, b2[n].y, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[m].z, b2[n].z, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[m].w, b2[n].w, acc[m][n], 0, 0, 0);
          }
      }
    bias_ptr = p.post_bias;
    relu_flag = p.post_relu;
  }

  if (SP == 1 && p.mfma_count && lane == 0)
    atomicAdd(p.mfma_count + (wg & (kMfmaCountSlots - 1)), (unsigned long long)n_exec);

  // ---- epilogue: bias, ReLU, optional 2x2x2 pool, store channels-last ----
  float out_max = 0.f;  // (ConvArgs::out_amax)
  const int So = p.pool ? S / 2 : S;
  float *out_b = p.out + (size_t)b * So * So * So * p.out_cs + p.out_c0;
  const int ncx = S / 2;  // cells per axis of the whole grid
#pragma unroll
  for (int m = 0; m < TM; m++) {
#pragma unroll
    for (int half = 0; half < 2; half++) {
      int cx, cy, cz;  // accumulator rows: bit2 = lane>>5, bit4 = reg>>3
      if (!cell_of(wm * TM + m, kh + 2 * half, cx, cy, cz)) continue;
      const int gcx = tx * p.tcx + cx, gcy = ty * p.tcy + cy, gcz = tz * p.tcz + cz;
      if (gcx >= ncx || gcy >= ncx || gcz >= ncx) continue;
#pragma unroll
      for (int n = 0; n < TN; n++) {
        const int ch = n_base + n * 32 + row;
        if (ch >= p.cout) continue;
        const float bias = bias_ptr[ch];
        if (p.pool == 1 && !p.argmax_out) {
          // forward-only max pool: rounding and ReLU are monotonic, so max_r relu(fl(acc_r + b)) = relu(fl(max_r acc_r + b))
          // exactly -- 6 VALU instructions per cell instead of 37 (the arg-max bookkeeping is for the gradient pass only)
          const f32x16 &ac = acc[m][n];
          const int h8 = half * 8;
          float mx = fmaxf(fmaxf(fmaxf(ac[h8 + 0], ac[h8 + 1]), fmaxf(ac[h8 + 2], ac[h8 + 3])),
                           fmaxf(fmaxf(ac[h8 + 4], ac[h8 + 5]), fmaxf(ac[h8 + 6], ac[h8 + 7])));
          mx = mx + bias;
          if (relu_flag) mx = fmaxf(mx, 0.f);
          out_b[(((size_t)gcx * So + gcy) * So + gcz) * p.out_cs + ch] = mx;
          continue;
        }
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
          float t = acc[m][n][half * 8 + r] + bias;
          v[r] = relu_flag ? fmaxf(t, 0.f) : t;
        }
        if (p.pool == 1) {
          float mx = v[0];
          int am = 0;
#pragma unroll
          for (int r = 1; r < 8; r++)
            if (v[r] > mx) {  // first maximum in (kd,kh,kw) scan order, like max_pool3d
              mx = v[r];
              am = r;
            }
          const size_t o = (((size_t)gcx * So + gcy) * So + gcz) * p.out_cs + ch;
          out_b[o] = mx;
          if (p.argmax_out) p.argmax_out[(size_t)b * So * So * So * p.out_cs + p.out_c0 + o] = (unsigned char)am;
        } else if (p.pool == 2) {
          float s = v[0];
#pragma unroll
          for (int r = 1; r < 8; r++) s = s + v[r];  // r = x*4 + y*2 + z: avg_pool3d's (kd,kh,kw) order
          out_b[(((size_t)gcx * So + gcy) * So + gcz) * p.out_cs + ch] = s * 0.125f;
        } else {
          const float osc = p.out_scale ? p.out_scale[ch] : 1.0f;
          // gradient pass (ConvArgs::out_mask / out_amax): the ReLU of the layer this gradient belongs to and its per-pose
          // maximum, on the channels [out_mask_c0, out_mask_c1).  The activations and, when accumulating, the old values are
          // fetched ahead of the first store (a store may alias the next load as far as the compiler knows).
          // (The loads sit behind wave-uniform branches only, every lane with a valid address: a per-lane `cond ? load : c` is
          // compiled into a branch, a wait and a select per element.)
          const bool in_range = ch >= p.out_mask_c0 && ch < p.out_mask_c1;
          const int ch_m = in_range ? ch : p.out_mask_c0;
          const bool any_msk = p.out_mask && __builtin_amdgcn_ballot_w64(in_range) != 0ull;  // (wave-uniform)
          float a8[8], o8[8];
#pragma unroll
          for (int r = 0; r < 8; r++) {
            const int vx = 2 * gcx + (r >> 2), vy = 2 * gcy + ((r >> 1) & 1), vz = 2 * gcz + (r & 1);
            const size_t vox = ((size_t)vx * So + vy) * So + vz;
            a8[r] = 1.f, o8[r] = 0.f;
            if (any_msk) a8[r] = p.out_mask[((size_t)b * So * So * So + vox) * p.out_mask_cs + ch_m];
            if (p.accumulate) o8[r] = out_b[vox * p.out_cs + ch];
          }
          // (every fetched value is consumed before the first store is issued: the compiler otherwise sinks each load down
          // to its use, behind the previous store it may alias)
#pragma unroll
          for (int r = 0; r < 8; r++) {
            float val = p.out_scale ? v[r] * osc : v[r];
            if (p.accumulate) val = o8[r] + val;
            if (in_range) {
              val = a8[r] > 0.f ? val : 0.f;
              out_max = fmaxf(out_max, fabsf(val));
            }
            v[r] = val;
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < 8; r++) {
            const int vx = 2 * gcx + (r >> 2), vy = 2 * gcy + ((r >> 1) & 1), vz = 2 * gcz + (r & 1);
            out_b[(((size_t)vx * So + vy) * So + vz) * p.out_cs + ch] = v[r];
          }
        }
      }
    }
  }
  if (p.out_amax) {
    for (int o = 32; o; o >>= 1) out_max = fmaxf(out_max, __shfl_xor(out_max, o));
    // (a plain read first: the maximum settles after a few workgroups, and same-line atomics serialize in L2 -- ~9 ns each,
      // 0.25 ms per launch when every wave issues one; a stale read only costs an atomic that changes nothing)
      if (lane == 0 && __float_as_uint(out_max) > __hip_atomic_load(p.out_amax + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(p.out_amax + b, __float_as_uint(out_max));
  }
}

// ---------------------------------------------------------------------------------------------
// 16-output-channel variant for the Dense blocks (Cout = 16): v_mfma_f32_16x16x4_f32.
// A 32-wide tile would run half empty; here N = 16 exactly.  An M-tile is 16 voxels = two 2x2x2
// cells (row = cell*8 + x*4 + y*2 + z); the four k of one instruction are channel j of FOUR
// different quads (lane group l>>4 picks the quad), so one ds_read_b128 per lane still feeds four
// MFMAs.  TM independent accumulators per wave cover the 40-cycle dependent latency of the 16x16x4
// form.  Dense-block convs have eval-BN on their input (folded into the staging), ReLU, no pooling,
// and write at a channel offset of the concat buffer.
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TM, bool SKIP>
__global__ __launch_bounds__(256) void conv3d_mfma16_kernel(ConvArgs p) {
  constexpr int NTHREADS = 256;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: M-tile / cell arithmetic on the scalar unit
  const int kq = lane >> 4;   // which of the four quads of a step this lane feeds
  const int row = lane & 15;  // A row / B column

  const int tiles_per_pose = p.ntx * p.nty * p.ntz;
  const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int b = wg / tiles_per_pose;
  int t = wg - b * tiles_per_pose;
  const int tz = t % p.ntz;
  t /= p.ntz;
  const int ty = t % p.nty, tx = t / p.nty;

  const int halo = p.ksize == 3 ? 1 : 0;
  const int HX = 2 * p.tcx + 2 * halo, HY = 2 * p.tcy + 2 * halo, HZ = 2 * p.tcz + 2 * halo;
  const int HV = HX * HY * HZ;
  const int CC4 = p.cc4, CCs = p.ccs;
  const int taps = p.ksize == 3 ? 27 : 1;
  const int Q = taps * CC4;       // quads per chunk
  const int P4 = (Q + 3) >> 2;    // quad quartets per chunk

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *s_tile = smem;
  // [Q + 12] byte offset of every quad inside the halo tile (tap shift + channel quad); the entries behind the last
  // quad repeat it: the K loop reads one step ahead and its last step may be partly padding (zero weights)
  int *s_qoff = reinterpret_cast<int *>(smem + (size_t)HV * CCs);
  int *s_vox = s_qoff + Q + 12;  // [HV] float offset of every halo voxel's channel row inside the pose, -1 = padding
  for (int q = tid; q < Q + 12; q += NTHREADS) {
    const int qq = q < Q ? q : Q - 1;
    int tap = qq / CC4, c4 = qq - tap * CC4;
    if (p.korder) {  // channel-major, snake walk over the taps (ConvArgs::korder)
      c4 = qq / taps;
      tap = qq - c4 * taps;
      if (taps == 27) tap = conv_snake_tap(tap);
    }
    int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    s_qoff[q] = ((p.ksize == 3 ? ((dx * HY + dy) * HZ + dz) * CCs : 0) + c4 * 4) * 4;
  }

  const int NC = p.tcx * p.tcy * p.tcz;
  const int oz = row & 1, oy = (row >> 1) & 1, ox = (row >> 2) & 1, cell_in_mt = row >> 3;
  int baseA[TM];
#pragma unroll
  for (int m = 0; m < TM; m++) {
    int cell = (wm * TM + m) * 2 + cell_in_mt;
    if (cell >= NC) cell = 0;
    int cz = cell % p.tcz, cy = (cell / p.tcz) % p.tcy, cx = cell / (p.tcz * p.tcy);
    baseA[m] = (((2 * cx + ox) * HY + (2 * cy + oy)) * HZ + (2 * cz + oz)) * CCs * 4;  // bytes
  }
  f32x4 acc[TM];
#pragma unroll
  for (int m = 0; m < TM; m++) acc[m] = {0.f, 0.f, 0.f, 0.f};

  const int S = p.S;
  const int x0 = tx * 2 * p.tcx - halo, y0 = ty * 2 * p.tcy - halo, z0 = tz * 2 * p.tcz - halo;
  const unsigned inv_hz = ((1u << 20) + HZ - 1) / HZ, inv_hy = ((1u << 20) + HY - 1) / HY;  // (see conv3d_mfma_kernel)
  for (int hv = tid; hv < HV; hv += NTHREADS) {
    const int t1 = (int)(((unsigned)hv * inv_hz) >> 20), hz = hv - t1 * HZ;
    const int hx = (int)(((unsigned)t1 * inv_hy) >> 20), hy = t1 - hx * HY;
    const int x = x0 + hx, y = y0 + hy, z = z0 + hz;
    const bool in = (unsigned)x < (unsigned)S && (unsigned)y < (unsigned)S && (unsigned)z < (unsigned)S;
    s_vox[hv] = in ? ((x * S + y) * S + z) * p.in_cs : -1;
    if (!in)  // the zero padding is laid down once per workgroup; staging then touches the voxels inside the grid only
      for (int c = 0; c < CCs; c += 4) *reinterpret_cast<float4 *>(s_tile + hv * CCs + c) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float *in_b = p.in + (size_t)b * S * S * S * p.in_cs;
  const size_t wstride = (size_t)p.coutp * 4;  // 16 * 4 floats per quad row

  for (int chunk = 0; chunk < p.nchunks; chunk++) {
    __syncthreads();
    const int c_base = chunk * CC4 * 4;
    // voxel-major staging (see conv3d_mfma_kernel: every VALU instruction here is time taken from the MFMAs)
    const float *src_c = in_b + c_base;
    const int nq = min(CC4, p.cin4 - chunk * CC4);
    constexpr int U = 4;
    // MASK (in_mode 1: this launch is a transposed conv behind a ReLU): the gradient passes where the forward activation
    // was > 0 -- same voxel offsets in the activation tensor (in_act_cs == in_cs: both are strides of the conv's output buffer)
    const float *act_c = p.in_mode == 1 ? p.in_act + (size_t)b * S * S * S * p.in_act_cs + c_base : nullptr;
    auto stage = [&](auto has_bn, auto has_mask) {
      constexpr bool BN = decltype(has_bn)::value, MASK = decltype(has_mask)::value;
      for (int hv = tid; hv < HV; hv += NTHREADS) {
        const int off = s_vox[hv];
        if (off < 0) continue;
        float *dst = s_tile + hv * CCs;
        const float *src = src_c + off;
        int qb = 0;
        for (; qb + U <= nq; qb += U) {
          float4 val[U], act[U];
          BnQuad bn[U];
#pragma unroll
          for (int u = 0; u < U; u++) {
            val[u] = *reinterpret_cast<const float4 *>(src + (qb + u) * 4);
            if constexpr (MASK) act[u] = *reinterpret_cast<const float4 *>(act_c + off + (qb + u) * 4);
            if constexpr (BN) bn[u] = bn_load(p.bn_scale, p.bn_shift, c_base + (qb + u) * 4);
          }
#pragma unroll
          for (int u = 0; u < U; u++) {
            float4 x = val[u];
            if constexpr (MASK) {
              x.x = act[u].x > 0.f ? x.x : 0.f;
              x.y = act[u].y > 0.f ? x.y : 0.f;
              x.z = act[u].z > 0.f ? x.z : 0.f;
              x.w = act[u].w > 0.f ? x.w : 0.f;
            }
            if constexpr (BN) bn_apply(x, bn[u]);  // eval BatchNorm on the conv input; padding stays exactly 0
            *reinterpret_cast<float4 *>(dst + (qb + u) * 4) = x;
          }
        }
        if (qb < nq) {  // the tail of 1 .. U - 1 quads, its loads batched as well
          float4 val[U - 1], act[U - 1];
#pragma unroll
          for (int u = 0; u < U - 1; u++)
            if (qb + u < nq) {
              val[u] = *reinterpret_cast<const float4 *>(src + (qb + u) * 4);
              if constexpr (MASK) act[u] = *reinterpret_cast<const float4 *>(act_c + off + (qb + u) * 4);
            }
#pragma unroll
          for (int u = 0; u < U - 1; u++)
            if (qb + u < nq) {
              float4 x = val[u];
              if constexpr (MASK) {
                x.x = act[u].x > 0.f ? x.x : 0.f;
                x.y = act[u].y > 0.f ? x.y : 0.f;
                x.z = act[u].z > 0.f ? x.z : 0.f;
                x.w = act[u].w > 0.f ? x.w : 0.f;
              }
              if constexpr (BN) bn_apply(x, bn_load(p.bn_scale, p.bn_shift, c_base + (qb + u) * 4));
              *reinterpret_cast<float4 *>(dst + (qb + u) * 4) = x;
            }
        }
      }
    };
    if (p.in_mode == 1)
      stage(std::false_type{}, std::true_type{});  // (a transposed conv has no BatchNorm on its input)
    else if (p.bn_scale)
      stage(std::true_type{}, std::false_type{});
    else
      stage(std::false_type{}, std::false_type{});
    if (nq < CC4)  // partial last chunk: its missing quads still hold the previous chunk's channels
      for (int it = tid; it < HV * (CC4 - nq); it += NTHREADS) {
        const int hv = it / (CC4 - nq), c4 = nq + it - hv * (CC4 - nq);
        *reinterpret_cast<float4 *>(s_tile + hv * CCs + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    __syncthreads();

    // K loop over quad quartets, ping-pong operand sets (see conv3d_mfma_kernel); weights packed
    // [chunk][quartet][4][16][4] with one all-zero quartet behind the last (ConvArgs::wrows): pad quads multiply by
    // zero, the read-ahead past the end needs no branch.  The tile offset of a lane's quad is read one step ahead;
    // the weight row is affine in the step: uniform base in SGPRs + 32-bit lane offset.
    const char *wbase = reinterpret_cast<const char *>(p.wp) + (size_t)chunk * p.wrows * wstride * 4;
    const unsigned wlane = (unsigned)row * 16u + (unsigned)kq * (unsigned)wstride * 4u;
    const unsigned wstep = 4u * (unsigned)wstride * 4u;  // bytes per quartet of rows
    const int *lp = s_qoff + kq;
    int e_next = lp[0];
    auto load_step = [&](int pr, float4 *aa, float4 &ww) {
      const int e = e_next;
#pragma unroll
      for (int m = 0; m < TM; m++) aa[m] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_tile) + baseA[m] + e);
      ww = *reinterpret_cast<const float4 *>(wbase + (wlane + (unsigned)pr * wstep));
      e_next = lp[4 * pr + 4];
    };
    auto mfma_step = [&](const float4 *aa, const float4 &ww) {
      if constexpr (SKIP) {
        // per-MFMA zero test (see conv3d_mfma_kernel): one instruction = one channel at 16 voxels x four neighbouring
        // taps (korder 1); the lane masks go straight into SGPR pairs, the branches are scalar
        unsigned long long live[TM][4];
#pragma unroll
        for (int m = 0; m < TM; m++) {
          const float ac[4] = {aa[m].x, aa[m].y, aa[m].z, aa[m].w};
#pragma unroll
          for (int j = 0; j < 4; j++) asm("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(live[m][j]) : "v"(ac[j]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int m = 0; m < TM; m++) {
            if (live[m][j] == 0ull) continue;
            const float a = j == 0 ? aa[m].x : j == 1 ? aa[m].y : j == 2 ? aa[m].z : aa[m].w;
            const float wc = j == 0 ? ww.x : j == 1 ? ww.y : j == 2 ? ww.z : ww.w;
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wc, acc[m], 0, 0, 0);
          }
        return;
      }
#pragma unroll
      for (int m = 0; m < TM; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[m].x, ww.x, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < TM; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[m].y, ww.y, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < TM; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[m].z, ww.z, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < TM; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[m].w, ww.w, acc[m], 0, 0, 0);
    };
    float4 a0[TM], a1[TM], w0, w1;
    load_step(0, a0, w0);
    int pr = 0;
    for (; pr + 1 < P4; pr += 2) {
      load_step(pr + 1, a1, w1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(a0, w0);
      __builtin_amdgcn_sched_barrier(0);
      load_step(pr + 2, a0, w0);  // (past the end: the zero quartet, unused)
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(a1, w1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (P4 & 1) mfma_step(a0, w0);
  }

  // epilogue: accumulator row = 4 * (lane >> 4) + reg, column = lane & 15
  float *out_b = p.out + (size_t)b * S * S * S * p.out_cs + p.out_c0;
  const int ncx = S / 2;
  const int ch = row;
  if (ch < p.cout) {
    const float bias = p.bias[ch];
#pragma unroll
    for (int m = 0; m < TM; m++) {
      const int cell = (wm * TM + m) * 2 + (kq >> 1);
      if (cell >= NC) continue;
      const int cz = cell % p.tcz, cy = (cell / p.tcz) % p.tcy, cx = cell / (p.tcz * p.tcy);
      const int gcx = tx * p.tcx + cx, gcy = ty * p.tcy + cy, gcz = tz * p.tcz + cz;
      if (gcx >= ncx || gcy >= ncx || gcz >= ncx) continue;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int vx = 2 * gcx + (kq & 1), vy = 2 * gcy + (r >> 1), vz = 2 * gcz + (r & 1);
        float v = acc[m][r] + bias;
        if (p.bias_tab) {  // BatchNorm shift of the input, by border class of the output voxel (ConvArgs::bias_tab)
          const int cls = ((vx == 0 ? 0 : vx == S - 1 ? 2 : 1) * 3 + (vy == 0 ? 0 : vy == S - 1 ? 2 : 1)) * 3 +
                          (vz == 0 ? 0 : vz == S - 1 ? 2 : 1);
          v = acc[m][r] + p.bias_tab[cls * 16 + ch];
        }
        if (p.relu) v = fmaxf(v, 0.f);
        out_b[(((size_t)vx * S + vy) * S + vz) * p.out_cs + ch] = v;
      }
    }
  }
}

size_t conv_lds_bytes(const ConvArgs &p) {
  const int halo = p.ksize == 3 ? 1 : 0;
  const size_t HV = (size_t)(2 * p.tcx + 2 * halo) * (2 * p.tcy + 2 * halo) * (2 * p.tcz + 2 * halo);
  const int Q = (p.ksize == 3 ? 27 : 1) * p.cc4;
  // (the N = 16 kernel keeps a [Q + 12] offset table where this one has its [Q + 5] int2 list: sized for the larger)
  const size_t main_bytes = HV * p.ccs * sizeof(float) + (size_t)(2 * (Q + 6) + p.nchunks * p.cc4 + (p.in_mode == 3 ? 2 : 1) * HV + 1 + 2 * 27) * sizeof(int);
  const size_t mid_bytes = p.post_w ? (size_t)p.post_rows * (p.coutp + 4) * sizeof(float) : 0;
  return main_bytes > mid_bytes ? main_bytes : mid_bytes;
}

template <int WM, int WN, int TM, int TN, int SP, bool MTX> static void launch_one(const ConvArgs &p, int B, hipStream_t s) {
  const int ngroups = (p.coutp / 32 + WN * TN - 1) / (WN * TN);
  dim3 grid(B * p.ntx * p.nty * p.ntz, ngroups), block(64 * WM * WN);
  const size_t lds = conv_lds_bytes(p);
  ensure_max_lds(reinterpret_cast<const void *>(conv3d_mfma_kernel<WM, WN, TM, TN, SP, MTX>), 160 * 1024);
  hipLaunchKernelGGL((conv3d_mfma_kernel<WM, WN, TM, TN, SP, MTX>), grid, block, lds, s, p);
}

// MTX (ConvArgs::mt_x, M-tiles stacked along x) is compiled only where the engine asks for it (the 4x1/2x1 first-conv
// tile): as a run-time branch it cost the TM = 7 kernel 8 VGPRs, i.e. its second wave per SIMD
template <int WM, int WN, int TM, int TN, bool MTX_OK = false> static void launch_cfg(const ConvArgs &p, int B, hipStream_t s) {
  if (p.mt_x && !MTX_OK) throw std::runtime_error("launch_conv: mt_x is not compiled for this tile configuration");
  if (MTX_OK && p.mt_x) {
    if (p.sparse)
      launch_one<WM, WN, TM, TN, 1, MTX_OK>(p, B, s);
    else
      launch_one<WM, WN, TM, TN, 0, MTX_OK>(p, B, s);
    return;
  }
  if (p.sparse)
    launch_one<WM, WN, TM, TN, 1, false>(p, B, s);
  else
    launch_one<WM, WN, TM, TN, 0, false>(p, B, s);
}

void launch_conv(const ConvArgs &p, int cfg, int B, hipStream_t s) {
  switch (cfg) {
    case CONV_CFG_4x1_2x1: launch_cfg<4, 1, 2, 1, true>(p, B, s); break;
    case CONV_CFG_1x4_7x1: launch_cfg<1, 4, 7, 1>(p, B, s); break;
    case CONV_CFG_4x1_2x3: launch_cfg<4, 1, 2, 3>(p, B, s); break;
    case CONV_CFG_4x1_1x3: launch_cfg<4, 1, 1, 3>(p, B, s); break;
    case CONV_CFG_2x2_3x1: launch_cfg<2, 2, 3, 1>(p, B, s); break;
    case CONV_CFG_4x1_1x1: launch_cfg<4, 1, 1, 1>(p, B, s); break;
    case CONV_CFG_4x1_1x5: launch_cfg<4, 1, 1, 5>(p, B, s); break;
    case CONV_CFG_N16_TM4:
    case CONV_CFG_N16_TM3:
    case CONV_CFG_N16_TM2:
    case CONV_CFG_N16_TM1: {
      dim3 grid(B * p.ntx * p.nty * p.ntz), block(256);
      const size_t lds = conv_lds_bytes(p);
      const bool skip = p.sparse == 2;
      auto go = [&](auto kern) {
        ensure_max_lds(reinterpret_cast<const void *>(kern), 160 * 1024);
        hipLaunchKernelGGL(kern, grid, block, lds, s, p);
      };
      if (cfg == CONV_CFG_N16_TM4)
        skip ? go(conv3d_mfma16_kernel<4, true>) : go(conv3d_mfma16_kernel<4, false>);
      else if (cfg == CONV_CFG_N16_TM3)
        skip ? go(conv3d_mfma16_kernel<3, true>) : go(conv3d_mfma16_kernel<3, false>);
      else if (cfg == CONV_CFG_N16_TM2)
        skip ? go(conv3d_mfma16_kernel<2, true>) : go(conv3d_mfma16_kernel<2, false>);
      else
        skip ? go(conv3d_mfma16_kernel<1, true>) : go(conv3d_mfma16_kernel<1, false>);
      break;
    }
    default: break;
  }
}

void conv_cfg_shape(int cfg, int *wm, int *wn, int *tm, int *tn) {
  switch (cfg) {
    case CONV_CFG_4x1_2x1: *wm = 4, *wn = 1, *tm = 2, *tn = 1; break;
    case CONV_CFG_4x1_2x3: *wm = 4, *wn = 1, *tm = 2, *tn = 3; break;
    case CONV_CFG_4x1_1x3: *wm = 4, *wn = 1, *tm = 1, *tn = 3; break;
    case CONV_CFG_2x2_3x1: *wm = 2, *wn = 2, *tm = 3, *tn = 1; break;
    case CONV_CFG_4x1_1x1: *wm = 4, *wn = 1, *tm = 1, *tn = 1; break;
    case CONV_CFG_N16_TM1: *wm = 4, *wn = 1, *tm = 1, *tn = 1; break;
    case CONV_CFG_N16_TM2: *wm = 4, *wn = 1, *tm = 2, *tn = 1; break;
    case CONV_CFG_4x1_1x5: *wm = 4, *wn = 1, *tm = 1, *tn = 5; break;
    case CONV_CFG_N16_TM4: *wm = 4, *wn = 1, *tm = 4, *tn = 1; break;
    case CONV_CFG_N16_TM3: *wm = 4, *wn = 1, *tm = 3, *tn = 1; break;
    default: *wm = 1, *wn = 4, *tm = 7, *tn = 1; break;
  }
}

// ---------------------------------------------------------------------------------------------
// Small helpers of the layer program
// ---------------------------------------------------------------------------------------------
// reference-layout grid [B][C][N][N][N] -> 2x2x2 pooled channels-last [B][N/2]^3[Cp]
// (used only by mi_model_forward_grids, the CNN-on-given-grids test entry point)
__global__ void pool_input_ncdhw_kernel(const float *in, float *out, int C, int Cp, int N, int mode, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int S = N / 2;
  int c = i % Cp;
  long r = i / Cp;
  int z = r % S;
  r /= S;
  int y = r % S;
  r /= S;
  int x = r % S;
  long b = r / S;
  float res = 0.f;
  if (c < C) {
    const float *src = in + (((size_t)b * C + c) * N) * N * N;
    float acc = 0.f;
    bool first = true;
    for (int dx = 0; dx < 2; dx++)
      for (int dy = 0; dy < 2; dy++)
        for (int dz = 0; dz < 2; dz++) {
          float v = src[((size_t)(2 * x + dx) * N + (2 * y + dy)) * N + (2 * z + dz)];
          if (mode == 1)
            acc = first ? v : fmaxf(acc, v);
          else
            acc = first ? v : acc + v;
          first = false;
        }
    res = mode == 1 ? acc : acc * 0.125f;
  }
  out[i] = res;
}

// How sparse is a channels-last activation tensor [B][S]^3[cs]?  out[0] += (2x2x2 cell, channel) groups that are
// all-zero, out[1] += groups looked at.  The engine asks once per ReLU'd conv layer (first launch of >= 32 poses) and
// keeps the per-MFMA zero test on that layer only if it pays (ConvArgs::sparse 2 vs 3).
__global__ __launch_bounds__(256) void zero_cell_probe_kernel(const float *in, long n_items, int S, int cs, int C4, unsigned *out) {
  const long it = (long)blockIdx.x * 256 + threadIdx.x;
  unsigned zeros = 0, groups = 0;
  if (it < n_items) {
    const int q = (int)(it % C4);
    long cell = it / C4;
    const int H = S / 2;
    const int cz = (int)(cell % H);
    cell /= H;
    const int cy = (int)(cell % H);
    cell /= H;
    const int cx = (int)(cell % H);
    const long b = cell / H;
    unsigned o[4] = {0u, 0u, 0u, 0u};
    for (int r = 0; r < 8; r++) {
      const int x = 2 * cx + (r >> 2), y = 2 * cy + ((r >> 1) & 1), z = 2 * cz + (r & 1);
      const float4 v = *reinterpret_cast<const float4 *>(in + (((b * S + x) * S + y) * S + z) * cs + q * 4);
      o[0] |= __float_as_uint(v.x), o[1] |= __float_as_uint(v.y), o[2] |= __float_as_uint(v.z), o[3] |= __float_as_uint(v.w);
    }
    for (int j = 0; j < 4; j++) zeros += o[j] == 0u;
    groups = 4;
  }
  for (int d = 32; d; d >>= 1) zeros += __shfl_down(zeros, d), groups += __shfl_down(groups, d);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out, zeros);
    atomicAdd(out + 1, groups);
  }
}

void launch_zero_cell_probe(const float *in, int B, int C, int cs, int S, unsigned *out, hipStream_t s) {
  const int C4 = C / 4;  // whole quads only (padding channels are zero by construction)
  const long n = (long)B * (S / 2) * (S / 2) * (S / 2) * C4;
  if (n <= 0) return;
  hipLaunchKernelGGL(zero_cell_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, S, cs, C4, out);
}

void launch_pool_input(const float *in, float *out, int B, int C, int Cp, int N, int mode, hipStream_t s) {
  const int S = N / 2;
  long total = (long)B * S * S * S * Cp;
  hipLaunchKernelGGL(pool_input_ncdhw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, C,
                     Cp, N, mode, total);
}

// channels-last 2x2x2 pool: in [B][S]^3[in_cs] (first C channels) -> out [B][S/2]^3[out_cs]
__global__ void pool_cl_kernel(const float *in, float *out, int C, int in_cs, int out_cs, int S, int mode,
                               long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int So = S / 2;
  int c = i % C;
  long r = i / C;
  int z = r % So;
  r /= So;
  int y = r % So;
  r /= So;
  int x = r % So;
  long b = r / So;
  float acc = 0.f;
  bool first = true;
  for (int dx = 0; dx < 2; dx++)
    for (int dy = 0; dy < 2; dy++)
      for (int dz = 0; dz < 2; dz++) {
        float v = in[((((size_t)b * S + 2 * x + dx) * S + 2 * y + dy) * S + 2 * z + dz) * in_cs + c];
        if (mode == 1)
          acc = first ? v : fmaxf(acc, v);
        else
          acc = first ? v : acc + v;
        first = false;
      }
  out[((((size_t)b * So + x) * So + y) * So + z) * out_cs + c] = mode == 1 ? acc : acc * 0.125f;
}

void launch_pool_cl(const float *in, float *out, int B, int C, int in_cs, int out_cs, int S, int mode,
                    hipStream_t s) {
  const int So = S / 2;
  long total = (long)B * So * So * So * C;
  hipLaunchKernelGGL(pool_cl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, C, in_cs,
                     out_cs, S, mode, total);
}

// global max over space: in [B][S]^3[in_cs] -> out [B][out_cs].  A workgroup takes 32 channels of a pose, eight threads per
// channel share the voxels (a maximum does not depend on the order it is taken in); one thread per channel walking all
// S^3 voxels took 41 us at B = 1 -- on the critical path of every Dense call.
__global__ __launch_bounds__(256) void gmax_kernel(const float *in, float *out, int C, int in_cs, int out_cs, int S3) {
  __shared__ float red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int c = blockIdx.y * 32 + (tid & 31), part = tid >> 5;
  const float *src = in + (size_t)b * S3 * in_cs + (c < C ? c : 0);
  float m = src[(size_t)(part < S3 ? part : 0) * in_cs];
  for (int v = part + 8; v < S3; v += 8) m = fmaxf(m, src[(size_t)v * in_cs]);
  red[tid] = m;
  __syncthreads();
  if (part == 0 && c < C) {
#pragma unroll
    for (int k = 1; k < 8; k++) m = fmaxf(m, red[tid + 32 * k]);
    out[(size_t)b * out_cs + c] = m;
  }
}

void launch_gmax(const float *in, float *out, int B, int C, int in_cs, int out_cs, int S, hipStream_t s) {
  hipLaunchKernelGGL(gmax_kernel, dim3(B, (C + 31) / 32), dim3(256), 0, s, in, out, C, in_cs, out_cs, S * S * S);
}

// max_pool3d(kernel = whole grid) backward: the gradient goes to the first maximum in (x, y, z) scan order.
// One workgroup of 1,024 threads per pose: four voxel slices x 256 channels.  A thread scans its slice (independent loads, the
// compiler batches them), the slices' candidates are combined in scan order -- a later slice wins only with a strictly larger
// value, which is the serial scan's "first maximum" -- and every thread writes its slice of the gradient.  (As one thread per
// channel walking all S^3 voxels this kernel was 56 us of a B = 1 gradient call of a Dense model: 216 dependent steps.)
__global__ __launch_bounds__(1024) void gmax_backward_kernel(const float *act, const float *g_out, float *g_in, int C, int in_cs, int out_cs,
                                                             int S3) {
  constexpr int kSlices = 4, kCh = 256;
  const int b = blockIdx.x;
  const int sl = threadIdx.x / kCh, cl = threadIdx.x % kCh;
  const int per = (S3 + kSlices - 1) / kSlices;
  const int v0 = sl * per, v1 = min(S3, v0 + per);
  __shared__ float s_m[kSlices][kCh];
  __shared__ int s_am[kSlices][kCh];
  for (int c0 = 0; c0 < C; c0 += kCh) {
    const int c = c0 + cl;
    float m = 0.f;
    int am = -1;  // (an empty slice has no candidate)
    if (c < C && v0 < v1) {
      const float *src = act + (size_t)b * S3 * in_cs + c;
      // (slice 0 starts from its first voxel like the serial scan -- a NaN there sticks in both; a later slice starts from
      // "no candidate", so that a NaN inside it is skipped as the serial scan skips it)
      int vb = v0;
      if (sl == 0) m = src[0], am = 0, vb = 1;
      else m = -__builtin_inff();
#pragma unroll 8
      for (int v = vb; v < v1; v++) {
        const float t = src[(size_t)v * in_cs];
        if (t > m) m = t, am = v;
      }
    }
    s_m[sl][cl] = m;
    s_am[sl][cl] = am;
    __syncthreads();
    if (c < C) {
      float bm = s_m[0][cl];
      int bam = s_am[0][cl];
#pragma unroll
      for (int k = 1; k < kSlices; k++) {
        const float t = s_m[k][cl];
        const int ta = s_am[k][cl];
        if (ta >= 0 && (bam < 0 || t > bm)) bm = t, bam = ta;
      }
      float *dst = g_in + (size_t)b * S3 * in_cs + c;
      const float g = g_out[(size_t)b * out_cs + c];
      for (int v = v0; v < v1; v++) dst[(size_t)v * in_cs] = v == bam ? g : 0.f;
    }
    __syncthreads();
  }
}

void launch_gmax_backward(const float *act, const float *g_out, float *g_in, int B, int C, int in_cs, int out_cs,
                          int S, hipStream_t s) {
  hipLaunchKernelGGL(gmax_backward_kernel, dim3(B), dim3(1024), 0, s, act, g_out, g_in, C, in_cs, out_cs, S * S * S);
}

// ---------------------------------------------------------------------------------------------
// FC heads + score post-processing
// ---------------------------------------------------------------------------------------------
// in [B][n_in] (channels-last flatten), w [3][n_in], bias [3] -> logits/affinity, then
// TorchModel::forward's post-processing (gninasrc/lib/torch_model.cpp:188-195):
//   logp = log_softmax(z); pose = softmax(logp)[1] (or logp[1] if skip_softmax);
//   loss = cross_entropy(logp, 1) (or -log(logp[1]) if apply_logistic_loss)
__global__ __launch_bounds__(256) void fc_heads_kernel(const float *in, const float *w, const float *bias, int n_in,
                                                       int skip_softmax, int logistic_loss, float *pose, float *aff,
                                                       float *loss, float *raw3) {
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const float4 *x = reinterpret_cast<const float4 *>(in + (size_t)b * n_in);
  const float4 *w0 = reinterpret_cast<const float4 *>(w);
  const float4 *w1 = reinterpret_cast<const float4 *>(w + n_in);
  const float4 *w2 = reinterpret_cast<const float4 *>(w + 2 * (size_t)n_in);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const int n4 = n_in / 4;
  // (eight iterations' operands are requested before the first is used: one round trip per eight instead of one per iteration --
  // with 27 iterations for Default2017's 27,648 inputs this kernel was 10 us of a per-pose call; the fmas keep their order)
  constexpr int kAhead = 8;
  for (int i0 = tid; i0 < n4; i0 += 256 * kAhead) {
    float4 xv[kAhead], a[kAhead], bb[kAhead], c[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; k++) {
      const int i = i0 + k * 256;
      if (i < n4) xv[k] = x[i], a[k] = w0[i], bb[k] = w1[i], c[k] = w2[i];
    }
#pragma unroll
    for (int k = 0; k < kAhead; k++) {
      if (i0 + k * 256 >= n4) break;
      s0 = fmaf(xv[k].x, a[k].x, s0); s0 = fmaf(xv[k].y, a[k].y, s0); s0 = fmaf(xv[k].z, a[k].z, s0); s0 = fmaf(xv[k].w, a[k].w, s0);
      s1 = fmaf(xv[k].x, bb[k].x, s1); s1 = fmaf(xv[k].y, bb[k].y, s1); s1 = fmaf(xv[k].z, bb[k].z, s1); s1 = fmaf(xv[k].w, bb[k].w, s1);
      s2 = fmaf(xv[k].x, c[k].x, s2); s2 = fmaf(xv[k].y, c[k].y, s2); s2 = fmaf(xv[k].z, c[k].z, s2); s2 = fmaf(xv[k].w, c[k].w, s2);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_down(s0, off);
    s1 += __shfl_down(s1, off);
    s2 += __shfl_down(s2, off);
  }
  __shared__ float red[3][4];
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = s0;
    red[1][tid >> 6] = s1;
    red[2][tid >> 6] = s2;
  }
  __syncthreads();
  if (tid == 0) {
    float z0 = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) + bias[0];
    float z1 = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) + bias[1];
    float a = ((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) + bias[2];
    // log_softmax (what the TorchScript module returns)
    float m = fmaxf(z0, z1);
    float lse = logf(expf(z0 - m) + expf(z1 - m));
    float lp0 = (z0 - m) - lse, lp1 = (z1 - m) - lse;
    // softmax of the log-probabilities (torch_model.cpp:189)
    float m2 = fmaxf(lp0, lp1);
    float e0 = expf(lp0 - m2), e1 = expf(lp1 - m2);
    float ps = skip_softmax ? lp1 : e1 / (e0 + e1);
    // cross_entropy(logp, label 1) = -log_softmax(logp)[1]   (torch_model.cpp:195)
    float ls = logistic_loss ? -logf(lp1) : -((lp1 - m2) - logf(e0 + e1));
    pose[b] = ps;
    aff[b] = a;
    loss[b] = ls;
    if (raw3) {
      raw3[3 * b + 0] = lp0;
      raw3[3 * b + 1] = lp1;
      raw3[3 * b + 2] = a;
    }
  }
}

// Global max pool + heads in one launch for per-pose calls (a Dense model ends "whole-grid max pool -> three heads": two
// dependent launches of ~9 and ~4 us on the critical path of every call).  One workgroup of 1,024 threads per pose: four voxel
// slices x 256 channels take the maxima (order-free), the first 256 threads then run fc_heads_kernel's arithmetic on them from
// LDS -- the same partial sums in the same order, the same bits.  The pooled activations are written out as gmax_kernel does.
__global__ __launch_bounds__(1024) void gmax_heads_kernel(const float *in, float *gmax_out, int C, int in_cs, int out_cs, int S3,
                                                          const float *w, const float *bias, int skip_softmax, int logistic_loss,
                                                          float *pose, float *aff, float *loss) {
  constexpr int kMaxC = 1024;
  __shared__ __attribute__((aligned(16))) float s_x[kMaxC];
  __shared__ float s_part[4][256];
  __shared__ float red[3][4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int part = tid >> 8, cl = tid & 255;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + cl;
    const float *src = in + (size_t)b * S3 * in_cs + (c < C ? c : 0);
    float m = src[(size_t)(part < S3 ? part : 0) * in_cs];
#pragma unroll 8
    for (int v = part + 4; v < S3; v += 4) m = fmaxf(m, src[(size_t)v * in_cs]);
    s_part[part][cl] = m;
    __syncthreads();
    if (part == 0 && c < C) {
      m = fmaxf(fmaxf(m, s_part[1][cl]), fmaxf(s_part[2][cl], s_part[3][cl]));
      s_x[c] = m;
      gmax_out[(size_t)b * out_cs + c] = m;
    }
    __syncthreads();
  }
  // ---- fc_heads_kernel on s_x (n_in = C), threads 0 .. 255 ----
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (tid < 256) {
    const float4 *x = reinterpret_cast<const float4 *>(s_x);
    const float4 *w0 = reinterpret_cast<const float4 *>(w);
    const float4 *w1 = reinterpret_cast<const float4 *>(w + C);
    const float4 *w2 = reinterpret_cast<const float4 *>(w + 2 * (size_t)C);
    const int n4 = C / 4;
    for (int i = tid; i < n4; i += 256) {
      float4 xv = x[i], a = w0[i], bb = w1[i], c = w2[i];
      s0 = fmaf(xv.x, a.x, s0); s0 = fmaf(xv.y, a.y, s0); s0 = fmaf(xv.z, a.z, s0); s0 = fmaf(xv.w, a.w, s0);
      s1 = fmaf(xv.x, bb.x, s1); s1 = fmaf(xv.y, bb.y, s1); s1 = fmaf(xv.z, bb.z, s1); s1 = fmaf(xv.w, bb.w, s1);
      s2 = fmaf(xv.x, c.x, s2); s2 = fmaf(xv.y, c.y, s2); s2 = fmaf(xv.z, c.z, s2); s2 = fmaf(xv.w, c.w, s2);
    }
    for (int off = 32; off > 0; off >>= 1) {
      s0 += __shfl_down(s0, off);
      s1 += __shfl_down(s1, off);
      s2 += __shfl_down(s2, off);
    }
    if ((tid & 63) == 0) {
      red[0][tid >> 6] = s0;
      red[1][tid >> 6] = s1;
      red[2][tid >> 6] = s2;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float z0 = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) + bias[0];
    float z1 = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) + bias[1];
    float a = ((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) + bias[2];
    float m = fmaxf(z0, z1);
    float lse = logf(expf(z0 - m) + expf(z1 - m));
    float lp0 = (z0 - m) - lse, lp1 = (z1 - m) - lse;
    float m2 = fmaxf(lp0, lp1);
    float e0 = expf(lp0 - m2), e1 = expf(lp1 - m2);
    float ps = skip_softmax ? lp1 : e1 / (e0 + e1);
    float ls = logistic_loss ? -logf(lp1) : -((lp1 - m2) - logf(e0 + e1));
    pose[b] = ps;
    aff[b] = a;
    loss[b] = ls;
  }
}

bool gmax_heads_covers(int C) { return C % 4 == 0 && C <= 1024; }
void launch_gmax_heads(const float *in, float *gmax_out, int B, int C, int in_cs, int out_cs, int S, const float *w, const float *bias,
                       int skip_softmax, int logistic_loss, float *pose, float *aff, float *loss, hipStream_t s) {
  hipLaunchKernelGGL(gmax_heads_kernel, dim3(B), dim3(1024), 0, s, in, gmax_out, C, in_cs, out_cs, S * S * S, w, bias, skip_softmax,
                     logistic_loss, pose, aff, loss);
}

void launch_fc_heads(const float *in, const float *w, const float *bias, int n_in, int skip_softmax,
                     int logistic_loss, float *pose, float *aff, float *loss, float *raw3, int B, hipStream_t s) {
  hipLaunchKernelGGL(fc_heads_kernel, dim3(B), dim3(256), 0, s, in, w, bias, n_in, skip_softmax, logistic_loss,
                     pose, aff, loss, raw3);
}

// ---- the Overlap toy model (test/gnina/data/overlap*.pt, the model of the reference's test_min.py) -----------
// grid [B][2][N3] (reference layout): ave = mean_v rec[v] * lig[v]; module output = [0, ave > 0 ? ave : 1e-20],
// affinity 0; with skip_softmax + apply_logistic_loss (torch_model.cpp:188-195): pose = ave0, loss = -log(ave0).
__global__ __launch_bounds__(256) void overlap_forward_kernel(const float *grid, long N3, float *pose, float *aff,
                                                              float *loss, float *ave_out) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *rec = grid + (size_t)b * 2 * N3, *lig = rec + N3;
  float s = 0.f;
  for (long v = tid; v < N3; v += 256) s += rec[v] * lig[v];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  __shared__ float red[4];
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    const float ave = ((red[0] + red[1]) + (red[2] + red[3])) / (float)N3;
    const float ave0 = ave > 0.f ? ave : 9.9999999999999995e-21f;
    pose[b] = ave0;
    aff[b] = 0.f;
    loss[b] = -logf(ave0);
    if (ave_out) ave_out[b] = ave;
  }
}

void launch_overlap_forward(const float *grid, int B, long N3, float *pose, float *aff, float *loss, float *ave_out,
                            hipStream_t s) {
  hipLaunchKernelGGL(overlap_forward_kernel, dim3(B), dim3(256), 0, s, grid, N3, pose, aff, loss, ave_out);
}

// d loss / d grid = -(1 / ave) * d ave / d grid for ave > 0 (the where() branch has no gradient otherwise):
// gg[rec][v] = -(lig[v] / N3) / ave, gg[lig][v] = -(rec[v] / N3) / ave
__global__ void overlap_backward_kernel(const float *grid, const float *ave, long N3, float *gg, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long b = i / (2 * N3), r = i - b * 2 * N3;
  const long v = r >= N3 ? r - N3 : r;
  const float other = grid[(size_t)b * 2 * N3 + (r >= N3 ? v : N3 + v)];
  const float a = ave[b];
  gg[i] = a > 0.f ? -(other / (float)N3) / a : 0.f;
}

void launch_overlap_backward(const float *grid, const float *ave, int B, long N3, float *gg, hipStream_t s) {
  const long total = (long)B * 2 * N3;
  hipLaunchKernelGGL(overlap_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, grid, ave, N3, gg,
                     total);
}

// ---- gradient pass helpers --------------------------------------------------------------------
// d loss / d (fc input).  loss = cross_entropy(log_softmax(z), 1) (torch_model.cpp:192-195) so
// dz = softmax(z) - onehot(1) = (exp(lp0), exp(lp1) - 1); the affinity head does not enter the loss.
__global__ void fc_backward_kernel(const float *raw3, const float *w, int n_in, float *g_in, const float *mask, unsigned *amax) {
  const int b = blockIdx.y;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n_in) {
    const float dz0 = expf(raw3[3 * b + 0]), dz1 = expf(raw3[3 * b + 1]) - 1.0f;
    const float4 w0 = *reinterpret_cast<const float4 *>(w + i), w1 = *reinterpret_cast<const float4 *>(w + n_in + i);
    g.x = w0.x * dz0 + w1.x * dz1;
    g.y = w0.y * dz0 + w1.y * dz1;
    g.z = w0.z * dz0 + w1.z * dz1;
    g.w = w0.w * dz0 + w1.w * dz1;
    if (mask) {  // the ReLU behind the FC input, applied here (ConvArgs::out_mask)
      const float4 a = *reinterpret_cast<const float4 *>(mask + (size_t)b * n_in + i);
      g.x = a.x > 0.f ? g.x : 0.f;
      g.y = a.y > 0.f ? g.y : 0.f;
      g.z = a.z > 0.f ? g.z : 0.f;
      g.w = a.w > 0.f ? g.w : 0.f;
    }
    *reinterpret_cast<float4 *>(g_in + (size_t)b * n_in + i) = g;
  }
  if (amax) {  // (NaN: fmaxf drops it -- the consumer's own range check of what it stages catches it)
    __shared__ float s_m[4];
    float m = fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w)));
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
      if (__float_as_uint(m) > __hip_atomic_load(amax + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax + b, __float_as_uint(m));
    }
  }
}
void launch_fc_backward(const float *raw3, const float *w, int n_in, float *g_in, int B, hipStream_t s, const float *mask,
                        unsigned *amax) {
  hipLaunchKernelGGL(fc_backward_kernel, dim3((n_in / 4 + 255) / 256, B), dim3(256), 0, s, raw3, w, n_in, g_in, mask, amax);
}

// avg_pool3d(2) backward: every voxel of a cell receives 1/8 of the pooled gradient
__global__ void unpool_avg_kernel(const float *g_pooled, float *g_full, int C, int in_cs, int out_cs, int S,
                                  long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c = i % C;
  long r = i / C;
  int z = r % S;
  r /= S;
  int y = r % S;
  r /= S;
  int x = r % S;
  long b = r / S;
  const int Sh = S / 2;
  g_full[((((size_t)b * S + x) * S + y) * S + z) * out_cs + c] =
      g_pooled[((((size_t)b * Sh + (x >> 1)) * Sh + (y >> 1)) * Sh + (z >> 1)) * in_cs + c] * 0.125f;
}

void launch_unpool_avg(const float *g_pooled, float *g_full, int B, int C, int in_cs, int out_cs, int S,
                       hipStream_t s) {
  long total = (long)B * S * S * S * C;
  hipLaunchKernelGGL(unpool_avg_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g_pooled, g_full, C,
                     in_cs, out_cs, S, total);
}

// Gradient pass, ahead of a split-fp16 transposed conv that reads a channel slice of a concat buffer's gradient (Dense
// blocks: the slice has accumulated contributions of every later layer, so no producer could prepare it): in place,
// g = act > 0 ? g : 0 over the slice's C channels (a multiple of 4), and the per-pose maximum of |g| (ConvArgs::in_amax).
__global__ void grad_mask_amax_kernel(float *g, const float *act, int C4, int g_cs, int act_cs, long vox_per_pose, unsigned *amax) {
  const int b = blockIdx.y;
  const long items = vox_per_pose * C4;
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (long)gridDim.x * blockDim.x) {
    const long vox = i / C4;
    const int q = (int)(i - vox * C4);
    float4 *gp = reinterpret_cast<float4 *>(g + ((size_t)b * vox_per_pose + vox) * g_cs + q * 4);
    const float4 a = *reinterpret_cast<const float4 *>(act + ((size_t)b * vox_per_pose + vox) * act_cs + q * 4);
    float4 v = *gp;
    v.x = a.x > 0.f ? v.x : 0.f;
    v.y = a.y > 0.f ? v.y : 0.f;
    v.z = a.z > 0.f ? v.z : 0.f;
    v.w = a.w > 0.f ? v.w : 0.f;
    *gp = v;
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  __shared__ float s_m[4];
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    if (__float_as_uint(m) > __hip_atomic_load(amax + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax + b, __float_as_uint(m));
  }
}

void launch_grad_mask_amax(float *g, const float *act, int C, int g_cs, int act_cs, long vox_per_pose, int B, unsigned *amax,
                           hipStream_t s) {
  const long items = vox_per_pose * (C / 4);
  const long want = (items + 1023) / 1024;
  const unsigned blocks = (unsigned)(want < 64 ? want : 64);  // (four items per thread and more)
  hipLaunchKernelGGL(grad_mask_amax_kernel, dim3(blocks, B), dim3(256), 0, s, g, act, C / 4, g_cs, act_cs, vox_per_pose, amax);
}

// ensemble mean / variance over models (cnn_torch_scorer.cpp:177-191)
// (ovf_in / ovf_out: the call's range flag, copied along -- a host-output call has the results and the flag written straight
// into pinned host memory by this kernel instead of two copies behind it: ~10 us of a per-pose call)
__global__ void ensemble_reduce_kernel(const float *pose_m, const float *aff_m, const float *loss_m, int n_models,
                                       int B, float *pose, float *aff, float *loss, float *var, unsigned *ovf_in, unsigned *ovf_out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b == 0 && ovf_out) {  // (and the device flag is left cleared for the next call: no memset in front of that one)
    *ovf_out = ovf_in ? *ovf_in : 0u;
    if (ovf_in) *ovf_in = 0u;
  }
  if (b >= B) return;
  double sc = 0.0;   // reference accumulates the score in double (cnn_torch_scorer.cpp:117)
  float af = 0.f, ls = 0.f;
  for (int m = 0; m < n_models; m++) {
    sc += (double)pose_m[(size_t)m * B + b];
    af += aff_m[(size_t)m * B + b];
    ls += loss_m[(size_t)m * B + b];
  }
  af /= (float)n_models;
  ls /= (float)n_models;
  sc /= (double)n_models;
  float v = 0.f;
  if (n_models > 1) {
    float sum = 0.f;
    for (int m = 0; m < n_models; m++) {
      float d = af - aff_m[(size_t)m * B + b];
      sum += d * d;
    }
    v = sum / (float)n_models;
  }
  pose[b] = (float)sc;
  aff[b] = af;
  loss[b] = ls;
  if (var) var[b] = v;
}

// n dwords of zeros in ONE launch (hipMemsetAsync takes two fill kernels for sizes like 25 dwords -- 5 us each on the critical
// path of a per-pose gradient call)
__global__ void zero_u32_kernel(unsigned *p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
void launch_zero_u32(unsigned *p, size_t n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n);
}

// dst[i] = ((0 + src[0][i]) + src[1][i]) + ... : the per-model gradients of a gradient call on lanes, added in model order --
// the additions a one-stream call makes when every model accumulates into the one zeroed buffer
__global__ void sum_models_kernel(const float *src, int n_models, size_t n, float *dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  for (int m = 0; m < n_models; m++) acc = acc + src[(size_t)m * n + i];
  dst[i] = acc;
}
void launch_sum_models(const float *src, int n_models, size_t n, float *dst, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(sum_models_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, n_models, n, dst);
}

void launch_ensemble_reduce(const float *pose_m, const float *aff_m, const float *loss_m, int n_models, int B,
                            float *pose, float *aff, float *loss, float *var, hipStream_t s, unsigned *ovf_in, unsigned *ovf_out) {
  hipLaunchKernelGGL(ensemble_reduce_kernel, dim3((B + 127) / 128), dim3(128), 0, s, pose_m, aff_m, loss_m,
                     n_models, B, pose, aff, loss, var, ovf_in, ovf_out);
}

}  // namespace mig
