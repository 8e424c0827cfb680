// conv3d_h2.hip -- the split-fp16 forward convolution: fp32 tensors, fp32 accumulation, f16 MFMA rate.
//
// gfx950 runs v_mfma_f32_32x32x16_f16 at 16x the rate of v_mfma_f32_32x32x2_f32, and the product of two fp16 numbers is
// exact in the fp32 accumulator.  Every fp32 operand is written as  a = h + l  with  h = RN_f16(a),  l = RN_f16(a - h)
// (22-23 significant bits between them) and the product as
//     a * w  ~=  h_a * h_w  +  h_a * l_w  +  l_a * h_w                      (dropped: l_a * l_w, relative 2^-22)
// = three MFMAs where the fp32 kernels (conv3d.hip) issue eight for the same K = 16: 5.3x their matrix rate.  Weights are
// split once at model load, scaled by a per-layer power of two (exact; it keeps their low parts out of the fp16 subnormal
// range) that the epilogue takes out again.  Activations are split either by their PRODUCER -- the voxelizer's pooled
// store, a conv3d_h2_kernel epilogue: the tensor then lives in HBM as [pose][octet][x][y][z][h8 | l8] ("split format",
// conv3d.h ConvArgs::in_split) and the consumer stages it by LDS-DMA -- or while an fp32 tensor is staged (after the eval
// BatchNorm, which stays fp32): layer by layer the kernels here are interchangeable with the fp32 kernels of the same
// program, and a layer computes the same bits from either kind of input.  Measured against the float64 forward of the
// same operands the scores move by <= 1e-6 (tools/experiments/split_precision_probe.py; the parity bar is 1e-4) -- unlike
// the bf16 kernels (conv3d_bf16.hip: 4e-2), this IS a parity path.  An activation beyond the fp16 range raises the scorer's
// flag (h2_report_overflow) and the call is repeated on the fp32 kernels (engine.cpp).  Forward only (scoring calls and
// the forward half of gradient calls): the transposed convolutions of the backward pass stay on the fp32 kernels.
//
// Three kernels:
//   conv3d_h2_kernel     3x3x3 layers, 32 output channels per wave column: one octet per K chunk, planar halo tile in LDS,
//                        LDS-DMA staging of split-format inputs, the chunk's weights through LDS (LAB.md §3.9)
//   conv3d_h2_k1_kernel  1x1x1 layers (round 3's kernel): [voxel][octet][h | l] tile, fp32 input split while staging
//   conv3d_h2_16_kernel  the Dense blocks' 16-channel layers (v_mfma_f32_16x16x32_f16)
// Decomposition as in conv3d.hip: a workgroup owns a box of 2x2x2 cells of one pose and all (or a group of) output
// channels; an M-tile is 32 voxels = four cells, so ReLU + pooling stay register-local in the 32x32 accumulator layout.
// K runs over OCTETS (8 consecutive input channels at one tap); lanes 0-31 feed k = 0..7 of an instruction, lanes 32-63
// k = 8..15 (the next octet of the chunk, or -- conv3d_h2_kernel -- the next tap of the same octet).
#include "common.h"
#include "conv3d.h"

#include <algorithm>
#include <type_traits>

#ifndef MI_H2_OCC2
#define MI_H2_OCC2 4  // conv3d_h2_kernel, split-format input: waves per SIMD the register allocation aims at, TM <= 2 ...
#endif
#ifndef MI_H2_OCC3
#define MI_H2_OCC3 3  // ... and TM = 3
#endif
#ifndef MI_H2_EXPERIMENT
#define MI_H2_EXPERIMENT 0  // tools/h2_experiments.sh: 1 = no K loop, 2 = no staging loads, 3 = neither (timing only, wrong results)
#endif

namespace mig {

typedef float h2_f32x16 __attribute__((ext_vector_type(16)));
typedef float h2_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) h2_f32x4 *H2ConstQuadPtr;

// a -> (h, l), four channels at a time: h = RN_f16(a), l = RN_f16(a - h).  12 VALU instructions per quad: four clamps (an
// activation beyond the fp16 range -- |a| > 65504, not a value these networks produce -- stays a large finite number
// instead of inf - inf = NaN), two packed conversions, four v_fma_mix_f32 (a - h with h read as fp16: exact, one
// instruction instead of v_cvt_f32_f16 + v_sub_f32), two packed conversions.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float h2_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_f16(float a, float b) {
  const h2_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ void split4(const float4 &x, uint2 &h, uint2 &l) {
  const float c0 = __builtin_amdgcn_fmed3f(x.x, -65504.f, 65504.f), c1 = __builtin_amdgcn_fmed3f(x.y, -65504.f, 65504.f);
  const float c2 = __builtin_amdgcn_fmed3f(x.z, -65504.f, 65504.f), c3 = __builtin_amdgcn_fmed3f(x.w, -65504.f, 65504.f);
  h.x = pk_f16(c0, c1);
  h.y = pk_f16(c2, c3);
  float r0, r1, r2, r3;
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h.x), "v"(c0));
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h.x), "v"(c1));
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h.y), "v"(c2));
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h.y), "v"(c3));
  l.x = pk_f16(r0, r1);
  l.y = pk_f16(r2, r3);
}
// The same with the range check of a CONSUMER: amax = running maximum of |x| over everything this lane staged (two
// v_maximum3_f32 per quad; compared with 65504 once, at the end of the kernel -- h2_report_overflow).  Producers (the epilogues
// below, the voxelizer) check the values they write, NaN included; this catches what reaches a split-fp16 kernel from
// elsewhere: an eval BatchNorm applied while staging, the output of an fp32-MFMA layer.
__device__ __forceinline__ void split4(const float4 &x, uint2 &h, uint2 &l, float &amax) {
  // (v_maximum3_f32: the IEEE-754-2019 maximum -- a NaN operand gives NaN, which the final comparison reports as well)
  asm("v_maximum3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(x.x), "v"(x.y));
  asm("v_maximum3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(x.z), "v"(x.w));
  split4(x, h, l);
}
// one value: (h, l) packed as h | l << 16
__device__ __forceinline__ unsigned split1(float x) {
  const float c = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
  const unsigned hp = pk_f16(c, 0.f);
  float r;
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(c));
  return pk_f16(c, r);
}
// sticky range flag of the scorer (ConvArgs::h2_overflow): raised by any lane of the wave
__device__ __forceinline__ void h2_report_overflow(unsigned *flag, bool lane_overflow) {
  if (flag && __builtin_amdgcn_ballot_w64(lane_overflow) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// conv3d_h2_k1_kernel: the 1x1x1 convolutions (instantiated with K1 = true only: no halo, at most one voxel per thread, up
// to six octets of it per chunk; the 3x3x3 layers run on conv3d_h2_kernel below).  Single-buffered [voxel][octet][h | l]
// tile, fp32 input split while staging.
// BWD: a transposed 1x1x1 conv of the gradient pass behind a fused max pool (Dense transitions: `in`, `in_argmax` at S / 2,
// in_mode 2; ConvArgs::in_amax scaling, out_scale, out_mask / out_amax on a channel range -- see conv3d_h2_kernel's BWD)
template <int WM, int WN, int TM, int TN, bool MTX, bool SKIP, bool K1 = false, bool BWD = false>
__global__ __launch_bounds__(64 * WM * WN, (TM * TN <= 2 ? 3 : TM * TN <= 4 ? 2 : 1)) void conv3d_h2_k1_kernel(ConvArgs p) {
  static_assert(!BWD || K1, "gradient-pass variant: 1x1x1 convs only");
  constexpr int NTHREADS = 64 * WM * WN;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int kh = lane >> 5;
  const int row = lane & 31;

  const int tiles_per_pose = p.ntx * p.nty * p.ntz;
  const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int b = wg / tiles_per_pose;
  int t = wg - b * tiles_per_pose;
  const int tz = t % p.ntz;
  t /= p.ntz;
  const int ty = t % p.nty, tx = t / p.nty;
  const int n_base = (blockIdx.y * WN + wn) * TN * 32;

  const int halo = p.ksize == 3 ? 1 : 0;
  const int HX = 2 * p.tcx + 2 * halo, HY = 2 * p.tcy + 2 * halo, HZ = 2 * p.tcz + 2 * halo;
  const int HV = HX * HY * HZ;
  const int CC8 = p.cc4, CCs = p.ccs;  // octets per K chunk / fp16 elements per halo voxel in LDS (16 per octet + pad)
  const int taps = p.ksize == 3 ? 27 : 1;
  const int Qmax = taps * CC8;

  extern __shared__ __attribute__((aligned(16))) _Float16 smem_h2[];
  _Float16 *s_tile = smem_h2;                                                              // [HV][CCs]
  int *s_qoff = reinterpret_cast<int *>(smem_h2 + (((size_t)HV * CCs + 7) & ~(size_t)7));  // [Qmax + 4] byte offsets
  int *s_vox = s_qoff + ((Qmax + 4 + 3) & ~3);  // [HV] float offset of every halo voxel's channel row, -1 = padding
  // SKIP: [2] "the halo tile of the chunk staged last has a non-zero somewhere" (one flag per chunk parity).  The ligand's
  // channels are zero in every workgroup tile away from the ligand: such a chunk's whole K loop is skipped, by all waves
  // alike, instead of finding its steps dead one LDS round trip at a time.
  int *s_live = s_vox + HV;
  if (SKIP && tid < 2) s_live[tid] = 0;

  // octet q of a chunk (channel-major: all taps of octet 0, then octet 1, ...) -> byte offset inside the halo tile.  The
  // four entries behind the last octet repeat it: an odd octet count leaves the second half-wave of the last step on real
  // (finite) data, which its all-zero weight rows cancel, and the K loop reads the table one step ahead.
  for (int q = tid; q < Qmax + 4; q += NTHREADS) {
    const int qq = q < Qmax ? q : Qmax - 1;
    const int c8 = qq / taps, tap = qq - c8 * taps;
    const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    s_qoff[q] = ((p.ksize == 3 ? ((dx * HY + dy) * HZ + dz) * CCs : 0) + c8 * 16) * 2;
  }

  const int NC = p.tcx * p.tcy * p.tcz;
  const int oz = row & 1, oy = (row >> 1) & 1, ox = (row >> 3) & 1;
  const int cell_in_mt = ((row >> 2) & 1) + 2 * ((row >> 4) & 1);
  auto cell_of = [&](int mt, int cim, int &cx, int &cy, int &cz) -> bool {  // (ConvArgs::mt_x, as in conv3d.hip)
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
    baseA[m] = (((2 * cx + ox) * HY + (2 * cy + oy)) * HZ + (2 * cz + oz)) * CCs * 2;  // bytes
  }

  h2_f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; m++)
#pragma unroll
    for (int n = 0; n < TN; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
  unsigned n_exec = 0;  // (M-tile, step) pairs whose MFMAs this wave executed (SKIP; a scalar counter, read in profile mode only)
  float amax = 0.f;     // running maximum of |staged value| (range check, see split4)
  float out_max = 0.f;  // BWD: running maximum of |stored value| on the masked channel range (ConvArgs::out_amax)

  const int S = p.S;
  const int x0 = tx * 2 * p.tcx - halo, y0 = ty * 2 * p.tcy - halo, z0 = tz * 2 * p.tcz - halo;
  const unsigned inv_hz = ((1u << 20) + HZ - 1) / HZ, inv_hy = ((1u << 20) + HY - 1) / HY;  // (see conv3d_mfma_kernel)
  for (int hv = tid; hv < HV; hv += NTHREADS) {
    const int t1 = (int)(((unsigned)hv * inv_hz) >> 20), hz = hv - t1 * HZ;
    const int hx = (int)(((unsigned)t1 * inv_hy) >> 20), hy = t1 - hx * HY;
    const int x = x0 + hx, y = y0 + hy, z = z0 + hz;
    const bool in = (unsigned)x < (unsigned)S && (unsigned)y < (unsigned)S && (unsigned)z < (unsigned)S;
    s_vox[hv] = in ? ((x * S + y) * S + z) * p.in_cs : -1;
    if constexpr (BWD)  // max-unpool while staging: the voxel's cell of the half-resolution tensor, its position inside the cell on top
      if (in && p.in_mode == 2)
        s_vox[hv] = ((((x >> 1) * (S >> 1) + (y >> 1)) * (S >> 1) + (z >> 1)) * p.in_cs) | ((((x & 1) << 2) | ((y & 1) << 1) | (z & 1)) << 28);
    if (!in)  // the zero padding is laid down once per workgroup; staging then touches the voxels inside the grid only
      for (int c = 0; c < CCs; c += 8) *reinterpret_cast<uint4 *>(s_tile + hv * CCs + c) = make_uint4(0u, 0u, 0u, 0u);
  }
  const bool unpool = BWD && p.in_mode == 2;
  const int Sin = unpool ? S >> 1 : S;
  const size_t pose_floats = (size_t)Sin * Sin * Sin * p.in_cs;
  const float *in_b = p.in + (size_t)b * pose_floats;
  const size_t wstride = (size_t)p.coutp * 16;  // fp16 elements per octet row of the packed weights
  const int Pmax = (Qmax + 1) >> 1;
  float s_in = 1.f, inv_s_in = 1.f;  // (ConvArgs::in_amax, as in conv3d_h2_kernel)
  if constexpr (BWD) {
    if (p.in_amax) {
      const int e = (int)((p.in_amax[b] >> 23) & 0xffu);
      if (e > 0 && e < 255) {
        const int eb = max(4, min(250, 268 - e));
        s_in = __uint_as_float((unsigned)eb << 23);
        inv_s_in = __uint_as_float((unsigned)(254 - eb) << 23);
      }
    }
  }

  // ---- staging, software-pipelined over the K chunks: the fp32 quads of chunk k + 1 are loaded into registers before the
  // K loop of chunk k and split / written to LDS after it.  (A load -> split -> ds_write chain per halo voxel exposes one
  // L2 latency per voxel and chunk; at the f16 MFMA rate that was longer than the K loop itself: 116 KB of LDS = one
  // workgroup per CU, nothing to overlap with, ran the first conv at 4.3 ms against 2.25 ms with three per CU.)
  // A thread owns halo voxels tid, tid + NTHREADS, ...: VPT of them, NQ channel quads each (the plans keep
  // HV <= VPT * NTHREADS and 2 CC8 <= NQ): 3 x 4 under a 3x3x3 conv's halo, 1 x 12 for a 1x1x1 conv.
  constexpr int VPT = K1 ? 1 : 3, NQ = K1 ? 12 : 4;
  float4 pre[VPT][NQ];
  unsigned pre_am[VPT][NQ];  // BWD, in_mode 2: the four arg-max bytes of the quad
  auto issue = [&](int chunk) {
    const float *src_c = in_b + chunk * CC8 * 8;
    const int nq = min(2 * CC8, p.cin4 - chunk * 2 * CC8);  // channel quads of this chunk that exist in the input
#pragma unroll
    for (int v = 0; v < VPT; v++) {
      const int hv = tid + v * NTHREADS;
      const int off0 = hv < HV ? s_vox[hv] : -1;
      const int off = (BWD && off0 >= 0) ? off0 & 0x0fffffff : off0;
#pragma unroll
      for (int q = 0; q < NQ; q++)
        if (off >= 0 && q < nq) {
#if MI_H2_EXPERIMENT & 2
          pre[v][q] = make_float4(1.f, 0.f, 0.25f, 0.f);
#else
          pre[v][q] = *reinterpret_cast<const float4 *>(src_c + off + q * 4);
#endif
          if constexpr (BWD)
            if (unpool) pre_am[v][q] = *reinterpret_cast<const unsigned *>(p.in_argmax + (size_t)b * pose_floats + chunk * CC8 * 8 + off + q * 4);
        }
    }
  };
  auto commit = [&](int chunk) {
    const int c_base = chunk * CC8 * 8;
    const int nq = min(2 * CC8, p.cin4 - chunk * 2 * CC8);
    bool wave_nonzero = false;
#pragma unroll
    for (int v = 0; v < VPT; v++) {
      const int hv = tid + v * NTHREADS;
      const int off = hv < HV ? s_vox[hv] : -1;
      if (off < 0) continue;  // zero padding, laid down once
      _Float16 *dst = s_tile + hv * CCs;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        if (q >= 2 * CC8) continue;
        _Float16 *d = dst + (q >> 1) * 16 + (q & 1) * 4;
        uint2 h = make_uint2(0u, 0u), l = make_uint2(0u, 0u);
        // quads the input does not have (channel padding of the last octet, a partial last chunk's unused octets --
        // which the pad entries of s_qoff may point at) are zero, not the previous chunk's channels
        if (q < nq) {
          float4 x = pre[v][q];
          if (p.bn_scale) {  // eval BatchNorm on the conv input (scalar loads: the quad is wave-uniform); padding stays 0
            const h2_f32x4 sc = *(H2ConstQuadPtr)(const void *)(p.bn_scale + c_base + q * 4);
            const h2_f32x4 sh = *(H2ConstQuadPtr)(const void *)(p.bn_shift + c_base + q * 4);
            x.x = x.x * sc.x + sh.x;
            x.y = x.y * sc.y + sh.y;
            x.z = x.z * sc.z + sh.z;
            x.w = x.w * sc.w + sh.w;
          }
          if constexpr (BWD) {
            if (unpool) {  // max-unpool: the cell's gradient goes to the voxel that was its maximum
              const unsigned am = pre_am[v][q], rr = (unsigned)off >> 28;
              x.x = (am & 0xffu) == rr ? x.x : 0.f;
              x.y = ((am >> 8) & 0xffu) == rr ? x.y : 0.f;
              x.z = ((am >> 16) & 0xffu) == rr ? x.z : 0.f;
              x.w = (am >> 24) == rr ? x.w : 0.f;
            }
          }
          // the pooled voxel grid and ReLU'd activations are mostly zeros: a quad that is zero in all 64 voxels of the wave
          // (3 VALU + a scalar branch to find out) needs no arithmetic
          const unsigned any = __float_as_uint(x.x) | __float_as_uint(x.y) | __float_as_uint(x.z) | __float_as_uint(x.w);
          if (__builtin_amdgcn_ballot_w64(any != 0u) != 0ull) {
            if constexpr (BWD) x.x *= s_in, x.y *= s_in, x.z *= s_in, x.w *= s_in;
            split4(x, h, l, amax);
            wave_nonzero = true;
          }
        }
        *reinterpret_cast<uint2 *>(d) = h;
        *reinterpret_cast<uint2 *>(d + 8) = l;
      }
    }
    // (wave_nonzero is a per-lane variable: it was set in the lanes that were staging a voxel at that moment -- not
    // necessarily in lane 0, whose halo voxel may be padding)
    if (SKIP && __builtin_amdgcn_ballot_w64(wave_nonzero) != 0ull && lane == 0) s_live[chunk & 1] = 1;
  };
  __syncthreads();  // s_vox
  issue(0);
  for (int chunk = 0; chunk < p.nchunks; chunk++) {
    if (chunk > 0) __syncthreads();  // every wave is through the previous chunk's K loop: the tile may be overwritten
    commit(chunk);
    if (chunk + 1 < p.nchunks) issue(chunk + 1);
    __syncthreads();
    if (SKIP) {
      const int live = s_live[chunk & 1];
      if (tid == 0) s_live[(chunk + 1) & 1] = 0;  // (last read in the previous chunk's K loop, next written behind the next barrier)
      if (__builtin_amdgcn_readfirstlane(live) == 0) continue;
    }
    const int nq = min(2 * CC8, p.cin4 - chunk * 2 * CC8);

    // ---- K loop over octet pairs, ping-pong operand sets (see conv3d_bf16.hip) ----
    const int cc8_here = min(CC8, (nq + 1) >> 1);      // octets of this chunk that carry input channels
    const int P = (cc8_here * taps + 1) >> 1;          // its octet pairs (the packed weights hold Pmax per chunk)
    // weight rows: uniform chunk base in SGPRs + a 32-bit lane offset that is affine in the step
    const char *wbase = reinterpret_cast<const char *>(p.wp) + (size_t)chunk * Pmax * 2 * wstride * 2;
    const unsigned wlane = ((unsigned)(n_base + row) * 16u + (unsigned)kh * (unsigned)wstride) * 2u;
    const unsigned wstep = 2u * (unsigned)wstride * 2u;  // bytes per octet pair
    const int *lp = s_qoff + kh;
    int qo_next = lp[0];  // the tile offset of a lane's octet is read one step ahead of its use
    uint4 wh0[TN], wl0[TN], wh1[TN], wl1[TN], ah0[TM], al0[TM], ah1[TM], al1[TM];
    auto load_pair = [&](int pr, uint4 *ah, uint4 *al, uint4 *wh, uint4 *wl) {
      const int qo = qo_next;
#pragma unroll
      for (int m = 0; m < TM; m++) {
        const char *a = reinterpret_cast<const char *>(s_tile) + baseA[m] + qo;
        ah[m] = *reinterpret_cast<const uint4 *>(a);
        al[m] = *reinterpret_cast<const uint4 *>(a + 16);
      }
#pragma unroll
      for (int n = 0; n < TN; n++) {
        const char *w = wbase + (wlane + (unsigned)pr * wstep + (unsigned)n * 32u * 32u);
        wh[n] = *reinterpret_cast<const uint4 *>(w);
        wl[n] = *reinterpret_cast<const uint4 *>(w + 16);
      }
      qo_next = lp[2 * pr + 2];  // (behind the last pair: a pad entry, or the next octet's first tap -- unused either way)
    };
    auto mfma_pair = [&](const uint4 *ah, const uint4 *al, const uint4 *wh, const uint4 *wl) {
#pragma unroll
      for (int m = 0; m < TM; m++) {
        if constexpr (SKIP) {
          // all 32 voxels x 16 k of this step zero (h = 0 implies l = 0): nothing to add.  One v_or3 + v_or + v_cmp into
          // an SGPR pair and a scalar branch against three (x TN) MFMAs
          const unsigned any = ah[m].x | ah[m].y | ah[m].z | ah[m].w;
          unsigned long long live;
          asm volatile("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(live) : "v"(any));
          if (live == 0ull) continue;
          n_exec++;
        }
#pragma unroll
        for (int n = 0; n < TN; n++) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[m]), __builtin_bit_cast(f16x8, wh[n]), acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[m]), __builtin_bit_cast(f16x8, wl[n]), acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[m]), __builtin_bit_cast(f16x8, wh[n]), acc[m][n], 0, 0, 0);
        }
      }
    };
#if MI_H2_EXPERIMENT & 1
    if (p.nchunks > 1000)
#endif
    load_pair(0, ah0, al0, wh0, wl0);
    int pr = 0;
#if MI_H2_EXPERIMENT & 1
    if (p.nchunks > 1000)
#endif
    for (; pr + 1 < P; pr += 2) {
      load_pair(pr + 1, ah1, al1, wh1, wl1);
      mfma_pair(ah0, al0, wh0, wl0);
      if (pr + 2 < P) load_pair(pr + 2, ah0, al0, wh0, wl0);
      mfma_pair(ah1, al1, wh1, wl1);
    }
#if MI_H2_EXPERIMENT & 1
    if (p.nchunks > 1000)
#endif
    if (P & 1) mfma_pair(ah0, al0, wh0, wl0);
  }

  // ---- fused 1x1x1 conv behind this one (Default2018: conv3 -> ReLU -> conv1 -> ReLU -> pool; ConvArgs::post_w): the ReLU'd
  // tile is split and laid down in LDS as [voxel row][octet][h | l] -- what the stand-alone 1x1x1 kernel's staging would
  // build from the tensor in HBM, which therefore never exists -- and a second, short K loop runs over its channels.  Same
  // operands, same MFMA order as the two separate kernels: same bits (the gradient program runs them separately).
  float unscale = p.h2_unscale * inv_s_in;
  const float *bias_ptr = p.bias;
  int relu_flag = p.relu;
  if constexpr (!K1 && TN == 1 && TM <= 3) {
    if (p.post_w) {
      __syncthreads();  // every wave is through its last K loop: the halo tile may be overwritten
      const int CCm = 2 * p.coutp + 8;  // fp16 elements per voxel row of the mid tile (odd number of 16-byte slots)
      _Float16 *s_mid = smem_h2;
#pragma unroll
      for (int m = 0; m < TM; m++) {
        const int ch = n_base + row;
        const float b1 = p.bias[ch];
        _Float16 *dcol = s_mid + (ch >> 3) * 16 + (ch & 7);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int rowl = (wm * TM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          float t = acc[m][0][r] * unscale + b1;
          if (p.relu) t = fmaxf(t, 0.f);
          asm("v_maximum3_f32 %0, %0, |%1|, |%1|" : "+v"(amax) : "v"(t));
          const float c = __builtin_amdgcn_fmed3f(t, -65504.f, 65504.f);
          const _Float16 hi = (_Float16)c;
          dcol[rowl * CCm] = hi;
          dcol[rowl * CCm + 8] = (_Float16)(c - (float)hi);
          acc[m][0][r] = 0.f;
        }
      }
      __syncthreads();
      const int npairs = p.post_cc4 >> 1;  // octet pairs = steps of the second K loop
      const char *w2 = reinterpret_cast<const char *>(p.post_w) + ((size_t)kh * p.coutp + n_base + row) * 32;
      for (int pr = 0; pr < npairs; pr++) {
        uint4 ah[TM], al[TM], wh[1], wl[1];
#pragma unroll
        for (int m = 0; m < TM; m++) {
          const char *a = reinterpret_cast<const char *>(s_mid) + (((wm * TM + m) * 32 + row) * CCm + (2 * pr + kh) * 16) * 2;
          ah[m] = *reinterpret_cast<const uint4 *>(a);
          al[m] = *reinterpret_cast<const uint4 *>(a + 16);
        }
        const char *w = w2 + (size_t)pr * 2 * p.coutp * 32;
        wh[0] = *reinterpret_cast<const uint4 *>(w);
        wl[0] = *reinterpret_cast<const uint4 *>(w + 16);
#pragma unroll
        for (int m = 0; m < TM; m++) {
          acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[m]), __builtin_bit_cast(f16x8, wh[0]), acc[m][0], 0, 0, 0);
          acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[m]), __builtin_bit_cast(f16x8, wl[0]), acc[m][0], 0, 0, 0);
          acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[m]), __builtin_bit_cast(f16x8, wh[0]), acc[m][0], 0, 0, 0);
        }
      }
      unscale = p.h2_post_unscale;
      bias_ptr = p.post_bias;
      relu_flag = p.post_relu;
    }
  }

  // profile mode: executed MFMA work in units of 4,096 FLOPs (the fp32 kernels count 32x32x2 instructions): an executed
  // (M-tile, step) is 3 TN instructions of 32,768 FLOPs
  if (SKIP && p.mfma_count && lane == 0)
    atomicAdd(p.mfma_count + (wg & (kMfmaCountSlots - 1)), (unsigned long long)n_exec * (3u * TN * 8u));

  // ---- epilogue: un-scale, bias, ReLU, optional 2x2x2 pool, store channels-last fp32 ----
  const int So = p.pool ? S / 2 : S;
  const size_t out_pose = (size_t)b * So * So * So * p.out_cs + p.out_c0;
  float *out_f = p.out + out_pose;
  const int ncx = S / 2;
#pragma unroll
  for (int m = 0; m < TM; m++) {
#pragma unroll
    for (int half = 0; half < 2; half++) {
      int cx, cy, cz;
      if (!cell_of(wm * TM + m, kh + 2 * half, cx, cy, cz)) continue;
      const int gcx = tx * p.tcx + cx, gcy = ty * p.tcy + cy, gcz = tz * p.tcz + cz;
      if (gcx >= ncx || gcy >= ncx || gcz >= ncx) continue;
#pragma unroll
      for (int n = 0; n < TN; n++) {
        const int ch = n_base + n * 32 + row;
        if (ch >= p.cout) continue;
        const float bias = bias_ptr[ch];
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const float tt = acc[m][n][half * 8 + r] * unscale + bias;
          v[r] = relu_flag ? fmaxf(tt, 0.f) : tt;
        }
        if (p.pool == 1) {
          float mx = v[0];
          int am = 0;
#pragma unroll
          for (int r = 1; r < 8; r++)
            if (v[r] > mx) mx = v[r], am = r;
          const size_t o = (((size_t)gcx * So + gcy) * So + gcz) * p.out_cs + ch;
          out_f[o] = mx;
          if (p.argmax_out) p.argmax_out[out_pose + o] = (unsigned char)am;
        } else if (p.pool == 2) {
          float s = v[0];
#pragma unroll
          for (int r = 1; r < 8; r++) s = s + v[r];
          out_f[(((size_t)gcx * So + gcy) * So + gcz) * p.out_cs + ch] = s * 0.125f;
        } else {
          if constexpr (BWD) {
            // gradient pass: BatchNorm scale of the forward layer's input (ConvArgs::out_scale), then the ReLU mask and the
            // maximum on the channels [out_mask_c0, out_mask_c1) (ConvArgs::out_mask / out_amax); the activations are
            // fetched, and consumed, ahead of the first store
            const bool in_range = ch >= p.out_mask_c0 && ch < p.out_mask_c1;
            const bool any_msk = p.out_mask && __builtin_amdgcn_ballot_w64(in_range) != 0ull;
            const int ch_m = in_range ? ch : p.out_mask_c0;
            const float osc = p.out_scale ? p.out_scale[ch] : 1.0f;
            float a8[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
              const int vx = 2 * gcx + (r >> 2), vy = 2 * gcy + ((r >> 1) & 1), vz = 2 * gcz + (r & 1);
              a8[r] = 1.f;
              if (any_msk) a8[r] = p.out_mask[((size_t)b * So * So * So + ((size_t)vx * So + vy) * So + vz) * p.out_mask_cs + ch_m];
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
              if (p.out_scale) v[r] = v[r] * osc;
              if (in_range) {
                v[r] = a8[r] > 0.f ? v[r] : 0.f;
                out_max = fmaxf(out_max, fabsf(v[r]));
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int r = 0; r < 8; r++) {
            const int vx = 2 * gcx + (r >> 2), vy = 2 * gcy + ((r >> 1) & 1), vz = 2 * gcz + (r & 1);
            out_f[(((size_t)vx * So + vy) * So + vz) * p.out_cs + ch] = v[r];
          }
        }
      }
    }
  }
  if constexpr (BWD) {
    if (p.out_amax) {
      for (int o = 32; o; o >>= 1) out_max = fmaxf(out_max, __shfl_xor(out_max, o));
      if (lane == 0 && __float_as_uint(out_max) > __hip_atomic_load(p.out_amax + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(p.out_amax + b, __float_as_uint(out_max));
    }
  }
  h2_report_overflow(p.h2_overflow, !(amax <= 65504.f));
}

// ---------------------------------------------------------------------------------------------
// conv3d_h2_kernel: the 3x3x3 convolutions, 32 output channels per wave column.
//
// One K chunk = ONE octet of input channels (27 taps = 14 instruction steps: lanes 0-31 feed tap 2p, lanes 32-63 tap
// 2p + 1).  The halo tile of a chunk is PLANAR in LDS -- [h plane | l plane], a plane = one 16-byte slot (the eight h or
// the eight l of the octet) per halo voxel, slot = x * SX + y * SY + z with pad slots behind z-rows / x-planes
// (ConvArgs::h2_pad_*, chosen against bank conflicts: a ds_read_b128 lane group then hits sixteen different slots) -- and
// DOUBLE-BUFFERED: chunk k + 1 lands in the other buffer while the K loop of chunk k runs, one barrier per chunk.
//
// INSPLIT (ConvArgs::in_split): the input tensor is already split in HBM by its producer -- [octet][voxel][h8 | l8]: a
// slot of an LDS plane is 16 of those bytes, and a chunk's halo tile is a box of ONE dense array, z-rows of 32-byte voxels
// (channels-last tensors cost a cache line per voxel and chunk) -- so staging is a copy, and the copy is LDS-DMA (buffer_load_dwordx4 ... lds: no VGPRs, no
// VALU).  The DMA writes 64 consecutive slots per wave-instruction, lane i -> slot base + i, whatever the lanes' source
// addresses: a lane computes once which (voxel, half) its slots hold and keeps the byte offsets of their sources; voxels
// outside the grid and pad slots take an out-of-range offset, for which a buffer load returns zeros (the zero padding).
// !INSPLIT: fp32 input (the gradient program's forward pass, tensors of fp32-MFMA layers): the octet of chunk k + 1 is
// loaded into registers before the K loop of chunk k and split / written to the other buffer after it (split4).
// Either way the LDS image and the K loop are the same: a pose scores the same bits on both.
//
// MT: which four cells form an M-tile -- 0 raster neighbours, 1 stacked along x (tcx % 4 == 0), 2 a 2 x 2 square in (x, y)
// (tcx == tcy == 2); the epilogue follows, the results do not depend on it.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void *LdsPtr;
// staging extents of conv3d_h2_kernel: at most kH2NS wave-DMAs per thread and chunk (2 PL <= kH2NS * 256 slots), at most
// kH2VPT halo voxels per thread on the fp32 path (HV <= kH2VPT * 256).  (Plain constants: with extents that depend on a
// template parameter hipcc 7.2 silently drops the host stubs of the INSPLIT instantiations.)
constexpr int kH2NS = 8, kH2VPT = 4;

template <int WM, int WN, int TM, int MT, bool SKIP, bool INSPLIT, bool WLDS = false, int NP = 1, bool BWD = false>
__global__ __launch_bounds__(64 * WM * WN, (WLDS ? (NP >= 4 ? 2 : 3) : INSPLIT ? (TM <= 2 ? MI_H2_OCC2 : TM <= 3 ? MI_H2_OCC3 : TM <= 4 ? 2 : 1) : (TM <= 2 ? 3 : TM <= 4 ? 2 : 1))) void conv3d_h2_kernel(ConvArgs p) {
  constexpr int NW = WM * WN, NTHREADS = 64 * NW;
  static_assert(NW == 4, "staging is laid out for four waves");
  static_assert(!WLDS || WN == 1, "weights in LDS: the four waves of a workgroup share one set of 32 output channels");
  // NP poses per workgroup (WLDS + INSPLIT): the same tile of NP consecutive poses, one after the other, on ONE copy of the
  // chunk's weights in LDS -- the weights are most of what a workgroup pulls out of L2 (143 KB against a 97 KB halo tile)
  static_assert(NP == 1 || ((NP == 2 || NP == 4) && WLDS && INSPLIT && TM <= 2), "two / four poses per workgroup: weights-in-LDS variant, split-format input");
  // BWD: a transposed conv of the gradient pass (ConvArgs::in_amax / in_mode 2 / out_mask / out_amax) -- its own instantiation
  // so that the forward kernels carry none of its registers
  static_assert(!BWD || (!INSPLIT && WLDS && NP == 1), "gradient-pass variant: fp32 tensors, weights through LDS");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int kh = lane >> 5;
  const int row = lane & 31;

  const int tiles_per_pose = p.ntx * p.nty * p.ntz;
  const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int b = (wg / tiles_per_pose) * NP;                      // first pose of this workgroup
  const int npose = NP == 1 ? 1 : min(NP, p.nposes - b);         // (the last workgroups of an odd batch have one)
  int t = wg - (wg / tiles_per_pose) * tiles_per_pose;
  const int tz = t % p.ntz;
  t /= p.ntz;
  const int ty = t % p.nty, tx = t / p.nty;
  const int n_base = (blockIdx.y * WN + wn) * 32;

  const int HX = 2 * p.tcx + 2, HY = 2 * p.tcy + 2, HZ = 2 * p.tcz + 2;
  const int HV = HX * HY * HZ;
  const int SY = HZ + p.h2_pad_y, SX = HY * SY + p.h2_pad_x;  // strides in 16-byte slots
  const int PL = (HX * SX + 31) & ~31;                        // slots per plane; a buffer = 2 PL slots = whole wave-DMAs
  const int PLB = PL * 16, BUFB = 2 * PLB;

  constexpr int P = 14;          // steps per chunk: 27 taps in pairs (the 28th tap's weight rows are zero)
  constexpr int NBUF = WLDS ? 1 : 2;
  constexpr int WBYTES = WLDS ? P * 2048 : 0;  // WLDS: the chunk's B operands, [step][h | l][half-wave][32 couts][8 fp16]
  extern __shared__ __attribute__((aligned(16))) char smem_h2c[];
  char *const s_buf = smem_h2c;                                   // [NBUF buffers][h plane | l plane]
  char *const s_w = smem_h2c + NBUF * BUFB;                       // [WBYTES]
  int *const s_qoff = reinterpret_cast<int *>(s_w + WBYTES);      // [32] byte offset of tap q inside a plane (q >= 27: tap 26)
  int *const s_live = s_qoff + 32;                                // [nchunks][4]: wave w found a non-zero in chunk c's tile (SKIP)
  if (tid < 32) {
    const int tap = tid < 27 ? tid : 26;
    const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    s_qoff[tid] = (dx * SX + dy * SY + dz) * 16;
  }

  const int NC = p.tcx * p.tcy * p.tcz;
  const int oz = row & 1, oy = (row >> 1) & 1, ox = (row >> 3) & 1;
  const int cell_in_mt = ((row >> 2) & 1) + 2 * ((row >> 4) & 1);
  auto cell_of = [&](int mt, int cim, int &cx, int &cy, int &cz) __attribute__((always_inline)) -> bool {
    if (MT == 1) {
      cz = mt % p.tcz;
      cy = (mt / p.tcz) % p.tcy;
      cx = 4 * (mt / (p.tcz * p.tcy)) + cim;
      return cx < p.tcx;
    }
    if (MT == 2) {
      cz = mt, cy = cim & 1, cx = cim >> 1;
      return mt < p.tcz;
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
    baseA[m] = ((2 * cx + ox) * SX + (2 * cy + oy) * SY + (2 * cz + oz)) * 16;  // bytes inside a plane
  }

  h2_f32x16 acc[NP][TM];
#pragma unroll
  for (int tp = 0; tp < NP; tp++)
#pragma unroll
    for (int m = 0; m < TM; m++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[tp][m][r] = 0.f;
  unsigned n_exec = 0;  // (M-tile, step) pairs whose MFMAs this wave executed (SKIP; read in profile mode only)
  float amax = 0.f;     // !INSPLIT: running maximum of |staged value| (range check, see split4)
  bool ovf_out = false; // out_split: a value this lane wrote left the fp16 range
  float out_max = 0.f;  // fp32 output: running maximum of |stored value| (ConvArgs::out_amax)

  const int S = p.S;
  const int x0 = tx * 2 * p.tcx - 1, y0 = ty * 2 * p.tcy - 1, z0 = tz * 2 * p.tcz - 1;
  // (!INSPLIT, in_mode 2 -- a transposed conv behind a fused max pool: `in` and `in_argmax` are at S / 2)
  const bool unpool = BWD && p.in_mode == 2;
  const int Sin = unpool ? S >> 1 : S;
  const size_t pose_floats = (size_t)Sin * Sin * Sin * p.in_cs;
  const float *in_b = p.in + (size_t)b * pose_floats;
  // dynamic range of a gradient tensor (ConvArgs::in_amax): staged values times s_in = 2^(14 - exponent of the pose's
  // largest), accumulators times 1 / s_in
  float s_in = 1.f, inv_s_in = 1.f;
  if constexpr (BWD) {
    if (p.in_amax) {
      const int e = (int)((p.in_amax[b] >> 23) & 0xffu);  // biased exponent; 0 = all zero (or subnormal), 255 = inf / NaN
      if (e > 0 && e < 255) {
        const int eb = max(4, min(250, 268 - e));  // biased exponent of 2^(14 - (e - 127)), both factors kept normal
        s_in = __uint_as_float((unsigned)eb << 23);
        inv_s_in = __uint_as_float((unsigned)(254 - eb) << 23);
      }
    }
  }

  // ---- which chunks (octets) have a non-zero in this workgroup's halo tile?  With the voxelizer's occupancy bytes
  // (ConvArgs::in_occ: one byte per pose, 4 x 4 x 4-cell block of the pooled grid and octet) that is known before anything
  // is staged, and an all-zero chunk -- the ligand's channels in every tile away from the ligand -- costs no DMA, no
  // barrier, nothing.  Without them the staging finds out (s_live, one flag per chunk and wave). ----
  unsigned todo = p.nchunks >= 32 ? 0xffffffffu : ((1u << p.nchunks) - 1u);
  unsigned todo_p[NP];  // per pose of this workgroup
#pragma unroll
  for (int tp = 0; tp < NP; tp++) todo_p[tp] = todo;
  bool occ_known = false;
  if constexpr (SKIP && INSPLIT) {
    if (p.in_occ && p.nchunks <= 8) {
      // blocks the halo tile touches, per axis: voxels [x0, x0 + HX) -> blocks (x0 >> 2) .. ((x0 + HX - 1) >> 2)
      const int bx0 = x0 >> 2, by0 = y0 >> 2, bz0 = z0 >> 2;
      const int nbx = ((x0 + HX - 1) >> 2) - bx0 + 1, nby = ((y0 + HY - 1) >> 2) - by0 + 1, nbz = ((z0 + HZ - 1) >> 2) - bz0 + 1;
      if (nbx * nby * nbz <= 64) {
        // wave tp reads pose tp's blocks (NP <= 4 waves) and reduces them to a mask of live octets
        uint2 o8 = make_uint2(0u, 0u);
        if (wave < npose && lane < nbx * nby * nbz) {
          const int bz = bz0 + lane % nbz, by = by0 + (lane / nbz) % nby, bx = bx0 + lane / (nbz * nby);
          const int nt = p.occ_nt;
          if ((unsigned)bx < (unsigned)nt && (unsigned)by < (unsigned)nt && (unsigned)bz < (unsigned)nt)
            o8 = *reinterpret_cast<const uint2 *>(p.in_occ + (((size_t)(b + wave) * nt + bx) * nt + by) * nt * 8 + (size_t)bz * 8);
        }
        if (wave < NP) {
          unsigned m = 0u;
#pragma unroll
          for (int o = 0; o < 8; o++) {
            const unsigned byte = ((o < 4 ? o8.x : o8.y) >> (8 * (o & 3))) & 0xffu;
            if (__builtin_amdgcn_ballot_w64(byte != 0u) != 0ull) m |= 1u << o;
          }
          if (lane == 0) s_live[wave] = (int)m;
        }
        __syncthreads();
        todo = 0u;
#pragma unroll
        for (int tp = 0; tp < NP; tp++) {
          todo_p[tp] &= (unsigned)__builtin_amdgcn_readfirstlane(s_live[tp]);
          if (tp < npose) todo |= todo_p[tp];
        }
        occ_known = true;
        __syncthreads();  // (s_live is reused below when a later launch path writes it: keep the read ahead of any write)
      }
    }
  }

  // ---- staging set-up ----
  // INSPLIT: NS wave-DMAs per thread and chunk; slot j = tid + i * NTHREADS of a buffer is half j / PL of plane slot j % PL
  constexpr int NS = kH2NS;  // (plans keep 2 PL <= NS * NTHREADS)
  unsigned voff[NS];
  // !INSPLIT: a thread owns halo voxels tid, tid + NTHREADS, ... (plans keep HV <= VPT * NTHREADS)
  constexpr int VPT = kH2VPT;
  int st_slot[VPT], st_off[VPT];
  float4 pre[VPT][2];
  unsigned pre_am[VPT][2];  // in_mode 2: the four arg-max bytes of the quad
  __amdgpu_buffer_rsrc_t rsrc;
  if constexpr (INSPLIT) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in_b), 0, (int)(pose_floats * 4), 0x00020000);
    const unsigned inv_sx = ((1u << 20) + SX - 1) / SX, inv_sy = ((1u << 20) + SY - 1) / SY;  // exact for n < 2^20 / d
#pragma unroll
    for (int i = 0; i < NS; i++) {
      const int j = tid + i * NTHREADS;
      const int half = j >= PL ? 1 : 0;
      const int ps = j - half * PL;
      const int hx = (int)(((unsigned)ps * inv_sx) >> 20);
      const int r1 = ps - hx * SX;
      const int hy = (int)(((unsigned)r1 * inv_sy) >> 20), hz = r1 - hy * SY;
      const int x = x0 + hx, y = y0 + hy, z = z0 + hz;
      const bool ok = hx < HX && hy < HY && hz < HZ && (unsigned)x < (unsigned)S && (unsigned)y < (unsigned)S && (unsigned)z < (unsigned)S;
      voff[i] = ok ? (unsigned)((x * S + y) * S + z) * 32u + (unsigned)half * 16u : 0x80000000u;  // inside an octet's [voxel][h | l] array
      if (p.h2_dbg & 32) voff[i] = (unsigned)(((x0 + 1) * S + (y0 + 1)) * S + z0 + 1) * 32u + (unsigned)j * 16u;  // (timing: contiguous sources)
    }
  } else {
    const unsigned inv_hz = ((1u << 20) + HZ - 1) / HZ, inv_hy = ((1u << 20) + HY - 1) / HY;
#pragma unroll
    for (int v = 0; v < VPT; v++) {
      const int hv = tid + v * NTHREADS;
      st_slot[v] = 0, st_off[v] = -1;
      if (hv >= HV) continue;
      const int t1 = (int)(((unsigned)hv * inv_hz) >> 20), hz = hv - t1 * HZ;
      const int hx = (int)(((unsigned)t1 * inv_hy) >> 20), hy = t1 - hx * HY;
      const int x = x0 + hx, y = y0 + hy, z = z0 + hz;
      const bool in = (unsigned)x < (unsigned)S && (unsigned)y < (unsigned)S && (unsigned)z < (unsigned)S;
      st_slot[v] = (hx * SX + hy * SY + hz) * 16;
      if (in) {
        st_off[v] = ((x * S + y) * S + z) * p.in_cs;
        if (unpool) {  // the voxel's cell of the pooled tensor; its position inside the cell rides in the slot's top bits
          st_off[v] = (((x >> 1) * Sin + (y >> 1)) * Sin + (z >> 1)) * p.in_cs;
          st_slot[v] |= (((x & 1) << 2) | ((y & 1) << 1) | (z & 1)) << 28;
        }
      } else {  // the zero padding is laid down once, in both buffers; staging then touches the voxels inside the grid only
#pragma unroll
        for (int k = 0; k < 2 * NBUF; k++) *reinterpret_cast<uint4 *>(s_buf + (k >> 1) * BUFB + (k & 1) * PLB + st_slot[v]) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  }
  const int octet_bytes = S * S * S * 32;  // split format: one octet's [voxel][h8 | l8] array
  auto issue_dma = [&](int chunk, int bsel, int tp = 0) {
    if constexpr (INSPLIT) {
      __amdgpu_buffer_rsrc_t rs = rsrc;
      if (NP > 1 && tp > 0) rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in_b + (size_t)tp * pose_floats), 0, (int)(pose_floats * 4), 0x00020000);
      char *dst = s_buf + bsel * BUFB + wave * 1024;
#pragma unroll
      for (int i = 0; i < NS; i++)
        if ((i * NW + wave) * 64 < 2 * PL && !(p.h2_dbg & 4))  // (wave-uniform; 2 PL is a multiple of 64)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LdsPtr)(dst + i * NW * 1024), 16, voff[i], chunk * octet_bytes, 0, 0);
    }
  };
  // L2 prefetch of a tile that will be staged one phase later: a single-buffered tile cannot be DMA'd ahead of time, and
  // its first-touch lines come from HBM, ~2 us away (measured: 2.3 us per (chunk, pose) phase with nothing else running).
  // One dword per voxel of the NEXT item's h plane is DMA'd into a 256-byte junk region while this item's K loop runs --
  // the lines (h and l share a 64-byte sector) are then in L2 when the real DMA asks for them.  Inline asm: the compiler
  // must not know this writes LDS (it would hold every ds_read of the K loop back until the prefetch has landed).
  typedef int h2_i32x4 __attribute__((ext_vector_type(4)));
  auto prefetch_tile = [&](int chunk, int tp) {
    if constexpr (INSPLIT) {
      const unsigned long long a = (unsigned long long)(in_b + (size_t)tp * pose_floats);
      h2_i32x4 rs;
      // (wave-uniform by construction; readfirstlane makes that a fact for the "s" constraint of the asm below -- without it a
      // shift in register pressure once put the descriptor in VGPRs, which the instruction does not take)
      rs.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull)), rs.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffull));
      rs.z = __builtin_amdgcn_readfirstlane((int)(pose_floats * 4)), rs.w = 0x00020000;
      const unsigned junk = (unsigned)(size_t)(s_live + p.nchunks * 4);
      const int soff = __builtin_amdgcn_readfirstlane(chunk * octet_bytes);
#pragma unroll
      for (int i = 0; i < (NS + 1) / 2; i++)
        if ((i * NW + wave) * 64 < PL) {
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep)
                       : "v"(voff[i]), "s"(junk), "s"(rs), "s"(soff)
                       : "memory");
        }
    }
  };
  auto issue_ld = [&](int chunk) {
    if constexpr (!INSPLIT) {
      const float *src_c = in_b + chunk * 8;
      const int nq = (p.h2_dbg & 4) ? 0 : min(2, p.cin4 - chunk * 2);  // channel quads of this octet that exist in the input
#pragma unroll
      for (int v = 0; v < VPT; v++)
#pragma unroll
        for (int q = 0; q < 2; q++)
          if (st_off[v] >= 0 && q < nq) {
            pre[v][q] = *reinterpret_cast<const float4 *>(src_c + st_off[v] + q * 4);
            if constexpr (BWD)
              if (unpool) pre_am[v][q] = *reinterpret_cast<const unsigned *>(p.in_argmax + (size_t)b * pose_floats + chunk * 8 + st_off[v] + q * 4);
          }
    }
  };
  auto commit = [&](int chunk, int bsel) {
    if constexpr (!INSPLIT) {
      const int c_base = chunk * 8;
      const int nq = min(2, p.cin4 - chunk * 2);
      char *dstb = s_buf + bsel * BUFB;
      bool lane_nonzero = false;
#pragma unroll
      for (int v = 0; v < VPT; v++) {
        if (st_off[v] < 0) continue;
        uint2 h[2], l[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
          h[q] = make_uint2(0u, 0u), l[q] = make_uint2(0u, 0u);
          if (q < nq) {  // (a quad the input does not have -- channel padding of the last octet -- is zero)
            float4 x = pre[v][q];
            if (p.bn_scale) {  // eval BatchNorm on the conv input (scalar loads: the quad is wave-uniform); padding stays 0
              const h2_f32x4 sc = *(H2ConstQuadPtr)(const void *)(p.bn_scale + c_base + q * 4);
              const h2_f32x4 sh = *(H2ConstQuadPtr)(const void *)(p.bn_shift + c_base + q * 4);
              x.x = x.x * sc.x + sh.x;
              x.y = x.y * sc.y + sh.y;
              x.z = x.z * sc.z + sh.z;
              x.w = x.w * sc.w + sh.w;
            }
            if constexpr (BWD) {
              if (unpool) {  // max-unpool: the cell's gradient goes to the voxel that was its maximum
                const unsigned am = pre_am[v][q], rr = (unsigned)st_slot[v] >> 28;
                x.x = (am & 0xffu) == rr ? x.x : 0.f;
                x.y = ((am >> 8) & 0xffu) == rr ? x.y : 0.f;
                x.z = ((am >> 16) & 0xffu) == rr ? x.z : 0.f;
                x.w = (am >> 24) == rr ? x.w : 0.f;
              }
            }
            const unsigned any = __float_as_uint(x.x) | __float_as_uint(x.y) | __float_as_uint(x.z) | __float_as_uint(x.w);
            if (__builtin_amdgcn_ballot_w64(any != 0u) != 0ull) {  // (a quad that is zero in all 64 voxels needs no arithmetic)
              if constexpr (BWD) x.x *= s_in, x.y *= s_in, x.z *= s_in, x.w *= s_in;
              split4(x, h[q], l[q], amax);
              lane_nonzero = true;
            }
          }
        }
        const int slot = BWD ? st_slot[v] & 0x0fffffff : st_slot[v];
        *reinterpret_cast<uint4 *>(dstb + slot) = make_uint4(h[0].x, h[0].y, h[1].x, h[1].y);
        *reinterpret_cast<uint4 *>(dstb + PLB + slot) = make_uint4(l[0].x, l[0].y, l[1].x, l[1].y);
      }
      if (SKIP && lane == 0) s_live[chunk * 4 + wave] = 0;
      if (SKIP && __builtin_amdgcn_ballot_w64(lane_nonzero) != 0ull && lane == 0) s_live[chunk * 4 + wave] = 1;
    }
  };
  // INSPLIT + SKIP: is there a non-zero in the slots of chunk `chunk` this wave's DMAs wrote?  (h = 0 implies l = 0: the
  // h plane decides.  A wave may read what its OWN DMAs wrote once its vmcnt covers them, no barrier needed.)
  auto probe_dma = [&](int chunk, int bsel) {
    if constexpr (INSPLIT && SKIP) {
      if (occ_known) return;
      const char *src = s_buf + bsel * BUFB;
      unsigned any = 0u;
#pragma unroll
      for (int i = 0; i < (NS + 1) / 2; i++) {
        const int j = tid + i * NTHREADS;
        if (j < PL) {
          const uint4 q = *reinterpret_cast<const uint4 *>(src + j * 16);
          any |= q.x | q.y | q.z | q.w;
        }
      }
      const bool live = __builtin_amdgcn_ballot_w64(any != 0u) != 0ull;
      if (lane == 0) s_live[chunk * 4 + wave] = live ? 1 : 0;
    }
  };

  // packed weights [chunk][step][h | l][half-wave][cout][8 fp16]: a lane's h row and, 2 coutp rows on, its l row
  const unsigned wlane = ((unsigned)kh * (unsigned)p.coutp + (unsigned)(n_base + row)) * 16u;
  const unsigned wl_off = 2u * (unsigned)p.coutp * 16u;
  const unsigned wstep = 2u * wl_off;  // bytes per step
  const int *lp = s_qoff + kh;
  uint4 wh0, wl0, wh1, wl1, ah0[TM], al0[TM], ah1[TM], al1[TM];
  if (p.h2_dbg & 24) {  // (timing experiments: operands that are not loaded)
    wh0 = wl0 = wh1 = wl1 = make_uint4(0x3c003c00u, 0u, 0x3c003c00u, 0u);
#pragma unroll
    for (int m = 0; m < TM; m++) ah0[m] = al0[m] = ah1[m] = al1[m] = make_uint4(0x3c003c00u, 1u, 0u, 0u);
  }
  // (the pose index is a compile-time constant everywhere: a pointer into acc[][] sends the accumulators to scratch)
  auto mfma_pair = [&](auto tpc, const uint4 *ah, const uint4 *al, const uint4 &wh, const uint4 &wl) __attribute__((always_inline)) {
    constexpr int tp = decltype(tpc)::value;
#pragma unroll
    for (int m = 0; m < TM; m++) {
      if constexpr (SKIP) {
        // all 32 voxels x 16 k of this step zero (h = 0 implies l = 0): nothing to add.  One v_or3 + v_or + v_cmp into
        // an SGPR pair and a scalar branch against three MFMAs
        const unsigned any = ah[m].x | ah[m].y | ah[m].z | ah[m].w;
        unsigned long long lv;
        asm volatile("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(lv) : "v"(any));
        if (lv == 0ull) continue;
        n_exec++;
      }
      acc[tp][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[m]), __builtin_bit_cast(f16x8, wh), acc[tp][m], 0, 0, 0);
      acc[tp][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[m]), __builtin_bit_cast(f16x8, wl), acc[tp][m], 0, 0, 0);
      acc[tp][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[m]), __builtin_bit_cast(f16x8, wh), acc[tp][m], 0, 0, 0);
    }
  };
  auto chunk_live = [&](int chunk) -> bool {
    bool live = true;
    if constexpr (SKIP) {
      const int4 lv = *reinterpret_cast<const int4 *>(s_live + chunk * 4);
      live = __builtin_amdgcn_readfirstlane(lv.x | lv.y | lv.z | lv.w) != 0;
      if (p.h2_dbg & 1) live = true;
    }
    if (p.h2_dbg & 2) live = false;
    return live;
  };

  if constexpr (WLDS) {
    // ---- B operands through LDS.  Read straight from L1 / L2, the weights are the K loop's bottleneck (measured: 0.5 of
    // the first conv's 1.8 ms; more of them in flight does not help -- it is the vector memory path's throughput, every
    // wave of a workgroup fetching the same 2 KB per step).  Here the four waves share them: a chunk's 28 KB go to LDS
    // by DMA, once per workgroup, next to a SINGLE halo-tile buffer (48 KB in all: three workgroups per CU, whose DMA and
    // K-loop phases overlap), and the K loop reads nothing but LDS. ----
    __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wp), 0, (int)(p.nchunks * P * wstep), 0x00020000);
    auto issue_w = [&](int chunk) {
      // piece q = (step, h | l) = 1 KB = one wave-DMA: lane (kh, n) fetches its 16 bytes of row kh * coutp + n_base + n
#pragma unroll
      for (int i = 0; i < 2 * P / NW; i++) {
        const int q = i * NW + wave;
        if (!(p.h2_dbg & 8))
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (LdsPtr)(s_w + q * 1024), 16, wlane, chunk * (int)(P * wstep) + q * (int)wl_off, 0, 0);
      }
    };
    static_assert(2 * P % NW == 0, "weight pieces per wave");
    int chunk = todo ? __builtin_ctz(todo) : -1;
    if constexpr (!INSPLIT)
      if (chunk >= 0) issue_ld(chunk);
    bool first = true;
    while (chunk >= 0) {
      const unsigned rest = chunk >= 31 ? 0u : (todo & ~((2u << chunk) - 1u));
      const int next = rest ? __builtin_ctz(rest) : -1;
      bool w_here = false;  // this chunk's weights are in LDS
      auto pose_pass = [&](auto tpc) __attribute__((always_inline)) {
        constexpr int tp = decltype(tpc)::value;
        if (occ_known && !((todo_p[tp] >> chunk) & 1u)) return;  // (the occupancy bytes say: nothing of this pose in this chunk)
        if (!first) __syncthreads();  // every wave is through the previous K loop: tile (and weights) may be overwritten
        first = false;
        if constexpr (INSPLIT) issue_dma(chunk, 0, tp);
        else commit(chunk, 0);
        if (!w_here) issue_w(chunk);
        w_here = true;
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        if constexpr (INSPLIT) probe_dma(chunk, 0);
        __syncthreads();
        if constexpr (!INSPLIT)
          if (next >= 0) issue_ld(next);  // (in flight during the K loop, which waits for nothing but LDS)
        if constexpr (INSPLIT) {
          if (p.h2_prefetch) {  // the next item's tile: towards L2 while this K loop runs
            if (tp + 1 < NP && tp + 1 < npose) prefetch_tile(chunk, tp + 1);
            else if (next >= 0) prefetch_tile(next, 0);
          }
        }
        const bool live = occ_known ? !(p.h2_dbg & 2) : chunk_live(chunk);
        if (live) {
          const char *tile = s_buf;
          const char *wl_ = s_w + lane * 16;
          int qo_next = lp[0];
          auto load_pair = [&](int pr, uint4 *ah, uint4 *al, uint4 &wh, uint4 &wl) {
            const int qo = qo_next;
#pragma unroll
            for (int m = 0; m < TM; m++) {
              const char *a = tile + baseA[m] + qo;
              if (p.h2_dbg & 16) continue;
              ah[m] = *reinterpret_cast<const uint4 *>(a);
              al[m] = *reinterpret_cast<const uint4 *>(a + PLB);
            }
            wh = *reinterpret_cast<const uint4 *>(wl_ + pr * 2048);
            wl = *reinterpret_cast<const uint4 *>(wl_ + pr * 2048 + 1024);
            qo_next = lp[2 * pr + 2];
          };
          load_pair(0, ah0, al0, wh0, wl0);
#pragma unroll 1
          for (int pr = 0; pr < P; pr += 2) {
            load_pair(pr + 1, ah1, al1, wh1, wl1);
            mfma_pair(tpc, ah0, al0, wh0, wl0);
            if (pr + 2 < P) load_pair(pr + 2, ah0, al0, wh0, wl0);
            mfma_pair(tpc, ah1, al1, wh1, wl1);
          }
        }
      };
      pose_pass(std::integral_constant<int, 0>{});
      if constexpr (NP > 1)
        if (npose > 1) pose_pass(std::integral_constant<int, 1>{});
      if constexpr (NP > 2) {
        if (npose > 2) pose_pass(std::integral_constant<int, 2>{});
        if (npose > 3) pose_pass(std::integral_constant<int, 3>{});
      }
      chunk = next;
    }
  } else {
    // the chunks to do, in order; chunk number it of them lands in buffer it & 1
    int chunk = todo ? __builtin_ctz(todo) : -1;
    if (chunk >= 0) {
      if constexpr (INSPLIT) issue_dma(chunk, 0);
      else issue_ld(chunk);
    }
    int it = 0;
    while (chunk >= 0) {
      const unsigned rest = chunk >= 31 ? 0u : (todo & ~((2u << chunk) - 1u));
      const int next = rest ? __builtin_ctz(rest) : -1;
      const int bsel = it & 1;
      // ---- `chunk` becomes visible in buffer bsel; behind the barrier every wave is also through the previous K loop,
      // i.e. through with the other buffer, which the staging of the next chunk may now overwrite ----
      if constexpr (INSPLIT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        probe_dma(chunk, bsel);
      } else {
        commit(chunk, bsel);
        if (next >= 0) issue_ld(next);
      }
      __syncthreads();
      const char *tile = s_buf + bsel * BUFB;
      const char *wbase = reinterpret_cast<const char *>(p.wp) + (size_t)chunk * P * wstep;
      const bool live = occ_known ? !(p.h2_dbg & 2) : chunk_live(chunk);

      // ---- K loop over tap pairs, ping-pong operand sets.  The operands of the first two steps are requested BEFORE the
      // DMAs of the next chunk (memory operations return in order: a weight load queued behind the DMAs would wait for
      // them; a ds_read behind an LDS-DMA waits for it too -- the compiler cannot tell the two buffers apart) ----
      int qo_next = lp[0];
      auto load_pair = [&](int pr, uint4 *ah, uint4 *al, uint4 &wh, uint4 &wl) {
        const int qo = qo_next;
#pragma unroll
        for (int m = 0; m < TM; m++) {
          const char *a = tile + baseA[m] + qo;
          if (p.h2_dbg & 16) continue;
          ah[m] = *reinterpret_cast<const uint4 *>(a);
          al[m] = *reinterpret_cast<const uint4 *>(a + PLB);
        }
        const char *w = wbase + (wlane + (unsigned)pr * wstep);
        if (!(p.h2_dbg & 8)) {
          wh = *reinterpret_cast<const uint4 *>(w);
          wl = *reinterpret_cast<const uint4 *>(w + wl_off);
        }
        qo_next = lp[2 * pr + 2];  // (behind the last pair: a pad entry -- unused)
      };
      if (live) {
        load_pair(0, ah0, al0, wh0, wl0);
        load_pair(1, ah1, al1, wh1, wl1);
      }
      if (next >= 0) issue_dma(next, bsel ^ 1);
      if (live) {
#pragma unroll 1
        for (int pr = 0; pr < P; pr += 2) {
          mfma_pair(std::integral_constant<int, 0>{}, ah0, al0, wh0, wl0);
          if (pr + 2 < P) load_pair(pr + 2, ah0, al0, wh0, wl0);
          mfma_pair(std::integral_constant<int, 0>{}, ah1, al1, wh1, wl1);
          if (pr + 3 < P) load_pair(pr + 3, ah1, al1, wh1, wl1);
        }
      }
      chunk = next;
      it++;
    }
  }

  // ---- fused 1x1x1 conv behind this one (Default2018: conv3 -> ReLU -> conv1 -> ReLU -> pool; ConvArgs::post_w): the ReLU'd
  // tile is split and laid down in LDS as [voxel row][octet][h | l] -- what the stand-alone 1x1x1 kernel's staging would
  // build from the tensor in HBM, which therefore never exists -- and a second, short K loop runs over its channels.  Same
  // operands, same MFMA order as the two separate kernels: same bits (the gradient program runs them separately).  (Inside
  // the per-pose loop below.)
  // profile mode: executed MFMA work in units of 4,096 FLOPs (the fp32 kernels count 32x32x2 instructions): an executed
  // (M-tile, step) is 3 instructions of 32,768 FLOPs
  if (SKIP && p.mfma_count && lane == 0)
    atomicAdd(p.mfma_count + (wg & (kMfmaCountSlots - 1)), (unsigned long long)n_exec * (3u * 8u));

  const int So = p.pool ? S / 2 : S;
  const int ncx = S / 2;
  const int ch = n_base + row;
  // split format (out_c0 = 0, whole octets): [octet][voxel][h8 | l8]; dword of this lane's store inside a voxel's 32 bytes
  const int ch_sp = ((ch & 7) >> 1) + (row & 1) * 4;
  const size_t oct_sp = (size_t)(ch >> 3) * So * So * So;
  auto finish_pose = [&](auto tpc) __attribute__((always_inline)) {  // the poses of this workgroup, one after the other
    constexpr int tp = decltype(tpc)::value;
    float unscale = p.h2_unscale * inv_s_in;
    bool post_done = false;
    int relu_flag = p.relu;
    if constexpr (TM <= 3) {
      if (p.post_w) {
        __syncthreads();  // every wave is through its last K loop (the previous pose's second pass): LDS may be overwritten
        const int CCm = 2 * p.coutp + 8;  // fp16 elements per voxel row of the mid tile (odd number of 16-byte slots)
        _Float16 *s_mid = reinterpret_cast<_Float16 *>(smem_h2c);
#pragma unroll
        for (int m = 0; m < TM; m++) {
          const float b1 = p.bias[ch];
          _Float16 *dcol = s_mid + (ch >> 3) * 16 + (ch & 7);
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int rowl = (wm * TM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            float tt = acc[tp][m][r] * unscale + b1;
            if (p.relu) tt = fmaxf(tt, 0.f);
            asm("v_maximum3_f32 %0, %0, |%1|, |%1|" : "+v"(amax) : "v"(tt));
            const float c = __builtin_amdgcn_fmed3f(tt, -65504.f, 65504.f);
            const _Float16 hi = (_Float16)c;
            dcol[rowl * CCm] = hi;
            dcol[rowl * CCm + 8] = (_Float16)(c - (float)hi);
            acc[tp][m][r] = 0.f;
          }
        }
        __syncthreads();
        const int npairs = p.post_cc4 >> 1;  // octet pairs = steps of the second K loop
        const char *w2 = reinterpret_cast<const char *>(p.post_w) + ((size_t)kh * p.coutp + n_base + row) * 32;
        for (int pr = 0; pr < npairs; pr++) {
          uint4 ah[TM], al[TM], wh, wl;
#pragma unroll
          for (int m = 0; m < TM; m++) {
            const char *a = reinterpret_cast<const char *>(s_mid) + (((wm * TM + m) * 32 + row) * CCm + (2 * pr + kh) * 16) * 2;
            ah[m] = *reinterpret_cast<const uint4 *>(a);
            al[m] = *reinterpret_cast<const uint4 *>(a + 16);
          }
          const char *w = w2 + (size_t)pr * 2 * p.coutp * 32;
          wh = *reinterpret_cast<const uint4 *>(w);
          wl = *reinterpret_cast<const uint4 *>(w + 16);
#pragma unroll
          for (int m = 0; m < TM; m++) {
            acc[tp][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[m]), __builtin_bit_cast(f16x8, wh), acc[tp][m], 0, 0, 0);
            acc[tp][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[m]), __builtin_bit_cast(f16x8, wl), acc[tp][m], 0, 0, 0);
            acc[tp][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[m]), __builtin_bit_cast(f16x8, wh), acc[tp][m], 0, 0, 0);
          }
        }
        unscale = p.h2_post_unscale;
        post_done = true;
        relu_flag = p.post_relu;
      }
    }

    // ---- epilogue: un-scale, bias, ReLU, optional 2x2x2 pool, store channels-last -- fp32, or split (out_split): the lanes
    // of channels 2j and 2j + 1 trade halves (one DPP move), the even lane stores the two h, the odd lane the two l ----
    const size_t out_pose = (size_t)(b + tp) * So * So * So * p.out_cs + p.out_c0;
    float *out_f = p.out + out_pose;
    auto store = [&](size_t vox, float v) __attribute__((always_inline)) {  // vox = voxel index inside the pose
      if (p.out_split) {
        ovf_out |= !(fabsf(v) <= 65504.f);
        const unsigned mine = split1(v);
        const unsigned other = (unsigned)__builtin_amdgcn_mov_dpp((int)mine, 0xb1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
        // even lane: h(mine) | h(other) << 16; odd lane: l(other) | l(mine) << 16
        const unsigned w = (row & 1) ? ((other >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (other << 16));
        if (ch < p.coutp) reinterpret_cast<unsigned *>(out_f)[(oct_sp + vox) * 8 + ch_sp] = w;
      } else if (ch < p.cout) {
        out_f[vox * p.out_cs + ch] = v;
      }
    };
    // (two loads and a select: a pointer chosen at run time would live in scratch)
    const float bias = ch < p.coutp ? (post_done ? p.post_bias[ch] : p.bias[ch]) : 0.f;
    if constexpr (BWD) {
      // Gradient pass (no ReLU, no pool, fp32 output).  Dense-block layers scale and accumulate (ConvArgs::out_scale /
      // accumulate, as in conv3d_mfma_kernel's epilogue); then the ReLU of the layer this gradient belongs to, on the channels
      // [out_mask_c0, out_mask_c1) (ConvArgs::out_mask), and the maximum of what is stored there.  Every activation and
      // old value of the wave's 2 TM cells is fetched ahead of the first store: a store may alias the next load as far as
      // the compiler knows, and one exposed round trip to HBM per cell is a fifth of a short K loop.
      const bool msk = p.out_mask && ch >= p.out_mask_c0 && ch < p.out_mask_c1 && ch < p.cout;
      const bool accum = p.accumulate && ch < p.cout;
      const float osc = (p.out_scale && ch < p.cout) ? p.out_scale[ch] : 1.0f;
      float a8[TM][2][8], o8[TM][2][8];
      size_t vox0[TM][2];
      bool valid[TM][2];
#pragma unroll
      for (int m = 0; m < TM; m++)
#pragma unroll
        for (int half = 0; half < 2; half++) {
          int cx, cy, cz;
          valid[m][half] = cell_of(wm * TM + m, kh + 2 * half, cx, cy, cz);
          const int gcx = tx * p.tcx + cx, gcy = ty * p.tcy + cy, gcz = tz * p.tcz + cz;
          valid[m][half] = valid[m][half] && gcx < ncx && gcy < ncx && gcz < ncx;
          vox0[m][half] = ((size_t)(2 * gcx) * So + 2 * gcy) * So + 2 * gcz;
          if (!valid[m][half]) vox0[m][half] = 0;
        }
      // (the loads sit behind wave-uniform branches only and every lane has a valid address -- a per-lane `cond ? load : c`
      // is compiled into a branch, a wait and a select per element)
      const int ch_m = msk ? ch : p.out_mask_c0, ch_a = accum ? ch : 0;
      const bool any_msk = __builtin_amdgcn_ballot_w64(msk) != 0ull;  // (waves of the other 32-channel groups skip the fetch)
#pragma unroll
      for (int m = 0; m < TM; m++)
#pragma unroll
        for (int half = 0; half < 2; half++)
#pragma unroll
          for (int r = 0; r < 8; r++) {
            const size_t vox = vox0[m][half] + ((size_t)(r >> 2) * So + ((r >> 1) & 1)) * So + (r & 1);
            a8[m][half][r] = 1.f, o8[m][half][r] = 0.f;
            if (any_msk) a8[m][half][r] = p.out_mask[((size_t)(b + tp) * So * So * So + vox) * p.out_mask_cs + ch_m];
            if (p.accumulate) o8[m][half][r] = out_f[vox * p.out_cs + ch_a];
          }
      // (every fetched value is consumed before the first store is issued: the compiler otherwise sinks each load down to
      // its use, behind the previous store it may alias)
#pragma unroll
      for (int m = 0; m < TM; m++)
#pragma unroll
        for (int half = 0; half < 2; half++)
#pragma unroll
          for (int r = 0; r < 8; r++) {
            float v = acc[tp][m][half * 8 + r] * unscale + bias;
            if (p.out_scale) v = v * osc;
            if (p.accumulate) v = o8[m][half][r] + v;
            if (msk) v = a8[m][half][r] > 0.f ? v : 0.f;
            if (valid[m][half] && ch >= p.out_mask_c0 && ch < p.out_mask_c1) out_max = fmaxf(out_max, fabsf(v));
            acc[tp][m][half * 8 + r] = v;
          }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < TM; m++)
#pragma unroll
        for (int half = 0; half < 2; half++) {
          if (!valid[m][half] || ch >= p.cout) continue;
#pragma unroll
          for (int r = 0; r < 8; r++) {
            const size_t vox = vox0[m][half] + ((size_t)(r >> 2) * So + ((r >> 1) & 1)) * So + (r & 1);
            out_f[vox * p.out_cs + ch] = acc[tp][m][half * 8 + r];
          }
        }
      return;
    }
#pragma unroll
    for (int m = 0; m < TM; m++) {
#pragma unroll
      for (int half = 0; half < 2; half++) {
        int cx, cy, cz;
        if (!cell_of(wm * TM + m, kh + 2 * half, cx, cy, cz)) continue;
        const int gcx = tx * p.tcx + cx, gcy = ty * p.tcy + cy, gcz = tz * p.tcz + cz;
        if (gcx >= ncx || gcy >= ncx || gcz >= ncx) continue;
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const float tt = acc[tp][m][half * 8 + r] * unscale + bias;
          v[r] = relu_flag ? fmaxf(tt, 0.f) : tt;
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
  };
  finish_pose(std::integral_constant<int, 0>{});
  if constexpr (NP > 1)
    if (npose > 1) finish_pose(std::integral_constant<int, 1>{});
  if constexpr (NP > 2) {
    if (npose > 2) finish_pose(std::integral_constant<int, 2>{});
    if (npose > 3) finish_pose(std::integral_constant<int, 3>{});
  }
  if constexpr (BWD) {
    if (p.out_amax) {  // (ConvArgs::out_amax)
      for (int o = 32; o; o >>= 1) out_max = fmaxf(out_max, __shfl_xor(out_max, o));
      // (a plain read first: the maximum settles after a few workgroups, and same-line atomics serialize in L2 -- ~9 ns each,
      // 0.25 ms per launch when every wave issues one; a stale read only costs an atomic that changes nothing)
      if (lane == 0 && __float_as_uint(out_max) > __hip_atomic_load(p.out_amax + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(p.out_amax + b, __float_as_uint(out_max));
    }
  }
  h2_report_overflow(p.h2_overflow, ovf_out || !(amax <= 65504.f));
}

// conv3d_h2_16_kernel's LDS in front of its optional weight buffers: tile, tap offsets, voxel offsets (16-byte aligned)
__host__ __device__ inline size_t h2_16_main_lds_bytes(size_t HX, size_t SX, int Qmax, size_t HV) {
  const size_t b = ((HX * SX + 7) & ~(size_t)7) * sizeof(_Float16) + (size_t)((Qmax + 8 + 3) & ~3) * sizeof(int) + (HV + 4) * sizeof(int);
  return (b + 1023) & ~(size_t)1023;
}

// ---------------------------------------------------------------------------------------------
// 16-output-channel variant for the Dense blocks (Cout = 16): v_mfma_f32_16x16x32_f16.  An M-tile is 16 voxels = two cells
// (row = cell * 8 + x * 4 + y * 2 + z, as in conv3d_mfma16_kernel); the 32 k of an instruction are four octets, lane
// group l >> 4 feeding the fourth it owns.  Eval BatchNorm (scale and shift, fp32) is applied while staging; no zero test
// (three 16-cycle MFMAs per test are not worth one, and BatchNorm'ed activations are not zeros).
// ---------------------------------------------------------------------------------------------
//
// WL (round 6; the 6^3 layers): the chunk's packed weights (Smax steps x 2 KB) go through LDS, DMA'd one chunk ahead into
// one of two buffers as [step][h | l][lane] -- a per-pose call has three workgroups on the chip for such a layer, each
// pulling 300 KB of weights that nothing else has touched since the previous call, and with the register ring (four steps
// ahead, ~100 MFMA cycles each) every step waited out most of an L2 miss: 3.5 us per chunk for 0.6 us of MFMAs.  Same
// operands into the same MFMAs in the same order.
template <int TM, bool WL = false>
__global__ __launch_bounds__(256, (TM <= 2 ? 3 : 2)) void conv3d_h2_16_kernel(ConvArgs p) {
  constexpr int NTHREADS = 256;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane >> 4;   // which of the four octets of a step this lane feeds
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
  const int CC8 = p.cc4, CCs = p.ccs;
  // LDS strides in fp16 elements: voxel, z-row, x-plane.  Rows and planes may carry pad slots of 16 bytes
  // (ConvArgs::h2_pad_y / h2_pad_x, chosen with tools/microbench/lds_bank_sim_h2.py against bank conflicts)
  const int SZ = CCs, SY = HZ * SZ + 8 * p.h2_pad_y, SX = HY * SY + 8 * p.h2_pad_x;
  const int taps = p.ksize == 3 ? 27 : 1;
  const int Qmax = taps * CC8;
  const int Smax = (Qmax + 3) >> 2;  // steps (octet quartets) per chunk of the packed weights

  extern __shared__ __attribute__((aligned(16))) _Float16 smem_h2[];
  _Float16 *s_tile = smem_h2;
  int *s_qoff = reinterpret_cast<int *>(smem_h2 + (((size_t)HX * SX + 7) & ~(size_t)7));  // [Qmax + 8] byte offsets
  int *s_vox = s_qoff + ((Qmax + 8 + 3) & ~3);
  // (WL) two weight buffers behind the kernel's other LDS: h2_16_main_lds_bytes
  char *const s_wbuf = reinterpret_cast<char *>(smem_h2) + h2_16_main_lds_bytes(HX, SX, Qmax, HV);
  for (int q = tid; q < Qmax + 8; q += NTHREADS) {
    const int qq = q < Qmax ? q : Qmax - 1;
    const int c8 = qq / taps, tap = qq - c8 * taps;
    const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    s_qoff[q] = ((p.ksize == 3 ? dx * SX + dy * SY + dz * SZ : 0) + c8 * 16) * 2;
  }

  const int NC = p.tcx * p.tcy * p.tcz;
  const int oz = row & 1, oy = (row >> 1) & 1, ox = (row >> 2) & 1, cell_in_mt = row >> 3;
  int baseA[TM];
#pragma unroll
  for (int m = 0; m < TM; m++) {
    int cell = (wm * TM + m) * 2 + cell_in_mt;
    if (cell >= NC) cell = 0;
    const int cz = cell % p.tcz, cy = (cell / p.tcz) % p.tcy, cx = cell / (p.tcz * p.tcy);
    baseA[m] = ((2 * cx + ox) * SX + (2 * cy + oy) * SY + (2 * cz + oz) * SZ) * 2;  // bytes
  }
  h2_f32x4 acc[TM];
#pragma unroll
  for (int m = 0; m < TM; m++) acc[m] = {0.f, 0.f, 0.f, 0.f};
  float amax = 0.f;  // running maximum of |staged value| (range check, see split4)
  // a wave none of whose M-tiles has a cell of the tile (a 1 x 1 x 3-cell latency tile at 6^3 keeps two of the four busy)
  // stages and synchronises with the others but skips the K loops: its LDS reads would compete with theirs
  const bool wave_has_cells = wm * TM * 2 < NC;

  const int S = p.S;
  const int x0 = tx * 2 * p.tcx - halo, y0 = ty * 2 * p.tcy - halo, z0 = tz * 2 * p.tcz - halo;
  const unsigned inv_hz = ((1u << 20) + HZ - 1) / HZ, inv_hy = ((1u << 20) + HY - 1) / HY;
  constexpr int VPT = 3, NQ = 4;  // staging: halo voxels per thread x channel quads per chunk (see conv3d_h2_kernel)
  int st_dst[VPT];                // LDS element of this thread's v-th halo voxel
#pragma unroll
  for (int v = 0; v < VPT; v++) {
    const int hv = tid + v * NTHREADS;
    st_dst[v] = 0;
    if (hv >= HV) continue;
    const int t1 = (int)(((unsigned)hv * inv_hz) >> 20), hz = hv - t1 * HZ;
    const int hx = (int)(((unsigned)t1 * inv_hy) >> 20), hy = t1 - hx * HY;
    const int x = x0 + hx, y = y0 + hy, z = z0 + hz;
    const bool in = (unsigned)x < (unsigned)S && (unsigned)y < (unsigned)S && (unsigned)z < (unsigned)S;
    s_vox[hv] = in ? ((x * S + y) * S + z) * p.in_cs : -1;
    st_dst[v] = hx * SX + hy * SY + hz * SZ;
    if (!in)
      for (int c = 0; c < CCs; c += 8) *reinterpret_cast<uint4 *>(s_tile + st_dst[v] + c) = make_uint4(0u, 0u, 0u, 0u);
  }
  const float *in_b = p.in + (size_t)b * S * S * S * p.in_cs;

  // staging, software-pipelined over the K chunks (see conv3d_h2_kernel)
  float4 pre[VPT][NQ];
  auto issue = [&](int chunk) {
    const float *src_c = in_b + chunk * CC8 * 8;
    const int nq = min(2 * CC8, p.cin4 - chunk * 2 * CC8);  // channel quads of this chunk that exist in the input
#pragma unroll
    for (int v = 0; v < VPT; v++) {
      const int hv = tid + v * NTHREADS;
      const int off = hv < HV ? s_vox[hv] : -1;
#pragma unroll
      for (int q = 0; q < NQ; q++)
        if (off >= 0 && q < nq) pre[v][q] = *reinterpret_cast<const float4 *>(src_c + off + q * 4);
    }
  };
  auto commit = [&](int chunk) {
    const int c_base = chunk * CC8 * 8;
    const int nq = min(2 * CC8, p.cin4 - chunk * 2 * CC8);
#pragma unroll
    for (int v = 0; v < VPT; v++) {
      const int hv = tid + v * NTHREADS;
      const int off = hv < HV ? s_vox[hv] : -1;
      if (off < 0) continue;  // zero padding, laid down once
      _Float16 *dst = s_tile + st_dst[v];
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        if (q >= 2 * CC8) continue;
        _Float16 *d = dst + (q >> 1) * 16 + (q & 1) * 4;
        uint2 h = make_uint2(0u, 0u), l = make_uint2(0u, 0u);
        // quads the input does not have (channel padding of the last octet, a partial last chunk's unused octets --
        // which the pad entries of s_qoff may point at) are zero, not the previous chunk's channels
        if (q < nq) {
          float4 x = pre[v][q];
          if (p.bn_scale) {  // eval BatchNorm on the conv input (scalar loads: the quad is wave-uniform); padding stays 0
            const h2_f32x4 sc = *(H2ConstQuadPtr)(const void *)(p.bn_scale + c_base + q * 4);
            const h2_f32x4 sh = *(H2ConstQuadPtr)(const void *)(p.bn_shift + c_base + q * 4);
            x.x = x.x * sc.x + sh.x;
            x.y = x.y * sc.y + sh.y;
            x.z = x.z * sc.z + sh.z;
            x.w = x.w * sc.w + sh.w;
          }
          split4(x, h, l, amax);
        }
        *reinterpret_cast<uint2 *>(d) = h;
        *reinterpret_cast<uint2 *>(d + 8) = l;
      }
    }
  };
  // (WL) a chunk's weights = 2 Smax pieces of 1 KB (step, h | l): wave w DMAs the pieces w, w + 4, ... (h or l of the steps
  // (w >> 1), (w >> 1) + 2, ...); lane i's 16 bytes come from byte i * 32 of the step's 2 KB and land at byte i * 16 of the piece
  const int wbuf_bytes = Smax * 2048;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(static_cast<const void *>(p.wp)), 0, p.nchunks * wbuf_bytes, 0x00020000);
  auto issue_w = [&](int chunk) {
    if constexpr (WL) {
      // (inline asm: behind the builtin the compiler waits for vmcnt(0) in front of the next LDS read it cannot tell apart)
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(s_wbuf - reinterpret_cast<char *>(smem_h2)) + (chunk & 1) * wbuf_bytes + wm * 1024);
      unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(chunk * wbuf_bytes + (wm >> 1) * 2048 + (wm & 1) * 16);
      const unsigned wl32 = (unsigned)lane * 32u;
      const int npieces = __builtin_amdgcn_readfirstlane((2 * Smax - wm + 3) >> 2);
      unsigned keep, m0v;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 %1, %2" : "=&s"(keep), "=&s"(m0v) : "s"(dst));
      for (int i = 0; i < npieces; i++) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(m0v), "v"(wl32), "s"(rsrc_w), "s"(soff) : "memory");
        m0v += 0x1000u;
        soff += 4096u;
      }
      asm volatile("s_mov_b32 m0, %0" : : "s"(keep) : "memory");
    }
  };
  __syncthreads();  // s_vox
  if constexpr (WL) issue_w(0);
  issue(0);
  for (int chunk = 0; chunk < p.nchunks; chunk++) {
    if (chunk > 0) __syncthreads();
    commit(chunk);
    if constexpr (WL) {
      // this chunk's weights (requested one K loop ago, in front of this chunk's activation loads) are in LDS; the other
      // buffer was last read by the K loop in front of the barrier above
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (chunk + 1 < p.nchunks) issue_w(chunk + 1);
    }
    if (chunk + 1 < p.nchunks) issue(chunk + 1);
    __syncthreads();
    const int nq = min(2 * CC8, p.cin4 - chunk * 2 * CC8);
    const int cc8_here = min(CC8, (nq + 1) >> 1);
    const int NS = (cc8_here * taps + 3) >> 2;  // steps of this chunk (the packed weights hold Smax per chunk)

    // K loop over octet quartets, ping-pong operand sets; weights packed [chunk][step][4][16][h8 | l8], zero rows behind
    // the last octet of a chunk
    const char *wbase = reinterpret_cast<const char *>(p.wp) + (size_t)chunk * Smax * 4 * 16 * 32;
    const unsigned wlane = ((unsigned)kg * 16u + (unsigned)row) * 32u;
    const int *lp = s_qoff + kg;
    int qo_next = lp[0];
    // The A operands come from LDS one step ahead (ping-pong); the weights come from L2 and are requested FOUR steps ahead
    // (a ring of four register sets, the loop unrolled by four): one step ahead, a chunk's 14 steps were a chain of 14 L2
    // round trips -- 3.6 us per chunk with one or two workgroups on the chip, which is what a B = 1 call of a Dense model
    // spends in its four 6^3 layers (47-59 us each).  Same MFMAs in the same order.
    uint4 wh[4], wl[4], ah0[TM], al0[TM], ah1[TM], al1[TM];
    auto load_a = [&](uint4 *ah, uint4 *al, int st_next) {
      const int qo = qo_next;
#pragma unroll
      for (int m = 0; m < TM; m++) {
        const char *a = reinterpret_cast<const char *>(s_tile) + baseA[m] + qo;
        ah[m] = *reinterpret_cast<const uint4 *>(a);
        al[m] = *reinterpret_cast<const uint4 *>(a + 16);
      }
      qo_next = lp[4 * st_next];
    };
    auto load_w = [&](int st, uint4 &h, uint4 &l) {
      if constexpr (WL) {
        const char *w = s_wbuf + (chunk & 1) * wbuf_bytes + st * 2048 + lane * 16;
        h = *reinterpret_cast<const uint4 *>(w);
        l = *reinterpret_cast<const uint4 *>(w + 1024);
      } else {
        const char *w = wbase + (wlane + (unsigned)st * (4u * 16u * 32u));
        h = *reinterpret_cast<const uint4 *>(w);
        l = *reinterpret_cast<const uint4 *>(w + 16);
      }
    };
    auto mfma_step = [&](const uint4 *ah, const uint4 *al, const uint4 &h, const uint4 &l) {
      // (three independent passes over the M-tiles: consecutive MFMAs never wait for each other's accumulator)
#pragma unroll
      for (int m = 0; m < TM; m++)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al[m]), __builtin_bit_cast(f16x8, h), acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < TM; m++)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[m]), __builtin_bit_cast(f16x8, l), acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < TM; m++)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[m]), __builtin_bit_cast(f16x8, h), acc[m], 0, 0, 0);
    };
    if (!wave_has_cells) continue;
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (u < NS) load_w(u, wh[u], wl[u]);
    load_a(ah0, al0, 1);
    for (int st = 0; st < NS; st += 4) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (st + u < NS) {
          uint4 *ah_c = (u & 1) ? ah1 : ah0, *al_c = (u & 1) ? al1 : al0;
          uint4 *ah_n = (u & 1) ? ah0 : ah1, *al_n = (u & 1) ? al0 : al1;
          if (st + u + 1 < NS) load_a(ah_n, al_n, st + u + 2);
          mfma_step(ah_c, al_c, wh[u], wl[u]);
          if (st + u + 4 < NS) load_w(st + u + 4, wh[u], wl[u]);
        }
      }
    }
  }

  // epilogue: accumulator row = 4 * (lane >> 4) + reg, column = lane & 15
  const float unscale = p.h2_unscale;
  float *out_b = p.out + (size_t)b * S * S * S * p.out_cs + p.out_c0;
  const int ncx = S / 2;
  const int ch = row;
  if (ch < p.cout) {
    const float bias = p.bias[ch];
#pragma unroll
    for (int m = 0; m < TM; m++) {
      const int cell = (wm * TM + m) * 2 + (kg >> 1);
      if (cell >= NC) continue;
      const int cz = cell % p.tcz, cy = (cell / p.tcz) % p.tcy, cx = cell / (p.tcz * p.tcy);
      const int gcx = tx * p.tcx + cx, gcy = ty * p.tcy + cy, gcz = tz * p.tcz + cz;
      if (gcx >= ncx || gcy >= ncx || gcz >= ncx) continue;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int vx = 2 * gcx + (kg & 1), vy = 2 * gcy + (r >> 1), vz = 2 * gcz + (r & 1);
        float v = acc[m][r] * unscale + bias;
        if (p.relu) v = fmaxf(v, 0.f);
        out_b[(((size_t)vx * S + vy) * S + vz) * p.out_cs + ch] = v;
      }
    }
  }
  h2_report_overflow(p.h2_overflow, !(amax <= 65504.f));
}

// ---------------------------------------------------------------------------------------------
// conv3d_h2_16_ring_kernel (round 6): conv3d_h2_16_kernel for launches of a few small workgroups -- the 6^3 layers of a
// per-pose call: nine workgroups of 1 x 1 x 3 cells on a 256-CU chip.  There the layer is a chain of memory round trips:
// every K chunk needs 16 fresh input channels of the tile (fp32, written by the previous launch) and 28 KB of packed weights
// (nobody else has touched them since the previous call), the K loop behind them is 0.4 us, and requested one chunk ahead
// each chunk waited ~1.5 us for its operands.  Here BOTH come by LDS-DMA into a ring of ConvArgs::h16_ring slots
// ([raw fp32 channels as [quad][halo voxel] | weights as [step][h | l][lane]]), requested h16_ring - 1 chunks ahead
// and counted in by hand (every wave issues the same number of wave-DMAs per chunk; the only VMEM operations of the loop);
// "staging" reads the raw slot from LDS, applies the BatchNorm, splits and writes the tile.  Same values into the same
// MFMAs in the same order as conv3d_h2_16_kernel: same bits.
// ---------------------------------------------------------------------------------------------
constexpr int kH16RingSteps = 14;  // steps (octet quartets) of a chunk of at most two octets x 27 taps
constexpr int kH16RingMaxRaw = 4;  // raw wave-DMAs per wave and chunk: quads x halo voxels (rounded to 64) <= 4 * 256

__device__ __forceinline__ void h2_wait_vm(int n) {  // s_waitcnt vmcnt(n), n wave-uniform: a scalar branch to the immediate
#define MIG_VM1(N) case N: __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14)); break;
#define MIG_VM4(N) MIG_VM1(N) MIG_VM1(N + 1) MIG_VM1(N + 2) MIG_VM1(N + 3)
#define MIG_VM16(N) MIG_VM4(N) MIG_VM4(N + 4) MIG_VM4(N + 8) MIG_VM4(N + 12)
  switch (n) {
    MIG_VM16(0) MIG_VM16(16) MIG_VM16(32)
    MIG_VM4(48) MIG_VM4(52) MIG_VM4(56) MIG_VM1(60) MIG_VM1(61) MIG_VM1(62)
    default: __builtin_amdgcn_s_waitcnt(0x0F70 | 15 | (3 << 14)); break;  // vmcnt(63)
  }
#undef MIG_VM16
#undef MIG_VM4
#undef MIG_VM1
}

template <int TM>
__global__ __launch_bounds__(256, 1) void conv3d_h2_16_ring_kernel(ConvArgs p) {
  constexpr int NTHREADS = 256;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane >> 4, row = lane & 15;

  const int tiles_per_pose = p.ntx * p.nty * p.ntz;
  const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int b = wg / tiles_per_pose;
  int t = wg - b * tiles_per_pose;
  const int tz = t % p.ntz;
  t /= p.ntz;
  const int ty = t % p.nty, tx = t / p.nty;

  const int HX = 2 * p.tcx + 2, HY = 2 * p.tcy + 2, HZ = 2 * p.tcz + 2;
  const int HV = HX * HY * HZ;
  const int CC8 = p.cc4, CCs = p.ccs;
  const int SZ = CCs, SY = HZ * SZ + 8 * p.h2_pad_y, SX = HY * SY + 8 * p.h2_pad_x;  // fp16 elements
  constexpr int taps = 27;
  const int Qmax = taps * CC8;
  const int Smax = (Qmax + 3) >> 2;
  const int NQ = 2 * CC8;                       // channel quads per voxel and chunk (<= 4)
  const int HVp = (HV + 63) & ~63;
  const int n_raw = (NQ * HVp + 255) >> 8;      // wave-DMAs per wave and chunk: the raw channels ...
  const int n_w = (2 * Smax + 3) >> 2;          // ... and the weights (pieces of 1 KB: (step, h | l))
  const int raw_bytes = n_raw * 4096, slot_bytes = raw_bytes + n_w * 4096;
  const int R = p.h16_ring;

  extern __shared__ __attribute__((aligned(16))) _Float16 smem_h2[];
  _Float16 *s_tile = smem_h2;
  int *s_qoff = reinterpret_cast<int *>(smem_h2 + (((size_t)HX * SX + 7) & ~(size_t)7));
  int *s_vox = s_qoff + ((Qmax + 8 + 3) & ~3);
  // (s_vox holds HVp entries here; behind it the layer's BatchNorm scale and shift, 2 x nchunks x 16 floats: as scalar loads
  // from global memory inside `commit` they were a round trip per chunk that nothing hid -- 2.7 us per chunk)
  const int cpad = p.nchunks * CC8 * 8;
  float *const s_bn = reinterpret_cast<float *>(reinterpret_cast<char *>(smem_h2) + h2_16_main_lds_bytes(HX, SX, Qmax, HVp));
  const int ring_off = (int)(h2_16_main_lds_bytes(HX, SX, Qmax, HVp) + (((size_t)2 * cpad * sizeof(float) + 1023) & ~(size_t)1023));
  char *const s_ring = reinterpret_cast<char *>(smem_h2) + ring_off;
  if (p.bn_scale)
    for (int c = tid; c < cpad; c += NTHREADS) {
      const bool has = c < p.cin4 * 4;  // (the arrays hold the layer's input channels, padded to whole quads)
      s_bn[c] = has ? p.bn_scale[c] : 1.f;
      s_bn[cpad + c] = has ? p.bn_shift[c] : 0.f;
    }
  const float bias_pre = p.bias[row < p.cout ? row : 0];  // (the epilogue must not start with a round trip either)
  for (int q = tid; q < Qmax + 8; q += NTHREADS) {
    const int qq = q < Qmax ? q : Qmax - 1;
    const int c8 = qq / taps, tap = qq - c8 * taps;
    const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    s_qoff[q] = ((dx * SX + dy * SY + dz * SZ) + c8 * 16) * 2;
  }

  const int NC = p.tcx * p.tcy * p.tcz;
  const int oz = row & 1, oy = (row >> 1) & 1, ox = (row >> 2) & 1, cell_in_mt = row >> 3;
  int baseA[TM];
#pragma unroll
  for (int m = 0; m < TM; m++) {
    int cell = (wm * TM + m) * 2 + cell_in_mt;
    if (cell >= NC) cell = 0;
    const int cz = cell % p.tcz, cy = (cell / p.tcz) % p.tcy, cx = cell / (p.tcz * p.tcy);
    baseA[m] = ((2 * cx + ox) * SX + (2 * cy + oy) * SY + (2 * cz + oz) * SZ) * 2;  // bytes
  }
  h2_f32x4 acc[TM];
#pragma unroll
  for (int m = 0; m < TM; m++) acc[m] = {0.f, 0.f, 0.f, 0.f};
  float amax = 0.f;
  const bool wave_has_cells = wm * TM * 2 < NC;

  const int S = p.S;
  const int x0 = tx * 2 * p.tcx - 1, y0 = ty * 2 * p.tcy - 1, z0 = tz * 2 * p.tcz - 1;
  const unsigned inv_hz = ((1u << 20) + HZ - 1) / HZ, inv_hy = ((1u << 20) + HY - 1) / HY;
  // this thread's halo voxel (HV <= 256: the launcher), zero padding laid down once
  int st_dst = 0, my_off = -1;
  if (tid < HV) {
    const int t1 = (int)(((unsigned)tid * inv_hz) >> 20), hz = tid - t1 * HZ;
    const int hx = (int)(((unsigned)t1 * inv_hy) >> 20), hy = t1 - hx * HY;
    const int x = x0 + hx, y = y0 + hy, z = z0 + hz;
    const bool in = (unsigned)x < (unsigned)S && (unsigned)y < (unsigned)S && (unsigned)z < (unsigned)S;
    my_off = in ? ((x * S + y) * S + z) * p.in_cs : -1;
    st_dst = hx * SX + hy * SY + hz * SZ;
    if (!in)
      for (int c = 0; c < CCs; c += 8) *reinterpret_cast<uint4 *>(s_tile + st_dst + c) = make_uint4(0u, 0u, 0u, 0u);
  }
  for (int hv = tid; hv < HVp; hv += NTHREADS) s_vox[hv] = hv == tid ? my_off : -1;
  __syncthreads();  // s_vox, s_qoff

  int qo[kH16RingSteps];  // this lane group's octet of every step of a chunk: byte offset inside the tile
#pragma unroll
  for (int st = 0; st < kH16RingSteps; st++) qo[st] = s_qoff[min(4 * st + kg, Qmax + 7)];
  // DMA sources of the raw channels: slot j = tid + i * 256 = quad j / HVp of halo voxel j % HVp (consecutive lanes =
  // consecutive voxels: staging reads its voxel's quads conflict-free); outside the grid / the slots: out of range = zeros
  unsigned voff[kH16RingMaxRaw];
#pragma unroll
  for (int i = 0; i < kH16RingMaxRaw; i++) {
    const int j = tid + i * NTHREADS;
    const int q = j / HVp, hv = j - q * HVp;
    const int off = (i < n_raw && q < NQ) ? s_vox[hv] : -1;
    voff[i] = off >= 0 ? (unsigned)(off + q * 4) * 4u : 0x80000000u;
  }
  typedef int h16_i32x4 __attribute__((ext_vector_type(4)));
  auto make_rsrc = [&](const void *base, unsigned bytes) __attribute__((always_inline)) {
    const unsigned long long a = (unsigned long long)base;
    h16_i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull)), rs.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffull));
    rs.z = __builtin_amdgcn_readfirstlane((int)bytes), rs.w = 0x00020000;
    return rs;
  };
  const size_t pose_floats = (size_t)S * S * S * p.in_cs;
  const h16_i32x4 rsrc_in = make_rsrc(p.in + (size_t)b * pose_floats, (unsigned)(pose_floats * 4));
  const int wchunk = Smax * 2048;
  const h16_i32x4 rsrc_w = make_rsrc(p.wp, (unsigned)(p.nchunks * wchunk));
  const unsigned wl32 = (unsigned)lane * 32u;
  // (inline asm: the compiler must not know these write LDS -- it would hold the K loop's reads back until every chunk
  // requested ahead has landed; conv3d_h2_ws.hip)
  auto issue_chunk = [&](int chunk) __attribute__((always_inline)) {
    const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane(ring_off + (chunk % R) * slot_bytes + wm * 1024);
    const unsigned soff_in = (unsigned)__builtin_amdgcn_readfirstlane(chunk * CC8 * 32);
    unsigned keep, m0v;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 %1, %2" : "=&s"(keep), "=&s"(m0v) : "s"(slot));
#pragma unroll
    for (int i = 0; i < kH16RingMaxRaw; i++)
      if (i < n_raw) {
        if (!(p.h2_dbg & 512))  // (timing only)
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(m0v), "v"(voff[i]), "s"(rsrc_in), "s"(soff_in) : "memory");
        m0v += 0x1000u;
      }
    unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(chunk * wchunk + (wm >> 1) * 2048 + (wm & 1) * 16);
    for (int i = 0; i < n_w; i++) {
      if (!(p.h2_dbg & 1024))  // (timing only)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(m0v), "v"(wl32), "s"(rsrc_w), "s"(soff) : "memory");
      m0v += 0x1000u;
      soff += 4096u;
    }
    asm volatile("s_mov_b32 m0, %0" : : "s"(keep) : "memory");
  };
  // VMEM operations per wave and chunk (the timing switches leave some out: they wait for everything instead)
  const int opc = (p.h2_dbg & (512 | 1024)) ? 0 : n_raw + n_w;

  auto commit = [&](int chunk) __attribute__((always_inline)) {
    if (my_off < 0) return;  // zero padding, laid down once (and the threads behind the halo)
    const int c_base = chunk * CC8 * 8;
    const int nq = min(NQ, p.cin4 - chunk * NQ);
    const char *raw = s_ring + (chunk % R) * slot_bytes;
    _Float16 *dst = s_tile + st_dst;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (q >= NQ) continue;
      _Float16 *d = dst + (q >> 1) * 16 + (q & 1) * 4;
      uint2 h = make_uint2(0u, 0u), l = make_uint2(0u, 0u);
      if (q < nq) {  // (quads the input does not have are zero: conv3d_h2_16_kernel)
        float4 x = *reinterpret_cast<const float4 *>(raw + ((size_t)q * HVp + tid) * 16);
        if (p.bn_scale) {
          const float4 sc = *reinterpret_cast<const float4 *>(s_bn + c_base + q * 4);
          const float4 sh = *reinterpret_cast<const float4 *>(s_bn + cpad + c_base + q * 4);
          x.x = x.x * sc.x + sh.x;
          x.y = x.y * sc.y + sh.y;
          x.z = x.z * sc.z + sh.z;
          x.w = x.w * sc.w + sh.w;
        }
        split4(x, h, l, amax);
      }
      *reinterpret_cast<uint2 *>(d) = h;
      *reinterpret_cast<uint2 *>(d + 8) = l;
    }
  };

  for (int c = 0; c < R - 1 && c < p.nchunks; c++) issue_chunk(c);
  for (int chunk = 0; chunk < p.nchunks; chunk++) {
    if (chunk > 0) __syncthreads();  // every wave is through K loop chunk - 1: the tile and ring slot (chunk - 1) % R are free
    if (chunk + R - 1 < p.nchunks) issue_chunk(chunk + R - 1);
    h2_wait_vm(opc * (min(chunk + R - 1, p.nchunks - 1) - chunk));  // this wave's part of chunk `chunk` has landed
    __syncthreads();                                                // ... and everybody's
    if (!(p.h2_dbg & 2048)) commit(chunk);  // (timing only)
    __syncthreads();
    if (!wave_has_cells || (p.h2_dbg & 256)) continue;  // (256: timing only, no K loop)
    const int nq = min(NQ, p.cin4 - chunk * NQ);
    const int cc8_here = min(CC8, (nq + 1) >> 1);
    const int NS = (cc8_here * taps + 3) >> 2;
    const char *wbuf = s_ring + (chunk % R) * slot_bytes + raw_bytes + lane * 16;
    // One wave per SIMD and nothing else on the CU: the K loop is a chain of LDS round trips unless the operands of a step
    // are requested several steps ahead -- a ring of four register sets, three steps ahead, the loop unrolled over the (at
    // most kH16RingSteps) steps of a chunk so that the tap offsets (qo, in registers) and the sets are addressed statically
    uint4 ah[4][TM], al[4][TM], wh[4], wl[4];
    auto load_step = [&](int st, int set) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < TM; m++) {
        const char *a = reinterpret_cast<const char *>(s_tile) + baseA[m] + qo[st];
        ah[set][m] = *reinterpret_cast<const uint4 *>(a);
        al[set][m] = *reinterpret_cast<const uint4 *>(a + 16);
      }
      wh[set] = *reinterpret_cast<const uint4 *>(wbuf + st * 2048);
      wl[set] = *reinterpret_cast<const uint4 *>(wbuf + st * 2048 + 1024);
    };
    auto mfma_step = [&](int set) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < TM; m++)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al[set][m]), __builtin_bit_cast(f16x8, wh[set]), acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < TM; m++)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[set][m]), __builtin_bit_cast(f16x8, wl[set]), acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < TM; m++)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[set][m]), __builtin_bit_cast(f16x8, wh[set]), acc[m], 0, 0, 0);
    };
#pragma unroll
    for (int st = 0; st < 3; st++)
      if (st < NS) load_step(st, st);
#pragma unroll
    for (int st = 0; st < kH16RingSteps; st++) {
      if (st < NS) {
        if (st + 3 < kH16RingSteps && st + 3 < NS) load_step(st + 3, (st + 3) & 3);
        mfma_step(st & 3);
      }
    }
  }

  // epilogue (conv3d_h2_16_kernel's)
  const float unscale = p.h2_unscale;
  float *out_b = p.out + (size_t)b * S * S * S * p.out_cs + p.out_c0;
  const int ncx = S / 2;
  const int ch = row;
  if (ch < p.cout) {
    const float bias = bias_pre;
#pragma unroll
    for (int m = 0; m < TM; m++) {
      const int cell = (wm * TM + m) * 2 + (kg >> 1);
      if (cell >= NC) continue;
      const int cz = cell % p.tcz, cy = (cell / p.tcz) % p.tcy, cx = cell / (p.tcz * p.tcy);
      const int gcx = tx * p.tcx + cx, gcy = ty * p.tcy + cy, gcz = tz * p.tcz + cz;
      if (gcx >= ncx || gcy >= ncx || gcz >= ncx) continue;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int vx = 2 * gcx + (kg & 1), vy = 2 * gcy + (r >> 1), vz = 2 * gcz + (r & 1);
        float v = acc[m][r] * unscale + bias;
        if (p.relu) v = fmaxf(v, 0.f);
        out_b[(((size_t)vx * S + vy) * S + vz) * p.out_cs + ch] = v;
      }
    }
  }
  h2_report_overflow(p.h2_overflow, !(amax <= 65504.f));
}

// ---------------------------------------------------------------------------------------------
// conv3d_h2_16_pc_kernel (round 6): conv3d_h2_16_ring_kernel for tiles of at most four cells and 128 halo voxels -- the
// 1 x 1 x 3-cell tiles of the 6^3 layers, where waves 0-1 hold the (one and a half) M-tiles and waves 2-3 have none.  There
// the ring kernel's chunk was: staging by all waves, barrier, a K loop of two waves (42 dependent MFMAs on one accumulator,
// ~1 us, nothing else to issue), barrier.  Here the idle waves are the PRODUCERS: while waves 0-1 run the K loop of chunk c
// on one tile buffer, waves 2-3 (thread 128 + v owns halo voxel v) stage chunk c + 1 into the other -- one barrier per chunk.
// Raw channels are requested three chunks ahead and weights two (each is needed two iterations after its request), three
// slots each.  Same values into the same MFMAs in the same order: same bits.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void conv3d_h2_16_pc_kernel(ConvArgs p) {
  constexpr int NTHREADS = 256, TM = 1, RS = 3;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane >> 4, row = lane & 15;
  const bool producer = wm >= 2;

  const int tiles_per_pose = p.ntx * p.nty * p.ntz;
  const int wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
  const int b = wg / tiles_per_pose;
  int t = wg - b * tiles_per_pose;
  const int tz = t % p.ntz;
  t /= p.ntz;
  const int ty = t % p.nty, tx = t / p.nty;

  const int HX = 2 * p.tcx + 2, HY = 2 * p.tcy + 2, HZ = 2 * p.tcz + 2;
  const int HV = HX * HY * HZ;  // <= 128: the launcher
  const int CC8 = p.cc4, CCs = p.ccs;
  const int SZ = CCs, SY = HZ * SZ + 8 * p.h2_pad_y, SX = HY * SY + 8 * p.h2_pad_x;  // fp16 elements
  constexpr int taps = 27;
  const int Qmax = taps * CC8;
  const int Smax = (Qmax + 3) >> 2;
  const int NQ = 2 * CC8;
  const int HVp = (HV + 63) & ~63;
  const int n_raw = (NQ * HVp + 255) >> 8, n_w = (2 * Smax + 3) >> 2;
  const int raw_bytes = n_raw * 4096, w_bytes = n_w * 4096;
  const int n = p.nchunks;

  extern __shared__ __attribute__((aligned(16))) _Float16 smem_h2[];
  const int tile_bytes = (int)((((size_t)HX * SX + 7) & ~(size_t)7) * sizeof(_Float16));
  char *const s_base = reinterpret_cast<char *>(smem_h2);
  int *s_qoff = reinterpret_cast<int *>(s_base + 2 * tile_bytes);
  int *s_vox = s_qoff + ((Qmax + 8 + 3) & ~3);
  const int cpad = n * CC8 * 8;
  const int bn_off = (2 * tile_bytes + ((Qmax + 8 + 3) & ~3) * 4 + (HVp + 4) * 4 + 1023) & ~1023;
  float *const s_bn = reinterpret_cast<float *>(s_base + bn_off);
  const int raw_off = bn_off + (int)(((size_t)2 * cpad * sizeof(float) + 1023) & ~(size_t)1023);
  const int w_off = raw_off + RS * raw_bytes;
  if (p.bn_scale)
    for (int c = tid; c < cpad; c += NTHREADS) {
      const bool has = c < p.cin4 * 4;
      s_bn[c] = has ? p.bn_scale[c] : 1.f;
      s_bn[cpad + c] = has ? p.bn_shift[c] : 0.f;
    }
  const float bias_pre = p.bias[row < p.cout ? row : 0];
  for (int q = tid; q < Qmax + 8; q += NTHREADS) {
    const int qq = q < Qmax ? q : Qmax - 1;
    const int c8 = qq / taps, tap = qq - c8 * taps;
    const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    s_qoff[q] = ((dx * SX + dy * SY + dz * SZ) + c8 * 16) * 2;
  }
  if (p.h2_dbg & 8192) return;  // (timing only: the launch and the first lines of the prologue)

  const int NC = p.tcx * p.tcy * p.tcz;  // <= 4: waves 0-1 cover them
  const int oz = row & 1, oy = (row >> 1) & 1, ox = (row >> 2) & 1, cell_in_mt = row >> 3;
  int baseA;
  {
    int cell = wm * 2 + cell_in_mt;
    if (cell >= NC) cell = 0;
    const int cz = cell % p.tcz, cy = (cell / p.tcz) % p.tcy, cx = cell / (p.tcz * p.tcy);
    baseA = ((2 * cx + ox) * SX + (2 * cy + oy) * SY + (2 * cz + oz) * SZ) * 2;  // bytes
  }
  h2_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float amax = 0.f;
  const bool wave_has_cells = wm * 2 < NC;

  const int S = p.S;
  const int x0 = tx * 2 * p.tcx - 1, y0 = ty * 2 * p.tcy - 1, z0 = tz * 2 * p.tcz - 1;
  const unsigned inv_hz = ((1u << 20) + HZ - 1) / HZ, inv_hy = ((1u << 20) + HY - 1) / HY;
  // thread 128 + v owns halo voxel v; the zero padding is laid down once, in both tile buffers
  const int hv_mine = tid - 128;
  int st_dst = 0, my_off = -1;
  if (hv_mine >= 0 && hv_mine < HV) {
    const int t1 = (int)(((unsigned)hv_mine * inv_hz) >> 20), hz = hv_mine - t1 * HZ;
    const int hx = (int)(((unsigned)t1 * inv_hy) >> 20), hy = t1 - hx * HY;
    const int x = x0 + hx, y = y0 + hy, z = z0 + hz;
    const bool in = (unsigned)x < (unsigned)S && (unsigned)y < (unsigned)S && (unsigned)z < (unsigned)S;
    my_off = in ? ((x * S + y) * S + z) * p.in_cs : -1;
    st_dst = hx * SX + hy * SY + hz * SZ;
    if (!in)
      for (int c = 0; c < CCs; c += 8) {
        *reinterpret_cast<uint4 *>(s_base + (st_dst + c) * 2) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4 *>(s_base + tile_bytes + (st_dst + c) * 2) = make_uint4(0u, 0u, 0u, 0u);
      }
  }
  if (tid < HVp) s_vox[tid] = -1;
  __syncthreads();
  if (hv_mine >= 0 && hv_mine < HVp) s_vox[hv_mine] = my_off;
  __syncthreads();  // s_vox, s_qoff

  int qo[kH16RingSteps];
#pragma unroll
  for (int st = 0; st < kH16RingSteps; st++) qo[st] = s_qoff[min(4 * st + kg, Qmax + 7)];
  unsigned voff[kH16RingMaxRaw];
#pragma unroll
  for (int i = 0; i < kH16RingMaxRaw; i++) {
    const int j = tid + i * NTHREADS;
    const int q = j / HVp, hv = j - q * HVp;
    const int off = (i < n_raw && q < NQ) ? s_vox[hv] : -1;
    voff[i] = off >= 0 ? (unsigned)(off + q * 4) * 4u : 0x80000000u;
  }
  typedef int h16_i32x4 __attribute__((ext_vector_type(4)));
  auto make_rsrc = [&](const void *base, unsigned bytes) __attribute__((always_inline)) {
    const unsigned long long a = (unsigned long long)base;
    h16_i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull)), rs.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffull));
    rs.z = __builtin_amdgcn_readfirstlane((int)bytes), rs.w = 0x00020000;
    return rs;
  };
  const size_t pose_floats = (size_t)S * S * S * p.in_cs;
  const h16_i32x4 rsrc_in = make_rsrc(p.in + (size_t)b * pose_floats, (unsigned)(pose_floats * 4));
  const int wchunk = Smax * 2048;
  const h16_i32x4 rsrc_w = make_rsrc(p.wp, (unsigned)(n * wchunk));
  const unsigned wl32 = (unsigned)lane * 32u;
  // (inline asm DMAs, counted by hand: conv3d_h2_16_ring_kernel)
  auto issue_raw = [&](int chunk) __attribute__((always_inline)) {
    if (chunk >= n) return;
    unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane(raw_off + (chunk % RS) * raw_bytes + wm * 1024);
    const unsigned soff_in = (unsigned)__builtin_amdgcn_readfirstlane(chunk * CC8 * 32);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=&s"(keep));
#pragma unroll
    for (int i = 0; i < kH16RingMaxRaw; i++)
      if (i < n_raw) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(m0v), "v"(voff[i]), "s"(rsrc_in), "s"(soff_in) : "memory");
        m0v += 0x1000u;
      }
    asm volatile("s_mov_b32 m0, %0" : : "s"(keep) : "memory");
  };
  auto issue_w = [&](int chunk) __attribute__((always_inline)) {
    if (chunk >= n) return;
    unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane(w_off + (chunk % RS) * w_bytes + wm * 1024);
    unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(chunk * wchunk + (wm >> 1) * 2048 + (wm & 1) * 16);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=&s"(keep));
    for (int i = 0; i < n_w; i++) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(m0v), "v"(wl32), "s"(rsrc_w), "s"(soff) : "memory");
      m0v += 0x1000u;
      soff += 4096u;
    }
    asm volatile("s_mov_b32 m0, %0" : : "s"(keep) : "memory");
  };
  // VMEM operations of "iteration i" (i >= -2): the raw channels of chunk i + 3 and the weights of chunk i + 2
  auto ops_of = [&](int i) { return (i + 3 < n ? n_raw : 0) + (i + 2 < n ? n_w : 0); };

  auto commit = [&](int chunk) __attribute__((always_inline)) {  // producers: chunk -> tile buffer chunk & 1
    if (my_off < 0) return;
    const int c_base = chunk * CC8 * 8;
    const int nq = min(NQ, p.cin4 - chunk * NQ);
    const char *raw = s_base + raw_off + (chunk % RS) * raw_bytes;
    _Float16 *dst = reinterpret_cast<_Float16 *>(s_base + (chunk & 1) * tile_bytes) + st_dst;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (q >= NQ) continue;
      _Float16 *d = dst + (q >> 1) * 16 + (q & 1) * 4;
      uint2 h = make_uint2(0u, 0u), l = make_uint2(0u, 0u);
      if (q < nq) {
        float4 x = *reinterpret_cast<const float4 *>(raw + ((size_t)q * HVp + hv_mine) * 16);
        if (p.bn_scale) {
          const float4 sc = *reinterpret_cast<const float4 *>(s_bn + c_base + q * 4);
          const float4 sh = *reinterpret_cast<const float4 *>(s_bn + cpad + c_base + q * 4);
          x.x = x.x * sc.x + sh.x;
          x.y = x.y * sc.y + sh.y;
          x.z = x.z * sc.z + sh.z;
          x.w = x.w * sc.w + sh.w;
        }
        split4(x, h, l, amax);
      }
      *reinterpret_cast<uint2 *>(d) = h;
      *reinterpret_cast<uint2 *>(d + 8) = l;
    }
  };

  // prologue: raw 0 | raw 1, weights 0 ("iteration -2") | raw 2, weights 1 ("iteration -1"); chunk 0 is staged before the loop
  issue_raw(0);
  issue_raw(1), issue_w(0);
  issue_raw(2), issue_w(1);
  h2_wait_vm(ops_of(-2) + ops_of(-1));
  __syncthreads();
  if (producer) commit(0);
  for (int chunk = 0; chunk < n; chunk++) {
    // raw channels of chunk + 1 and weights of `chunk` (requested two iterations ago) have landed: this wave's part ...
    h2_wait_vm(ops_of(chunk - 1));
    __syncthreads();  // ... and everybody's; tile buffer chunk & 1 is staged, the K loop of chunk - 1 and the staging of `chunk` are over
    issue_raw(chunk + 3), issue_w(chunk + 2);
    if (producer) {
      if (chunk + 1 < n) commit(chunk + 1);
      continue;
    }
    if (!wave_has_cells) continue;
    const int nq = min(NQ, p.cin4 - chunk * NQ);
    const int cc8_here = min(CC8, (nq + 1) >> 1);
    const int NS = (cc8_here * taps + 3) >> 2;
    const char *wbuf = s_base + w_off + (chunk % RS) * w_bytes + lane * 16;
    const char *tile = s_base + (chunk & 1) * tile_bytes;
    uint4 ah[4], al[4], wh[4], wl[4];
    auto load_step = [&](int st, int set) __attribute__((always_inline)) {
      const char *a = tile + baseA + qo[st];
      ah[set] = *reinterpret_cast<const uint4 *>(a);
      al[set] = *reinterpret_cast<const uint4 *>(a + 16);
      wh[set] = *reinterpret_cast<const uint4 *>(wbuf + st * 2048);
      wl[set] = *reinterpret_cast<const uint4 *>(wbuf + st * 2048 + 1024);
    };
#pragma unroll
    for (int st = 0; st < 3; st++)
      if (st < NS) load_step(st, st);
#pragma unroll
    for (int st = 0; st < kH16RingSteps; st++) {
      if (st < NS) {
        if (st + 3 < kH16RingSteps && st + 3 < NS) load_step(st + 3, (st + 3) & 3);
        const int set = st & 3;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al[set]), __builtin_bit_cast(f16x8, wh[set]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[set]), __builtin_bit_cast(f16x8, wl[set]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[set]), __builtin_bit_cast(f16x8, wh[set]), acc, 0, 0, 0);
      }
    }
  }

  // epilogue (conv3d_h2_16_kernel's)
  const float unscale = p.h2_unscale;
  float *out_b = p.out + (size_t)b * S * S * S * p.out_cs + p.out_c0;
  const int ncx = S / 2;
  const int ch = row;
  if (ch < p.cout && wave_has_cells) {
    const int cell = wm * 2 + (kg >> 1);
    if (cell < NC) {
      const int cz = cell % p.tcz, cy = (cell / p.tcz) % p.tcy, cx = cell / (p.tcz * p.tcy);
      const int gcx = tx * p.tcx + cx, gcy = ty * p.tcy + cy, gcz = tz * p.tcz + cz;
      if (gcx < ncx && gcy < ncx && gcz < ncx) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int vx = 2 * gcx + (kg & 1), vy = 2 * gcy + (r >> 1), vz = 2 * gcz + (r & 1);
          float v = acc[r] * unscale + bias_pre;
          if (p.relu) v = fmaxf(v, 0.f);
          out_b[(((size_t)vx * S + vy) * S + vz) * p.out_cs + ch] = v;
        }
      }
    }
  }
  h2_report_overflow(p.h2_overflow, !(amax <= 65504.f));
}

// LDS of conv3d_h2_16_pc_kernel; 0 = the launch is outside what the kernel covers
size_t conv_h2_16_pc_lds_bytes(const ConvArgs &p) {
  if (p.ksize != 3 || p.coutp != 16 || p.cc4 < 1 || p.cc4 > 2) return 0;
  const size_t HX = 2 * p.tcx + 2, HY = 2 * p.tcy + 2, HZ = 2 * p.tcz + 2, HV = HX * HY * HZ;
  if (HV > 128 || p.tcx * p.tcy * p.tcz > 4) return 0;
  const size_t HVp = (HV + 63) & ~(size_t)63, NQ = 2 * p.cc4;
  const size_t n_raw = (NQ * HVp + 255) >> 8;
  const int Q = 27 * p.cc4, Smax = (Q + 3) >> 2;
  const size_t n_w = (2 * (size_t)Smax + 3) >> 2;
  if (n_raw > (size_t)kH16RingMaxRaw || 2 * (n_raw + n_w) > 62) return 0;
  const size_t SY = HZ * p.ccs + 8 * p.h2_pad_y, SX = HY * SY + 8 * p.h2_pad_x;
  const size_t tile_bytes = ((HX * SX + 7) & ~(size_t)7) * sizeof(_Float16);
  const size_t bn_off = (2 * tile_bytes + (size_t)((Q + 8 + 3) & ~3) * 4 + (HVp + 4) * 4 + 1023) & ~(size_t)1023;
  const size_t bn_bytes = ((size_t)2 * p.nchunks * p.cc4 * 8 * sizeof(float) + 1023) & ~(size_t)1023;
  return bn_off + bn_bytes + 3 * (n_raw + n_w) * 4096;
}

// LDS of conv3d_h2_16_ring_kernel with `ring` slots; 0 = the launch is outside what the kernel covers
size_t conv_h2_16_ring_lds_bytes(const ConvArgs &p, int ring) {
  if (p.ksize != 3 || p.coutp != 16 || p.cc4 < 1 || p.cc4 > 2 || ring < 2) return 0;
  const size_t HX = 2 * p.tcx + 2, HY = 2 * p.tcy + 2, HZ = 2 * p.tcz + 2, HV = HX * HY * HZ;
  if (HV > 256) return 0;
  const size_t HVp = (HV + 63) & ~(size_t)63, NQ = 2 * p.cc4;
  const size_t n_raw = (NQ * HVp + 255) >> 8;
  if (n_raw > (size_t)kH16RingMaxRaw) return 0;
  const int Q = 27 * p.cc4, Smax = (Q + 3) >> 2;
  const size_t n_w = (2 * (size_t)Smax + 3) >> 2;
  if ((n_raw + n_w) * (size_t)(ring - 1) > 62) return 0;  // (vmcnt counts to 63)
  const size_t SY = HZ * p.ccs + 8 * p.h2_pad_y, SX = HY * SY + 8 * p.h2_pad_x;
  const size_t bn_bytes = ((size_t)2 * p.nchunks * p.cc4 * 8 * sizeof(float) + 1023) & ~(size_t)1023;
  return h2_16_main_lds_bytes(HX, SX, Q, HVp) + bn_bytes + (size_t)ring * (n_raw + n_w) * 4096;
}

// geometry of conv3d_h2_kernel's planar halo tile (16-byte slots): z-row and x-plane strides, slots per plane
void conv_h2_planar_geo(const ConvArgs &p, int *sy, int *sx, int *pl) {
  const int HX = 2 * p.tcx + 2, HY = 2 * p.tcy + 2, HZ = 2 * p.tcz + 2;
  *sy = HZ + p.h2_pad_y;
  *sx = HY * *sy + p.h2_pad_x;
  *pl = (HX * *sx + 31) & ~31;
}

size_t conv_h2_lds_bytes(const ConvArgs &p) {
  const size_t mid_bytes = p.post_w ? (size_t)p.post_rows * (2 * p.coutp + 8) * sizeof(_Float16) : 0;  // fused 1x1x1 conv
  if (p.ksize == 3 && p.coutp != 16) {  // conv3d_h2_kernel: two buffers of two planes, tap offsets, live flags
    int sy, sx, pl;
    conv_h2_planar_geo(p, &sy, &sx, &pl);
    // (h2_wlds: one buffer and the chunk's weights, 14 steps of 2 KB; else two buffers)
    const size_t tiles = p.h2_wlds ? (size_t)2 * pl * 16 + 14 * 2048 : (size_t)4 * pl * 16;
    return std::max(tiles + 32 * sizeof(int) + (size_t)p.nchunks * 4 * sizeof(int) + 256 /* prefetch_tile's junk region */, mid_bytes);
  }
  const int halo = p.ksize == 3 ? 1 : 0;
  const size_t HX = 2 * p.tcx + 2 * halo, HY = 2 * p.tcy + 2 * halo, HZ = 2 * p.tcz + 2 * halo, HV = HX * HY * HZ;
  const int Q = (p.ksize == 3 ? 27 : 1) * p.cc4;
  const size_t SY = HZ * p.ccs + 8 * p.h2_pad_y, SX = HY * SY + 8 * p.h2_pad_x;  // (pads: 16-wide kernel only)
  if (p.coutp == 16 && p.h2_wlds)  // conv3d_h2_16_kernel<TM, true>: two buffers of a chunk's packed weights behind the rest
    return h2_16_main_lds_bytes(HX, SX, Q, HV) + (size_t)2 * ((Q + 3) / 4) * 2048;
  const size_t main_bytes = ((HX * SX + 7) & ~(size_t)7) * sizeof(_Float16) + (size_t)((Q + 8 + 3) & ~3) * sizeof(int) + (HV + 4) * sizeof(int);
  return std::max(main_bytes, mid_bytes);
}

// 1x1x1 layers (conv3d_h2_k1_kernel; no zero test: one step per octet pair, nothing to skip ahead of)
template <int WM, int WN, int TM, int TN> static void launch_h2_k1(const ConvArgs &p, int B, hipStream_t s) {
  const int ngroups = (p.coutp / 32 + WN * TN - 1) / (WN * TN);
  dim3 grid(B * p.ntx * p.nty * p.ntz, ngroups), block(64 * WM * WN);
  const bool bwd = p.in_amax || p.out_amax || p.out_mask || p.in_mode == 2 || p.out_scale;
  if (bwd) {  // a transposed conv of the gradient pass: the shapes of conv_h2_has_bwd_k1
    if (p.accumulate || (p.in_mode != 0 && p.in_mode != 2)) throw Error(2, "launch_conv_h2: 1x1x1 gradient-pass launch: no accumulation, plain or max-unpooled input");
    if constexpr (WM == 4 && WN == 1 && TM == 1) {
      auto kern = conv3d_h2_k1_kernel<WM, WN, TM, TN, false, false, true, true>;
      ensure_max_lds(reinterpret_cast<const void *>(kern), 160 * 1024);
      hipLaunchKernelGGL(kern, grid, block, conv_h2_lds_bytes(p), s, p);
      return;
    }
    throw Error(2, "launch_conv_h2: 1x1x1 gradient-pass variant not compiled for this tile shape");
  }
  auto kern = conv3d_h2_k1_kernel<WM, WN, TM, TN, false, false, true>;
  ensure_max_lds(reinterpret_cast<const void *>(kern), 160 * 1024);
  hipLaunchKernelGGL(kern, grid, block, conv_h2_lds_bytes(p), s, p);
}

// 3x3x3 layers (conv3d_h2_kernel).  MT = the M-tile geometries compiled for this shape besides raster order (ConvArgs::mt_x);
// SKIP_OK: the zero-skipping variant exists (the throughput tile of a first conv: the pooled voxel grid)
template <int WM, int WN, int TM, int MTMASK, bool SKIP_OK> static void launch_h2_k3(ConvArgs p, int B, hipStream_t s) {
  p.nposes = B;
  const int ngroups = (p.coutp / 32 + WN - 1) / WN;
  dim3 grid(B * p.ntx * p.nty * p.ntz, ngroups), block(64 * WM * WN);
  const size_t lds = conv_h2_lds_bytes(p);
  int sy, sx, pl;
  conv_h2_planar_geo(p, &sy, &sx, &pl);
  if (2 * pl > kH2NS * 64 * WM * WN || (2 * p.tcx + 2) * (2 * p.tcy + 2) * (2 * p.tcz + 2) > kH2VPT * 64 * WM * WN)
    throw Error(2, "launch_conv_h2: halo tile larger than the kernel's staging covers");
  if (p.in_split && (p.in_cs % 8 || p.bn_scale)) throw Error(2, "launch_conv_h2: split-format input needs whole octets and no BatchNorm");
  if (p.out_split && (p.out_cs % 8 || p.out_c0 || p.coutp != p.cout || p.argmax_out))
    throw Error(2, "launch_conv_h2: split-format output needs whole octets, no channel offset, no arg-max");
  auto go = [&](auto kern) {
    ensure_max_lds(reinterpret_cast<const void *>(kern), 160 * 1024);
    hipLaunchKernelGGL(kern, grid, block, lds, s, p);
  };
  // a transposed conv of the gradient pass: the weights-in-LDS shapes only (conv_h2_has_bwd)
  const bool bwd = p.in_amax || p.out_amax || p.out_mask || p.in_mode == 2 || p.out_scale || p.accumulate;
  if (bwd && (p.in_split || p.out_split || !p.h2_wlds)) throw Error(2, "launch_conv_h2: gradient-pass launch needs fp32 tensors and the weights-in-LDS variant");
  auto by_input = [&](auto mt, auto skip) {
    constexpr int MT = decltype(mt)::value;
    constexpr bool SK = decltype(skip)::value;
    if constexpr (WN == 1 && TM <= 2) {  // (weights through LDS: the shapes whose four waves share their output channels)
      if (p.h2_wlds) {
        if constexpr (TM == 2) {  // (the throughput shape) two poses per workgroup on one copy of the weights
          if constexpr (SK && MT == 1 && WM == 4)
            if (p.in_split && p.h2_ws > 0 && B >= 32 && conv_h2_ws_covers(p, B)) {  // stationary weights, ring of tiles (conv3d_h2_ws.hip)
              launch_conv_h2_ws(p, B, p.h2_ws, s);
              return;
            }
          if constexpr (SK && MT == 1 && WM == 4)
            if (p.in_split && p.h2_wlds >= 4 && B >= 4 && !p.post_w) {  // (experiment: four poses per workgroup on one copy of the weights)
              grid.x = (unsigned)((B + 3) / 4 * p.ntx * p.nty * p.ntz);
              go(conv3d_h2_kernel<WM, WN, TM, MT, SK, true, true, 4>);
              return;
            }
          if (p.in_split && p.h2_wlds >= 2 && B >= 2) {
            grid.x = (unsigned)((B + 1) / 2 * p.ntx * p.nty * p.ntz);
            go(conv3d_h2_kernel<WM, WN, TM, MT, SK, true, true, 2>);
            return;
          }
        }
        if (p.in_split) go(conv3d_h2_kernel<WM, WN, TM, MT, SK, true, true>);
        else if (bwd) go(conv3d_h2_kernel<WM, WN, TM, MT, SK, false, true, 1, true>);
        else go(conv3d_h2_kernel<WM, WN, TM, MT, SK, false, true>);
        return;
      }
    }
    if (bwd) throw Error(2, "launch_conv_h2: gradient-pass variant not compiled for this tile shape");
    if (p.h2_wlds) throw Error(2, "launch_conv_h2: weights-in-LDS variant not compiled for this tile shape");
    if (p.in_split) go(conv3d_h2_kernel<WM, WN, TM, MT, SK, true>);
    else go(conv3d_h2_kernel<WM, WN, TM, MT, SK, false>);
  };
  auto by_skip = [&](auto mt) {
    if constexpr (SKIP_OK) {
      if (p.sparse) return by_input(mt, std::true_type{});
    }
    by_input(mt, std::false_type{});
  };
  bool launched = false;
  if (p.mt_x == 0) by_skip(std::integral_constant<int, 0>{}), launched = true;
  if constexpr ((MTMASK & 2) != 0)
    if (p.mt_x == 1) by_skip(std::integral_constant<int, 1>{}), launched = true;
  if constexpr ((MTMASK & 4) != 0)
    if (p.mt_x == 2) by_skip(std::integral_constant<int, 2>{}), launched = true;
  if (!launched) throw Error(2, "launch_conv_h2: M-tile geometry not compiled for this tile shape");
}

template <int TM> static void launch_h2_16(ConvArgs p, int B, hipStream_t s) {
  dim3 grid(B * p.ntx * p.nty * p.ntz), block(256);
  // weights through LDS (WL) for the 3x3x3 layers at 6^3 (h2_wlds: 0 = never, 1 = those, 2 = every 3x3x3 layer the buffers fit;
  // + 4: no ring kernel for launches of few small workgroups)
  const int mode = p.h2_wlds & 3;
  const bool ring_ok = p.h2_wlds > 0 && !(p.h2_wlds & 4), pc_ok = !(p.h2_wlds & 8);
  p.h2_wlds = 0;
  // (a launch that fills the chip hides the weights' latency behind its other workgroups, and the buffers cost it one of
  // them per CU: Dense at 1,024 poses per step 68.4 k poses/s without, 67.1 k with)
  if (p.ksize == 3 && mode > 0 && (mode >= 2 || (p.S <= 6 && grid.x <= 512))) {
    p.h2_wlds = 1;
    if (conv_h2_lds_bytes(p) > 160 * 1024) p.h2_wlds = 0;
  }
  // a few small workgroups (the 6^3 layers of a per-pose call): operands through a ring, requested chunks ahead; tiles whose
  // cells fit two waves take the producer / consumer form (h2_wlds + 8: not)
  if constexpr (TM == 1)
    if (p.ksize == 3 && ring_ok && pc_ok && grid.x <= 256 && !p.accumulate) {
      const size_t lds = conv_h2_16_pc_lds_bytes(p);
      if (lds != 0 && lds <= 160 * 1024) {
        ensure_max_lds(reinterpret_cast<const void *>(conv3d_h2_16_pc_kernel), 160 * 1024);
        hipLaunchKernelGGL(conv3d_h2_16_pc_kernel, grid, block, lds, s, p);
        return;
      }
    }
  if constexpr (TM <= 2)
    if (p.ksize == 3 && ring_ok && grid.x <= 256 && !p.accumulate) {
      for (int ring = 4; ring >= 3; ring--) {
        const size_t lds = conv_h2_16_ring_lds_bytes(p, ring);
        if (lds == 0 || lds > 160 * 1024) continue;
        p.h16_ring = std::min(ring, std::max(2, p.nchunks));
        ensure_max_lds(reinterpret_cast<const void *>(conv3d_h2_16_ring_kernel<TM>), 160 * 1024);
        hipLaunchKernelGGL((conv3d_h2_16_ring_kernel<TM>), grid, block, lds, s, p);
        return;
      }
    }
  if (p.h2_wlds) {
    ensure_max_lds(reinterpret_cast<const void *>(conv3d_h2_16_kernel<TM, true>), 160 * 1024);
    hipLaunchKernelGGL((conv3d_h2_16_kernel<TM, true>), grid, block, conv_h2_lds_bytes(p), s, p);
    return;
  }
  ensure_max_lds(reinterpret_cast<const void *>(conv3d_h2_16_kernel<TM>), 160 * 1024);
  hipLaunchKernelGGL((conv3d_h2_16_kernel<TM>), grid, block, conv_h2_lds_bytes(p), s, p);
}

bool conv_h2_has_cfg(int cfg) {
  switch (cfg) {
    case CONV_CFG_4x1_2x1:
    case CONV_CFG_2x2_3x1:
    case CONV_CFG_1x4_7x1:
    case CONV_CFG_4x1_1x3:
    case CONV_CFG_4x1_1x5:
    case CONV_CFG_4x1_1x1:
    case CONV_CFG_N16_TM1:
    case CONV_CFG_N16_TM2:
    case CONV_CFG_N16_TM3:
    case CONV_CFG_N16_TM4: return true;
    default: return false;
  }
}

// tile shapes whose gradient-pass (transposed conv) variant is compiled
bool conv_h2_has_bwd(int cfg) { return cfg == CONV_CFG_4x1_2x1 || cfg == CONV_CFG_4x1_1x1; }
bool conv_h2_has_bwd_k1(int cfg) { return cfg == CONV_CFG_4x1_1x1 || cfg == CONV_CFG_4x1_1x3 || cfg == CONV_CFG_4x1_1x5; }

int conv_h2_mt_mask(int cfg) {
  switch (cfg) {
    case CONV_CFG_4x1_2x1: return 1 | 2 | 4;
    case CONV_CFG_2x2_3x1: return 1 | 4;
    case CONV_CFG_4x1_1x1: return 1 | 4;
    default: return 1;
  }
}

void launch_conv_h2(const ConvArgs &p, int cfg, int B, hipStream_t s) {
  const bool k1 = p.ksize == 1;
  switch (cfg) {
    case CONV_CFG_4x1_2x1:
      if (k1) launch_h2_k1<4, 1, 2, 1>(p, B, s);
      else launch_h2_k3<4, 1, 2, 1 | 2 | 4, true>(p, B, s);
      break;
    case CONV_CFG_2x2_3x1:
      if (k1) launch_h2_k1<2, 2, 3, 1>(p, B, s);
      else launch_h2_k3<2, 2, 3, 1 | 4, false>(p, B, s);
      break;
    case CONV_CFG_1x4_7x1:
      if (k1) launch_h2_k1<1, 4, 7, 1>(p, B, s);
      else launch_h2_k3<1, 4, 7, 1, false>(p, B, s);
      break;
    case CONV_CFG_4x1_1x1:
      if (k1) launch_h2_k1<4, 1, 1, 1>(p, B, s);
      else launch_h2_k3<4, 1, 1, 1 | 4, false>(p, B, s);
      break;
    case CONV_CFG_4x1_1x3:
      if (!k1) throw Error(2, "launch_conv_h2: tile configuration compiled for 1x1x1 convolutions only");
      launch_h2_k1<4, 1, 1, 3>(p, B, s);
      break;
    case CONV_CFG_4x1_1x5:
      if (!k1) throw Error(2, "launch_conv_h2: tile configuration compiled for 1x1x1 convolutions only");
      launch_h2_k1<4, 1, 1, 5>(p, B, s);
      break;
    case CONV_CFG_N16_TM1: launch_h2_16<1>(p, B, s); break;
    case CONV_CFG_N16_TM2: launch_h2_16<2>(p, B, s); break;
    case CONV_CFG_N16_TM3: launch_h2_16<3>(p, B, s); break;
    case CONV_CFG_N16_TM4: launch_h2_16<4>(p, B, s); break;
    default: throw Error(2, "launch_conv_h2: tile configuration not compiled");
  }
}

}  // namespace mig
