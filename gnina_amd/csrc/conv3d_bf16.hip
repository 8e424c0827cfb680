// conv3d_bf16.hip -- the bf16-MFMA variant of the implicit-GEMM 3-D convolution (BASELINE config 5: Dense
// networks at 0.25 A / 96^3, "bf16 MFMA path").
//
// Same decomposition as conv3d.hip (M-tile = 32 voxels = four 2x2x2 pooling cells, so that ReLU + pooling
// stay register-local in the dtype-independent 32x32 accumulator layout), but on
// v_mfma_f32_32x32x16_bf16: K runs over OCTETS (one tap x 8 consecutive input channels); lanes 0-31 feed
// k = 0..7 of an instruction from octet 2p, lanes 32-63 k = 8..15 from octet 2p+1, one ds_read_b128 per
// lane per MFMA.  Activations live in HBM as bf16 channels-last (the pooled voxel grid and the tensors
// feeding the fully connected heads stay fp32: `in_f32` / `out_f32`), are converted while the halo tile is
// staged (eval BatchNorm is applied in fp32 first), and accumulate in fp32.  This is NOT the parity path:
// scores differ from the fp32 kernels by the bf16 rounding of activations and weights (measured in
// tests/test_gpu_bf16.py); the exact path is conv3d.hip.
//
// At the bf16 rate the MFMA pipe (32 cycles per instruction) is no longer the bound: every MFMA needs 1 KB of
// A operand from LDS (128 B/clk/CU = one MFMA per SIMD per 32 clk with TN = 1), and the 16-output-channel
// Dense-block layers fill half of a 32-wide tile, so the kernel is LDS-bandwidth / staging bound.
#include "common.h"
#include "conv3d.h"

namespace mig {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even (v_cvt_pk_bf16_f32)
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

__device__ __forceinline__ void load8(const void *base, size_t elem, int is_f32, float *v) {
  if (is_f32) {
    const float4 a = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + elem);
    const float4 b = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + elem + 4);
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
  } else {
    const uint4 r = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned short *>(base) + elem);
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      v[2 * i] = __builtin_bit_cast(float, w[i] << 16);
      v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
  }
}

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * WM * WN) void conv3d_bf16_kernel(ConvArgs p) {
  constexpr int NTHREADS = 64 * WM * WN;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
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
  const int CC8 = p.cc4, CCs = p.ccs;  // octets per K chunk / LDS elements per halo voxel
  const int taps = p.ksize == 3 ? 27 : 1;
  const int Q = taps * CC8;
  const int P = (Q + 1) >> 1;

  extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
  unsigned short *s_tile = smem_h;                                           // [HV][CCs]
  int *s_qoff = reinterpret_cast<int *>(smem_h + (((size_t)HV * CCs + 7) & ~(size_t)7));  // [Q]
  float *s_bn = reinterpret_cast<float *>(s_qoff + ((Q + 3) & ~3));                         // [2][CC8 * 8] scale, shift of the chunk
  int *s_vox = reinterpret_cast<int *>(s_bn + 2 * CC8 * 8);  // [HV] voxel index of every halo position inside the pose, -1 = padding

  for (int q = tid; q < Q; q += NTHREADS) {
    int tap = q / CC8, c8 = q - tap * CC8;
    int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    s_qoff[q] = (p.ksize == 3 ? ((dx * HY + dy) * HZ + dz) * CCs : 0) + c8 * 8;
  }

  const int S = p.S;
  const int x0 = tx * 2 * p.tcx - halo, y0 = ty * 2 * p.tcy - halo, z0 = tz * 2 * p.tcz - halo;
  // halo position -> voxel, once per workgroup: the staging loops below then run without integer divisions
  for (int hv = tid; hv < HV; hv += NTHREADS) {
    const int hz = hv % HZ, hy = (hv / HZ) % HY, hx = hv / (HZ * HY);
    const int x = x0 + hx, y = y0 + hy, z = z0 + hz;
    const bool in = (unsigned)x < (unsigned)S && (unsigned)y < (unsigned)S && (unsigned)z < (unsigned)S;
    s_vox[hv] = !in ? -1 : (p.in_mode == 2 ? ((x >> 1) * (S >> 1) + (y >> 1)) * (S >> 1) + (z >> 1) + (((x & 1) << 2 | (y & 1) << 1 | (z & 1)) << 28)
                                           : (x * S + y) * S + z);
  }
  const unsigned inv_cc8 = ((1u << 20) + CC8 - 1) / CC8;  // it / CC8 == (it * inv_cc8) >> 20 for it < 2^20 / CC8

  const int NC = p.tcx * p.tcy * p.tcz;
  const int oz = row & 1, oy = (row >> 1) & 1, ox = (row >> 3) & 1;
  const int cell_in_mt = ((row >> 2) & 1) + 2 * ((row >> 4) & 1);
  int baseA[TM];
#pragma unroll
  for (int m = 0; m < TM; m++) {
    int cell = (wm * TM + m) * 4 + cell_in_mt;
    if (cell >= NC) cell = 0;
    int cz = cell % p.tcz, cy = (cell / p.tcz) % p.tcy, cx = cell / (p.tcz * p.tcy);
    baseA[m] = (((2 * cx + ox) * HY + (2 * cy + oy)) * HZ + (2 * cz + oz)) * CCs;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; m++)
#pragma unroll
    for (int n = 0; n < TN; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

  const unsigned short *wq = reinterpret_cast<const unsigned short *>(p.wp) + (size_t)(n_base + row) * 8;
  const size_t wstride = (size_t)p.coutp * 8;  // elements per octet row of the packed weights

  for (int chunk = 0; chunk < p.nchunks; chunk++) {
    __syncthreads();
    const int c_base = chunk * CC8 * 8;
    if (p.bn_scale) {
      for (int i = tid; i < CC8 * 8; i += NTHREADS) {
        s_bn[i] = p.bn_scale[c_base + i];
        s_bn[CC8 * 8 + i] = p.bn_shift[c_base + i];
      }
      __syncthreads();
    }
    // bf16 input, no mask: all global loads of a batch of U items per thread are issued before the first one
    // is consumed (a dependent load -> convert -> ds_write chain per item would serialise ~1-2k cycles of
    // L2/HBM latency per iteration, which at the bf16 MFMA rate is longer than the MFMA phase itself)
    const bool batched = !p.in_f32 && p.in_mode == 0;
    const int total_items = HV * CC8;
    constexpr int U = 4;
    for (int base = 0; batched && base < total_items; base += NTHREADS * U) {
      uint4 raw[U];
      int dst[U], cc[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int it = base + u * NTHREADS + tid;
        raw[u] = make_uint4(0u, 0u, 0u, 0u);
        dst[u] = -1;
        cc[u] = 0;
        if (it < total_items) {
          const int hv = (int)(((unsigned)it * inv_cc8) >> 20), c8 = it - hv * CC8;
          const int vi = s_vox[hv];
          const int c = c_base + c8 * 8;
          dst[u] = hv * CCs + c8 * 8;
          cc[u] = c8;
          if (vi >= 0 && c < p.cin4 * 8 && c < p.in_cs)
            raw[u] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned short *>(p.in) +
                                                   ((size_t)b * S * S * S + vi) * p.in_cs + c);
          else
            cc[u] = -1;  // padding stays exactly zero (PyTorch pads after the BatchNorm)
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (dst[u] < 0) continue;
        uint4 pk = raw[u];
        if (p.bn_scale && cc[u] >= 0) {
          const unsigned wd[4] = {pk.x, pk.y, pk.z, pk.w};
          float v[8];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            v[2 * i] = __builtin_bit_cast(float, wd[i] << 16);
            v[2 * i + 1] = __builtin_bit_cast(float, wd[i] & 0xffff0000u);
          }
          const float4 s0 = *reinterpret_cast<const float4 *>(s_bn + cc[u] * 8), s1 = *reinterpret_cast<const float4 *>(s_bn + cc[u] * 8 + 4);
          const float4 h0 = *reinterpret_cast<const float4 *>(s_bn + CC8 * 8 + cc[u] * 8),
                       h1 = *reinterpret_cast<const float4 *>(s_bn + CC8 * 8 + cc[u] * 8 + 4);
          pk.x = pack2(v[0] * s0.x + h0.x, v[1] * s0.y + h0.y);
          pk.y = pack2(v[2] * s0.z + h0.z, v[3] * s0.w + h0.w);
          pk.z = pack2(v[4] * s1.x + h1.x, v[5] * s1.y + h1.y);
          pk.w = pack2(v[6] * s1.z + h1.z, v[7] * s1.w + h1.w);
        }
        *reinterpret_cast<uint4 *>(s_tile + dst[u]) = pk;
      }
    }
    for (int it = tid; !batched && it < HV * CC8; it += NTHREADS) {
      const int hv = (int)(((unsigned)it * inv_cc8) >> 20), c8 = it - hv * CC8;
      const int vi = s_vox[hv];
      const int c = c_base + c8 * 8;
      const bool live = vi >= 0 && c < p.cin4 * 8 && c < p.in_cs;  // (an even octet count may pad past the tensor)
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (live) {
        if (p.in_mode == 2) {  // max-unpool on load (see conv3d.hip)
          const int Sh = S >> 1;
          const size_t cell = (size_t)b * Sh * Sh * Sh + (vi & 0x0fffffff);
          const int r = vi >> 28;
          float g[8], act[8];
          load8(p.in, cell * p.in_cs + c, p.in_f32, g);
          load8(p.in_act, cell * p.in_act_cs + c, p.act_f32, act);
          const uint2 am = *reinterpret_cast<const uint2 *>(p.in_argmax + cell * p.in_cs + c);
#pragma unroll
          for (int i = 0; i < 8; i++) {
            const unsigned a = ((i < 4 ? am.x : am.y) >> (8 * (i & 3))) & 0xffu;
            v[i] = ((int)a == r && act[i] > 0.f) ? g[i] : 0.f;
          }
        } else {
          const size_t vox = (size_t)b * S * S * S + vi;
          if (p.in_f32 && c + 8 > p.in_cs) {  // the fp32 voxel grid has 28 (or 36) channels: half an octet at the end
            const float4 a = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p.in) + vox * p.in_cs + c);
            v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w;
          } else {
            load8(p.in, vox * p.in_cs + c, p.in_f32, v);
          }
          if (p.in_mode == 1) {
            float act[8];
            load8(p.in_act, vox * p.in_act_cs + c, p.act_f32, act);
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = act[i] > 0.f ? v[i] : 0.f;
          }
          if (p.bn_scale) {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = v[i] * s_bn[c8 * 8 + i] + s_bn[CC8 * 8 + c8 * 8 + i];
          }
        }
      }
      uint4 pk;
      pk.x = pack2(v[0], v[1]);
      pk.y = pack2(v[2], v[3]);
      pk.z = pack2(v[4], v[5]);
      pk.w = pack2(v[6], v[7]);
      *reinterpret_cast<uint4 *>(s_tile + (size_t)hv * CCs + c8 * 8) = pk;
    }
    __syncthreads();

    // K loop over octet pairs, unrolled by two with two operand register sets (ping-pong): the LDS / L2 loads
    // of pair pr + 1 are in flight while the MFMAs of pair pr run.  (A single-set "load next, copy" loop is
    // folded by the compiler back into load -> wait -> MFMA, which exposes the full L2 latency per pair.)
    const unsigned short *wchunk = wq + (size_t)chunk * P * 2 * wstride + (size_t)kh * wstride;
    uint4 w0[TN], w1[TN], a0[TM], a1[TM];
    auto load_pair = [&](int pr, uint4 *aa, uint4 *ww) {
      const int qo = s_qoff[2 * pr + kh];  // Q is even (CC8 even): both half-waves always have a real octet
#pragma unroll
      for (int m = 0; m < TM; m++) aa[m] = *reinterpret_cast<const uint4 *>(s_tile + baseA[m] + qo);
#pragma unroll
      for (int n = 0; n < TN; n++)
        ww[n] = *reinterpret_cast<const uint4 *>(wchunk + (size_t)pr * 2 * wstride + (size_t)n * 32 * 8);
    };
    auto mfma_pair = [&](const uint4 *aa, const uint4 *ww) {
#pragma unroll
      for (int m = 0; m < TM; m++)
#pragma unroll
        for (int n = 0; n < TN; n++)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aa[m]),
                                                              __builtin_bit_cast(bf16x8, ww[n]), acc[m][n], 0, 0, 0);
    };
    load_pair(0, a0, w0);
    int pr = 0;
    for (; pr + 1 < P; pr += 2) {
      load_pair(pr + 1, a1, w1);
      mfma_pair(a0, w0);
      if (pr + 2 < P) load_pair(pr + 2, a0, w0);
      mfma_pair(a1, w1);
    }
    if (P & 1) mfma_pair(a0, w0);
  }

  // ---- epilogue: bias, ReLU, optional 2x2x2 pool, store channels-last (bf16 or fp32) ----
  const int So = p.pool ? S / 2 : S;
  const size_t out_pose = (size_t)b * So * So * So * p.out_cs + p.out_c0;
  float *out_f = reinterpret_cast<float *>(p.out) + out_pose;
  unsigned short *out_h = reinterpret_cast<unsigned short *>(p.out) + out_pose;
  const int ncx = S / 2;
#pragma unroll
  for (int m = 0; m < TM; m++) {
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int cell = (wm * TM + m) * 4 + kh + 2 * half;
      if (cell >= NC) continue;
      const int cz = cell % p.tcz, cy = (cell / p.tcz) % p.tcy, cx = cell / (p.tcz * p.tcy);
      const int gcx = tx * p.tcx + cx, gcy = ty * p.tcy + cy, gcz = tz * p.tcz + cz;
      if (gcx >= ncx || gcy >= ncx || gcz >= ncx) continue;
#pragma unroll
      for (int n = 0; n < TN; n++) {
        const int ch = n_base + n * 32 + row;
        if (ch >= p.cout) continue;
        const float bias = p.bias[ch];
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
          float tt = acc[m][n][half * 8 + r] + bias;
          v[r] = p.relu ? fmaxf(tt, 0.f) : tt;
        }
        if (p.pool == 1) {
          float mx = v[0];
          int am = 0;
#pragma unroll
          for (int r = 1; r < 8; r++)
            if (v[r] > mx) mx = v[r], am = r;
          const size_t o = (((size_t)gcx * So + gcy) * So + gcz) * p.out_cs + ch;
          if (p.out_f32) out_f[o] = mx;
          else out_h[o] = f2bf(mx);
          if (p.argmax_out) p.argmax_out[out_pose + o] = (unsigned char)am;
        } else if (p.pool == 2) {
          float s = v[0];
#pragma unroll
          for (int r = 1; r < 8; r++) s = s + v[r];
          const size_t o = (((size_t)gcx * So + gcy) * So + gcz) * p.out_cs + ch;
          if (p.out_f32) out_f[o] = s * 0.125f;
          else out_h[o] = f2bf(s * 0.125f);
        } else {
          const float osc = p.out_scale ? p.out_scale[ch] : 1.0f;
#pragma unroll
          for (int r = 0; r < 8; r++) {
            const int vx = 2 * gcx + (r >> 2), vy = 2 * gcy + ((r >> 1) & 1), vz = 2 * gcz + (r & 1);
            const size_t o = (((size_t)vx * So + vy) * So + vz) * p.out_cs + ch;
            float val = p.out_scale ? v[r] * osc : v[r];
            if (p.out_f32) {
              if (p.accumulate) val = out_f[o] + val;
              out_f[o] = val;
            } else {
              if (p.accumulate) val = bf2f(out_h[o]) + val;
              out_h[o] = f2bf(val);
            }
          }
        }
      }
    }
  }
}

size_t conv_bf16_lds_bytes(const ConvArgs &p) {
  const int halo = p.ksize == 3 ? 1 : 0;
  const size_t HV = (size_t)(2 * p.tcx + 2 * halo) * (2 * p.tcy + 2 * halo) * (2 * p.tcz + 2 * halo);
  const int Q = (p.ksize == 3 ? 27 : 1) * p.cc4;
  return ((HV * p.ccs + 7) & ~(size_t)7) * sizeof(unsigned short) + (size_t)((Q + 3) & ~3) * sizeof(int) +
         (size_t)2 * p.cc4 * 8 * sizeof(float) + HV * sizeof(int);
}

template <int WM, int WN, int TM, int TN> static void launch_bf16(const ConvArgs &p, int B, hipStream_t s) {
  const int ngroups = (p.coutp / 32 + WN * TN - 1) / (WN * TN);
  dim3 grid(B * p.ntx * p.nty * p.ntz, ngroups), block(64 * WM * WN);
  ensure_max_lds(reinterpret_cast<const void *>(conv3d_bf16_kernel<WM, WN, TM, TN>), 160 * 1024);
  hipLaunchKernelGGL((conv3d_bf16_kernel<WM, WN, TM, TN>), grid, block, conv_bf16_lds_bytes(p), s, p);
}

void launch_conv_bf16(const ConvArgs &p, int cfg, int B, hipStream_t s) {
  switch (cfg) {
    case CONV_CFG_4x1_2x1: launch_bf16<4, 1, 2, 1>(p, B, s); break;
    case CONV_CFG_2x2_3x1: launch_bf16<2, 2, 3, 1>(p, B, s); break;
    case CONV_CFG_1x4_7x1: launch_bf16<1, 4, 7, 1>(p, B, s); break;
    case CONV_CFG_4x1_2x3: launch_bf16<4, 1, 2, 3>(p, B, s); break;
    case CONV_CFG_4x1_1x5: launch_bf16<4, 1, 1, 5>(p, B, s); break;
    default: break;
  }
}

// global max over space of a bf16 tensor: in [B][S]^3[in_cs] -> out fp32 [B][out_cs]
__global__ void gmax_bf16_kernel(const unsigned short *in, float *out, int C, int in_cs, int out_cs, int S3) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const unsigned short *src = in + (size_t)b * S3 * in_cs + c;
    float m = bf2f(src[0]);
    for (int v = 1; v < S3; v++) m = fmaxf(m, bf2f(src[(size_t)v * in_cs]));
    out[(size_t)b * out_cs + c] = m;
  }
}

void launch_gmax_bf16(const void *in, float *out, int B, int C, int in_cs, int out_cs, int S, hipStream_t s) {
  hipLaunchKernelGGL(gmax_bf16_kernel, dim3(B), dim3(256), 0, s, reinterpret_cast<const unsigned short *>(in), out, C,
                     in_cs, out_cs, S * S * S);
}

// max_pool3d(kernel = whole grid) backward on a bf16 activation: fp32 gradient to the first arg-max voxel
__global__ void gmax_backward_bf16_kernel(const unsigned short *act, const float *g_out, float *g_in, int C, int in_cs,
                                          int out_cs, int S3) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const unsigned short *src = act + (size_t)b * S3 * in_cs + c;
    float m = bf2f(src[0]);
    int am = 0;
    for (int v = 1; v < S3; v++) {
      const float t = bf2f(src[(size_t)v * in_cs]);
      if (t > m) m = t, am = v;
    }
    float *dst = g_in + (size_t)b * S3 * in_cs + c;
    const float g = g_out[(size_t)b * out_cs + c];
    for (int v = 0; v < S3; v++) dst[(size_t)v * in_cs] = v == am ? g : 0.f;
  }
}

void launch_gmax_backward_bf16(const void *act, const float *g_out, float *g_in, int B, int C, int in_cs, int out_cs,
                               int S, hipStream_t s) {
  hipLaunchKernelGGL(gmax_backward_bf16_kernel, dim3(B), dim3(256), 0, s,
                     reinterpret_cast<const unsigned short *>(act), g_out, g_in, C, in_cs, out_cs, S * S * S);
}

}  // namespace mig
