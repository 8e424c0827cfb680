// pk_f32_next_to_mfma.hip -- do the packed-fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32) of one kernel give wrong
// results while ANOTHER kernel's MFMA waves share the SIMD?
//
// Round 6: voxelize_tiles next to a Dense scorer's conv
// kernels on a second hardware queue produced, in ~2 % of its launches, squared distances that were wrong in the UPPER HALF
// of a wavefront (lanes 32-63); the build of the same kernel without v_pk_*_f32 was clean in 20,000 launches.  This is the
// smallest program that asks the hardware the same question:
//   victim   : single-wave workgroups; per iteration each lane forms d = g - a (g per lane, a wave-uniform) and d * d once with
//              the packed instruction (the exact encodings hipcc emitted in voxelize_tiles: SGPR-pair source, op_sel_hi,
//              neg_lo / neg_hi) and once with v_sub_f32 / v_mul_f32; bit differences are counted per half-wave
//   aggressor: 256-thread workgroups on a second stream spinning on one MFMA shape (f16 16x16x32, f16 32x32x16, fp32
//              32x32x2), on plain VALU work, or idle
// hipcc --offload-arch=gfx950 -O3 -o pk_f32_next_to_mfma pk_f32_next_to_mfma.hip ; ./pk_f32_next_to_mfma
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) {                                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                    \
      exit(1);                                                                                     \
    }                                                                                              \
  } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MODE 0: v_pk_add_f32 d, g, s[a:a+1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]   (g - a, a from an SGPR pair's low word)
// MODE 1: the same subtraction with a in VGPRs (v_pk_add_f32 d, g, va neg_lo:[0,1] neg_hi:[0,1])
// MODE 2: v_pk_mul_f32 q, d, d
// MODE 3: v_pk_add_f32 r, x, y op_sel_hi:[1,0]   (y's low word added to both halves of x: the xy0 + dz0 step)
// MODE 4: the voxelizer's whole chain as hipcc compiles it from vector-typed source (no asm)
// MODE 5: MODE 0's subtraction with the wave-uniform operand coming from an s_load_dwordx8 (constant address space) while the
//         NEXT record's s_load_dwordx8 is in flight -- voxelize_tiles' hit loop; MODE 6: the same, the next load waited for
//         before the packed instructions run
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) f32x8 *ConstRec;

template <int MODE>
__global__ __launch_bounds__(64) void victim_sload(const float *table, int iters, unsigned long long *bad) {
  const int lane = threadIdx.x;
  const ConstRec recs = (ConstRec)(const void *)table;  // 512 records of 8 floats
  unsigned errs = 0;
  float g0 = table[(blockIdx.x * 64 + lane) & 4095], g1 = table[(blockIdx.x * 64 + lane + 1777) & 4095];
  unsigned idx = (blockIdx.x * 7919u) & 511u;
  f32x8 rec_n = recs[idx];
  for (int it = 0; it < iters; it++) {
    const f32x8 rec = rec_n;
    idx = (idx * 1664525u + 1013904223u) & 511u;
    rec_n = recs[idx];  // s_load_dwordx8, in flight from here on
    if (MODE == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(rec_n));
    const float ax = rec[0], ay = rec[1], az = rec[2];
    f32x2 axy = {ax, ay}, azz = {az, rec[3]}, gx = {g0, g1}, gy = {g1, g0}, gz = {g0 + 0.5f, g1 - 0.5f}, dx, dy, dz;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx) : "v"(gx), "s"(axy));
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy) : "v"(gy), "s"(axy));
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dz) : "v"(gz), "s"(azz));
    float r[6];
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[0]) : "v"(gx[0]), "s"(ax));
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[1]) : "v"(gx[1]), "s"(ax));
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[2]) : "v"(gy[0]), "s"(ay));
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[3]) : "v"(gy[1]), "s"(ay));
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[4]) : "v"(gz[0]), "s"(az));
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[5]) : "v"(gz[1]), "s"(az));
    errs += (__float_as_uint(dx[0]) != __float_as_uint(r[0])) + (__float_as_uint(dx[1]) != __float_as_uint(r[1])) +
            (__float_as_uint(dy[0]) != __float_as_uint(r[2])) + (__float_as_uint(dy[1]) != __float_as_uint(r[3])) +
            (__float_as_uint(dz[0]) != __float_as_uint(r[4])) + (__float_as_uint(dz[1]) != __float_as_uint(r[5]));
    g0 = r[0] * 0.5f + 1.0f, g1 = r[3] * 0.25f - 1.0f;
    if (!(fabsf(g0) < 1e3f)) g0 = 1.f;
    if (!(fabsf(g1) < 1e3f)) g1 = 2.f;
  }
  unsigned e = errs;
  for (int off = 16; off > 0; off >>= 1) e += __shfl_xor(e, off);
  if ((lane & 31) == 0 && e) atomicAdd(bad + (lane >> 5), (unsigned long long)e);
}

// MODE 7 / 8 / 9: the voxelizer's chain as ONE asm statement, instruction for instruction what hipcc emitted in voxelize_tiles
// (7: its own s_nop 0 between directly dependent packed instructions; 8: s_nop 3 -- four wait states -- behind EVERY
// instruction; 9: s_nop 7), against the same arithmetic in scalar instructions
#define PK_CHAIN(N1, NA)                                                                      \
  "v_pk_add_f32 %4, %8, %11 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" NA             \
  "v_pk_add_f32 %5, %9, %11 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t" NA                \
  "v_pk_add_f32 %6, %10, %12 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" NA            \
  "v_pk_mul_f32 %4, %4, %4\n\t" NA                                                        \
  "v_pk_mul_f32 %5, %5, %5\n\t" NA                                                        \
  "v_pk_mul_f32 %6, %6, %6\n\t" NA                                                        \
  "v_pk_add_f32 %7, %4, %5 op_sel_hi:[1,0]\n\t" N1 NA                                     \
  "v_pk_add_f32 %0, %7, %6 op_sel_hi:[1,0]\n\t" NA                                        \
  "v_pk_add_f32 %1, %7, %6 op_sel:[0,1]\n\t" NA                                           \
  "v_pk_add_f32 %7, %4, %5 op_sel:[0,1]\n\t" N1 NA                                        \
  "v_pk_add_f32 %2, %7, %6 op_sel_hi:[1,0]\n\t" NA                                        \
  "v_pk_add_f32 %3, %7, %6 op_sel:[0,1]\n\t" NA

template <int MODE>
__global__ __launch_bounds__(64) void victim_chain(const float *table, int iters, unsigned long long *bad) {
  const int lane = threadIdx.x;
  unsigned errs = 0;
  float g0 = table[(blockIdx.x * 64 + lane) & 4095], g1 = table[(blockIdx.x * 64 + lane + 1777) & 4095];
  for (int it = 0; it < iters; it++) {
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, table[(blockIdx.x * 131 + it * 7) & 4095])));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, table[(blockIdx.x * 17 + it * 3 + 5) & 4095])));
    const float c = a * 0.5f + b;
    const f32x2 gx = {g0, g1}, gy = {g1 * 0.5f, g0 + 0.25f}, gz = {g0 - 0.75f, g1 + 0.5f}, sab = {a, b}, scd = {c, a};
    f32x2 r00, r01, r10, r11, t0, t1, t2, t3;
    if (MODE == 7)
      asm volatile(PK_CHAIN("s_nop 0\n\t", "") : "=&v"(r00), "=&v"(r01), "=&v"(r10), "=&v"(r11), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(gx), "v"(gy), "v"(gz), "s"(sab), "s"(scd));
    else if (MODE == 8)
      asm volatile(PK_CHAIN("", "s_nop 3\n\t") : "=&v"(r00), "=&v"(r01), "=&v"(r10), "=&v"(r11), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(gx), "v"(gy), "v"(gz), "s"(sab), "s"(scd));
    else
      asm volatile(PK_CHAIN("", "s_nop 7\n\t") : "=&v"(r00), "=&v"(r01), "=&v"(r10), "=&v"(r11), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(gx), "v"(gy), "v"(gz), "s"(sab), "s"(scd));
    float ref[8];
    {
      float ex[2], ey[2], ez[2];
      const float gxs[2] = {gx[0], gx[1]}, gys[2] = {gy[0], gy[1]}, gzs[2] = {gz[0], gz[1]};
      for (int d = 0; d < 2; d++) {
        float t;
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t) : "v"(gxs[d]), "s"(a));
        asm volatile("v_mul_f32 %0, %1, %1" : "=v"(ex[d]) : "v"(t));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t) : "v"(gys[d]), "s"(b));
        asm volatile("v_mul_f32 %0, %1, %1" : "=v"(ey[d]) : "v"(t));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t) : "v"(gzs[d]), "s"(c));
        asm volatile("v_mul_f32 %0, %1, %1" : "=v"(ez[d]) : "v"(t));
      }
      for (int dy = 0; dy < 2; dy++)
        for (int dz = 0; dz < 2; dz++)
          for (int dx = 0; dx < 2; dx++) {
            float t;
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(ex[dx]), "v"(ey[dy]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ref[(dy * 2 + dz) * 2 + dx]) : "v"(t), "v"(ez[dz]));
          }
    }
    const f32x2 rr[4] = {r00, r01, r10, r11};
    for (int k = 0; k < 4; k++)
      for (int dx = 0; dx < 2; dx++) errs += __float_as_uint(rr[k][dx]) != __float_as_uint(ref[k * 2 + dx]);
    g0 = g0 * 0.999f + a * 1e-3f, g1 = g1 * 0.998f - b * 1e-3f;
  }
  unsigned e = errs;
  for (int off = 16; off > 0; off >>= 1) e += __shfl_xor(e, off);
  if ((lane & 31) == 0 && e) atomicAdd(bad + (lane >> 5), (unsigned long long)e);
}

template <int MODE>
__global__ __launch_bounds__(64) void victim(const float *table, int iters, unsigned long long *bad /* [2]: lanes 0-31, 32-63 */) {
  const int lane = threadIdx.x;
  unsigned errs = 0;
  float g0 = table[(blockIdx.x * 64 + lane) & 4095], g1 = table[(blockIdx.x * 64 + lane + 1777) & 4095];
  for (int it = 0; it < iters; it++) {
    // wave-uniform operand, a new value every iteration (an s_load result in the voxelizer)
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, table[(blockIdx.x * 131 + it * 7) & 4095])));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, table[(blockIdx.x * 17 + it * 3 + 5) & 4095])));
    f32x2 g = {g0, g1};
    if (MODE == 0) {
      f32x2 as = {a, b}, d;
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(g), "s"(as));
      float r0, r1;
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(g0), "s"(a));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(g1), "s"(a));
      errs += (__float_as_uint(d[0]) != __float_as_uint(r0)) + (__float_as_uint(d[1]) != __float_as_uint(r1));
      g0 = r0 * 0.5f + 1.0f, g1 = r1 * 0.25f - 1.0f;
    } else if (MODE == 1) {
      f32x2 av = {a, a}, d;
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(g), "v"(av));
      float r0, r1;
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(g0), "s"(a));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(g1), "s"(a));
      errs += (__float_as_uint(d[0]) != __float_as_uint(r0)) + (__float_as_uint(d[1]) != __float_as_uint(r1));
      g0 = r0 * 0.5f + 1.0f, g1 = r1 * 0.25f - 1.0f;
    } else if (MODE == 2) {
      f32x2 q;
      asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(q) : "v"(g));
      float r0, r1;
      asm volatile("v_mul_f32 %0, %1, %1" : "=v"(r0) : "v"(g0));
      asm volatile("v_mul_f32 %0, %1, %1" : "=v"(r1) : "v"(g1));
      errs += (__float_as_uint(q[0]) != __float_as_uint(r0)) + (__float_as_uint(q[1]) != __float_as_uint(r1));
      g0 = r0 * 0.5f + a, g1 = r1 * 0.25f - b;
      if (!(fabsf(g0) < 1e3f)) g0 = a;
      if (!(fabsf(g1) < 1e3f)) g1 = b;
    } else if (MODE == 3) {
      f32x2 y = {g1 + a, g0 - b}, r;
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(g), "v"(y));
      float r0, r1;
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(r0) : "v"(g0), "v"(y[0]));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(r1) : "v"(g1), "v"(y[0]));
      errs += (__float_as_uint(r[0]) != __float_as_uint(r0)) + (__float_as_uint(r[1]) != __float_as_uint(r1));
      g0 = r0 * 0.5f + 1.0f, g1 = r1 * 0.25f - 1.0f;
    } else {
      // voxelize_tiles' distance arithmetic, vector-typed (hipcc: 3 v_pk_add with SGPR sources, 3 v_pk_mul, 6 v_pk_add)
      const float c = a * 0.5f + b;
      const f32x2 gx2 = {g0, g1}, gy2 = {g1 * 0.5f, g0 + 0.25f}, gz2 = {g0 - 0.75f, g1 + 0.5f};
      const f32x2 ax2 = {a, a}, ay2 = {b, b}, az2 = {c, c};
      f32x2 dxx = gx2 - ax2, dyy = gy2 - ay2, dzz = gz2 - az2;
      dxx = dxx * dxx, dyy = dyy * dyy, dzz = dzz * dzz;
      const f32x2 dy0 = {dyy[0], dyy[0]}, dy1 = {dyy[1], dyy[1]}, dz0 = {dzz[0], dzz[0]}, dz1 = {dzz[1], dzz[1]};
      const f32x2 xy0 = dxx + dy0, xy1 = dxx + dy1;
      const f32x2 r00 = xy0 + dz0, r01 = xy0 + dz1, r10 = xy1 + dz0, r11 = xy1 + dz1;
      // the same through opaque scalar instructions
      float ref[8];
      {
        float ex[2], ey[2], ez[2];
        const float gxs[2] = {g0, g1}, gys[2] = {g1 * 0.5f, g0 + 0.25f}, gzs[2] = {g0 - 0.75f, g1 + 0.5f};
        for (int d = 0; d < 2; d++) {
          float t;
          asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t) : "v"(gxs[d]), "s"(a));
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(ex[d]) : "v"(t));
          asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t) : "v"(gys[d]), "s"(b));
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(ey[d]) : "v"(t));
          asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t) : "v"(gzs[d]), "s"(c));
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(ez[d]) : "v"(t));
        }
        for (int dy = 0; dy < 2; dy++)
          for (int dz = 0; dz < 2; dz++)
            for (int dx = 0; dx < 2; dx++) {
              float t;
              asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(ex[dx]), "v"(ey[dy]));
              asm volatile("v_add_f32 %0, %1, %2" : "=v"(ref[(dy * 2 + dz) * 2 + dx]) : "v"(t), "v"(ez[dz]));
            }
      }
      const f32x2 rr[4] = {r00, r01, r10, r11};
      for (int k = 0; k < 4; k++)
        for (int dx = 0; dx < 2; dx++) errs += __float_as_uint(rr[k][dx]) != __float_as_uint(ref[k * 2 + dx]);
      g0 = g0 * 0.999f + a * 1e-3f, g1 = g1 * 0.998f - b * 1e-3f;
    }
  }
  const unsigned lo = __builtin_amdgcn_readfirstlane(0);
  (void)lo;
  // per half-wave totals
  unsigned e = errs;
  for (int off = 16; off > 0; off >>= 1) e += __shfl_xor(e, off);
  if ((lane & 31) == 0 && e) atomicAdd(bad + (lane >> 5), (unsigned long long)e);
}

// KIND 0: v_mfma_f32_16x16x32_f16, 1: v_mfma_f32_32x32x16_f16, 2: v_mfma_f32_32x32x2_f32, 3: VALU only, 4: s_sleep only,
// 5: v_mfma_f32_16x16x32_f16 with both operands of every instruction read from LDS by ds_read_b128 (a conv K loop), 6: the
// ds_read_b128 traffic alone, 7: conv3d_h2_k1s_kernel's K loop -- v_mfma_f32_32x32x16_f16 whose B operands arrive by
// global_load_dwordx4 one step ahead (the one aggressor kernel whose K loop has to run for the fault to appear), 8: those
// global loads without the MFMAs
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(int rounds, float *sink, const uint4 *wts = nullptr) {
  const int lane = threadIdx.x & 63;
  __shared__ __attribute__((aligned(16))) uint4 s_ops[2048];  // 32 KB
  if (KIND == 5 || KIND == 6) {
    for (int i = threadIdx.x; i < 2048; i += 256) s_ops[i] = make_uint4(0x3c003c00u + i, 0x38003800u, 0x34003400u + i, 0x30003000u);
    __syncthreads();
  }
  f16x8 a, b;
  for (int i = 0; i < 8; i++) a[i] = (_Float16)(0.001f * (lane + i)), b[i] = (_Float16)(0.002f * (lane - i));
  f32x4 c4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x16 c16[2];
  for (int i = 0; i < 16; i++) c16[0][i] = 0.f, c16[1][i] = 0.f;
  float v = (float)lane;
  for (int r = 0; r < rounds; r++) {
    if (KIND == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) c4[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4[k & 3], 0, 0, 0);
    } else if (KIND == 1) {
#pragma unroll
      for (int k = 0; k < 4; k++) c16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c16[k & 1], 0, 0, 0);
    } else if (KIND == 2) {
#pragma unroll
      for (int k = 0; k < 4; k++) c16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, 1.0f, c16[k & 1], 0, 0, 0);
    } else if (KIND == 3) {
#pragma unroll
      for (int k = 0; k < 32; k++) v = v * 1.0001f + 0.5f;
    } else if (KIND == 7 || KIND == 8) {
      // weights: 36 KB region walked by every wave (L1 / L2 hits), three 16-byte B operands per step like TN = 3
      uint4 w0 = wts[(threadIdx.x + 192 * r) & 2047], w1 = wts[(threadIdx.x + 192 * r + 64) & 2047], w2 = wts[(threadIdx.x + 192 * r + 128) & 2047];
      if (KIND == 7) {
        c16[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, w0), c16[0], 0, 0, 0);
        c16[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, w1), c16[1], 0, 0, 0);
        c16[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, __builtin_bit_cast(f16x8, w2), c16[0], 0, 0, 0);
      } else {
        v += __uint_as_float((w0.x ^ w1.y ^ w2.z) & 0x007fffffu) * 1e-30f;
      }
    } else if (KIND == 5 || KIND == 6) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint4 ua = s_ops[(threadIdx.x + 67 * k + 13 * r) & 2047], ub = s_ops[(threadIdx.x * 3 + 129 * k + 7 * r) & 2047];
        if (KIND == 5) c4[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ua), __builtin_bit_cast(f16x8, ub), c4[k & 3], 0, 0, 0);
        else v += __uint_as_float(ua.x ^ ub.w) * 1e-30f;
      }
    } else {
      __builtin_amdgcn_s_sleep(32);
    }
  }
  float s = v;
  for (int i = 0; i < 4; i++) s += c4[i][0] + c4[i][3];
  s += c16[0][0] + c16[1][5];
  if (s == 1234.5f) sink[0] = s;
}

template <int MODE>
static void run_mode(const char *what, const float *d_table, unsigned long long *d_bad, hipStream_t sv, hipStream_t sa, float *d_sink) {
  const char *kinds[] = {"mfma f16 16x16x32", "mfma f16 32x32x16", "mfma f32 32x32x2", "valu only", "idle (s_sleep)", "no second queue", "mfma f16 + ds_read_b128", "ds_read_b128 only",
                         "mfma f16 32x32x16 + global_load_dwordx4", "global_load_dwordx4 only"};
  static uint4 *d_wts = nullptr;
  if (!d_wts) {
    CHECK(hipMalloc(&d_wts, 2048 * 16));
    CHECK(hipMemset(d_wts, 0x3c, 2048 * 16));
  }
  for (int kind = 0; kind < 10; kind++) {
    CHECK(hipMemset(d_bad, 0, 16));
    const int agg_wgs = 256 * 2, rounds = 400000;  // ~tens of ms: covers the victim launches
    if (kind == 0) hipLaunchKernelGGL(aggressor<0>, dim3(agg_wgs), dim3(256), 0, sa, rounds, d_sink);
    if (kind == 1) hipLaunchKernelGGL(aggressor<1>, dim3(agg_wgs), dim3(256), 0, sa, rounds / 2, d_sink);
    if (kind == 2) hipLaunchKernelGGL(aggressor<2>, dim3(agg_wgs), dim3(256), 0, sa, rounds / 2, d_sink);
    if (kind == 3) hipLaunchKernelGGL(aggressor<3>, dim3(agg_wgs), dim3(256), 0, sa, rounds / 2, d_sink);
    if (kind == 4) hipLaunchKernelGGL(aggressor<4>, dim3(agg_wgs), dim3(256), 0, sa, rounds / 20, d_sink);
    if (kind == 6) hipLaunchKernelGGL(aggressor<5>, dim3(agg_wgs), dim3(256), 0, sa, rounds / 2, d_sink);
    if (kind == 7) hipLaunchKernelGGL(aggressor<6>, dim3(agg_wgs), dim3(256), 0, sa, rounds / 2, d_sink);
    if (kind == 8) hipLaunchKernelGGL(aggressor<7>, dim3(agg_wgs), dim3(256), 0, sa, rounds, d_sink, d_wts);
    if (kind == 9) hipLaunchKernelGGL(aggressor<8>, dim3(agg_wgs), dim3(256), 0, sa, rounds, d_sink, d_wts);
    const int launches = 40, wgs = 1024, iters = 2000;
    for (int l = 0; l < launches; l++) {
      if constexpr (MODE >= 5) hipLaunchKernelGGL(victim_sload<MODE>, dim3(wgs), dim3(64), 0, sv, d_table, iters, d_bad);
      else hipLaunchKernelGGL(victim<MODE>, dim3(wgs), dim3(64), 0, sv, d_table, iters, d_bad);
    }
    CHECK(hipStreamSynchronize(sv));
    const bool agg_running = kind != 5 && hipStreamQuery(sa) == hipErrorNotReady;
    CHECK(hipStreamSynchronize(sa));
    unsigned long long bad[2];
    CHECK(hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost));
    const double checked = (double)launches * wgs * 64 * iters * (MODE == 4 ? 8 : MODE >= 5 ? 6 : 2);
    printf("%-44s next to %-20s: wrong results lanes 0-31: %llu, lanes 32-63: %llu  of %.3g%s\n", what, kinds[kind], bad[0], bad[1], checked,
           kind != 5 && !agg_running ? "   (the aggressor had finished before the victim did)" : "");
    fflush(stdout);
  }
}

// The victims as a library (hipcc -shared -fPIC -DPK_VICTIM_LIB): tools/experiments/pk_victim_next_to_scorer.py runs them next to a
// REAL aggressor, a Dense scorer of libmi_gnina.so on a second host thread.  bad2[0..1] = wrong results in lanes 0-31 / 32-63.
extern "C" int pk_victim_run(int mode, int launches, int wgs, int iters, unsigned long long *bad2) {
  static float *d_table = nullptr;
  static unsigned long long *d_bad = nullptr;
  static hipStream_t sv = nullptr;
  if (!d_table) {
    float *h = (float *)malloc(4096 * 4);
    unsigned s = 12345u;
    for (int i = 0; i < 4096; i++) {
      s = s * 1664525u + 1013904223u;
      h[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 24.0f;
    }
    CHECK(hipMalloc(&d_table, 4096 * 4));
    CHECK(hipMalloc(&d_bad, 16));
    CHECK(hipMemcpy(d_table, h, 4096 * 4, hipMemcpyHostToDevice));
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    free(h);
  }
  CHECK(hipMemsetAsync(d_bad, 0, 16, sv));
  for (int l = 0; l < launches; l++) {
    switch (mode) {
      case 0: hipLaunchKernelGGL(victim<0>, dim3(wgs), dim3(64), 0, sv, d_table, iters, d_bad); break;
      case 1: hipLaunchKernelGGL(victim<1>, dim3(wgs), dim3(64), 0, sv, d_table, iters, d_bad); break;
      case 2: hipLaunchKernelGGL(victim<2>, dim3(wgs), dim3(64), 0, sv, d_table, iters, d_bad); break;
      case 3: hipLaunchKernelGGL(victim<3>, dim3(wgs), dim3(64), 0, sv, d_table, iters, d_bad); break;
      case 4: hipLaunchKernelGGL(victim<4>, dim3(wgs), dim3(64), 0, sv, d_table, iters, d_bad); break;
      case 5: hipLaunchKernelGGL(victim_sload<5>, dim3(wgs), dim3(64), 0, sv, d_table, iters, d_bad); break;
      case 6: hipLaunchKernelGGL(victim_sload<6>, dim3(wgs), dim3(64), 0, sv, d_table, iters, d_bad); break;
      case 7: hipLaunchKernelGGL(victim_chain<7>, dim3(wgs), dim3(64), 0, sv, d_table, iters, d_bad); break;
      case 8: hipLaunchKernelGGL(victim_chain<8>, dim3(wgs), dim3(64), 0, sv, d_table, iters, d_bad); break;
      default: hipLaunchKernelGGL(victim_chain<9>, dim3(wgs), dim3(64), 0, sv, d_table, iters, d_bad); break;
    }
  }
  CHECK(hipMemcpyAsync(bad2, d_bad, 16, hipMemcpyDeviceToHost, sv));
  CHECK(hipStreamSynchronize(sv));
  return 0;
}

#ifndef PK_VICTIM_LIB
int main() {
  CHECK(hipSetDevice(0));
  float *h = (float *)malloc(4096 * 4);
  unsigned s = 12345u;
  for (int i = 0; i < 4096; i++) {
    s = s * 1664525u + 1013904223u;
    h[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 24.0f;
  }
  float *d_table, *d_sink;
  unsigned long long *d_bad;
  CHECK(hipMalloc(&d_table, 4096 * 4));
  CHECK(hipMalloc(&d_sink, 64));
  CHECK(hipMalloc(&d_bad, 16));
  CHECK(hipMemcpy(d_table, h, 4096 * 4, hipMemcpyHostToDevice));
  hipStream_t sv, sa;
  CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  run_mode<4>("the voxelizer's chain, compiler-generated", d_table, d_bad, sv, sa, d_sink);
  if (getenv("PK_SHORT")) return 0;
  run_mode<5>("v_pk_add_f32 g - s[a], next s_load in flight", d_table, d_bad, sv, sa, d_sink);
  run_mode<6>("v_pk_add_f32 g - s[a], next s_load waited for", d_table, d_bad, sv, sa, d_sink);
  run_mode<0>("v_pk_add_f32 g - s[a] (op_sel_hi, neg)", d_table, d_bad, sv, sa, d_sink);
  run_mode<1>("v_pk_add_f32 g - v[a] (neg)", d_table, d_bad, sv, sa, d_sink);
  run_mode<2>("v_pk_mul_f32 d * d", d_table, d_bad, sv, sa, d_sink);
  run_mode<3>("v_pk_add_f32 x + y.lo (op_sel_hi)", d_table, d_bad, sv, sa, d_sink);
  return 0;
}
#endif
