// conv3d.h -- argument block + launchers of the CNN kernels (conv3d.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace mig {

struct ConvArgs {
  const float *in;   // [B][S][S][S][in_cs] channels last
  int in_cs;
  float *out;        // [B][So][So][So][out_cs], So = S (pool 0) or S/2
  int out_cs, out_c0;
  const float *wp;   // packed weights [chunk][pair][2][coutp][4]
  const float *bias; // [coutp]
  const float *bn_scale, *bn_shift;  // [cin4*4] or nullptr
  int S;
  int cout, coutp;   // real / padded-to-32 output channels
  int ksize;         // 1 or 3
  int relu, pool;    // pool: 0 none, 1 max, 2 avg (2x2x2 after ReLU)
  int tcx, tcy, tcz; // workgroup tile in 2x2x2 cells
  int ntx, nty, ntz; // tiles per axis
  // backward-data support (the same kernel runs the transposed convolution):
  //   in_mode 0: plain input
  //   in_mode 1: input masked by (in_act > 0) at the same voxel      (ReLU backward)
  //   in_mode 2: max-unpool on load: `in`, `in_argmax`, `in_act` are at S/2; a voxel receives the pooled
  //              gradient iff it was the arg-max of its 2x2x2 cell and the pooled activation was > 0
  //   in_mode 3: average-unpool on load (generic fp32 kernel only): `in` is at S/2, `in_act` at S; a voxel receives
  //              1/8 of its cell's gradient iff its own activation was > 0
  int in_mode;
  const unsigned char *in_argmax;  // [B][S/2]^3[in_cs] (in_mode 2)
  const float *in_act;             // mask source; stride in_act_cs
  int in_act_cs;
  unsigned char *argmax_out;       // pool == 1: arg-max index (x*4+y*2+z) per pooled output, or nullptr
  // Dense-block backward (pool == 0 only): out[ch] (+)= acc * out_scale[ch].  out_scale is the eval-BatchNorm
  // scale of the forward layer's input (d(BN x)/dx); accumulate adds into the concat buffer's gradient.
  const float *out_scale;          // [coutp] or nullptr
  int accumulate;
  // fused 1x1x1 conv after this one (Default2018: conv3 -> ReLU -> conv1 -> ReLU -> pool): the ReLU'd tile goes
  // through LDS into a second MFMA pass (cout -> cout channels), then the usual epilogue.  fp32 kernel only.
  const float *post_w;             // packed [pair][2][coutp][4] weights of the 1x1 conv, or nullptr
  const float *post_bias;          // [coutp]
  int post_relu;
  int post_rows;                   // WM * TM * 32: voxel rows of a workgroup (LDS sizing)
  int post_cc4, post_wrows;        // K-chunk geometry of post_w (the 1x1 conv's own plan: cc4, wrows)
  // bf16 kernels (conv3d_bf16.hip) only: element type of the tensors behind in / out / in_act
  // (1 = fp32, 0 = bf16; cc4 / ccs / cin4 then count OCTETS of 8 channels and wp is a packed bf16 array)
  int in_f32, out_f32, act_f32;
  int sparse;        // 1: skip channel quads that are all-zero inside a tile (first conv: pooled voxel grid; un-pooled gradients) and,
                     // per MFMA, all-zero A operands; 2: the per-MFMA test only, every quad listed (ReLU'd activations).
                     // (0 with korder = 1 runs the K order of 2 without the test: same bits, for inputs that are not sparse enough)
  int cc4;           // channel quads per K chunk
  int ccs;           // LDS floats per halo voxel (>= 4*cc4, padded against bank conflicts)
  int nchunks;
  int cin4;          // input-channel quads actually present (the last chunk may hold fewer than cc4)
  int wrows;         // fp32 32x32x2 kernel: quad rows per chunk of the packed weights, 2 * (Q / 2 + 1) >= Q + 1; rows >= Q are zero
  // korder 1 = the K loop walks the quads channel-major (the order of the listed / sparse K loop) instead of tap-major.
  // N = 16 kernel (Dense-block convs) with the per-MFMA zero test: korder 1 = the K loop walks the quads channel-major
  // (all 27 taps of a channel quad in conv_snake_tap order, then the next quad; weights packed to match) so that the four
  // k of an instruction are four neighbouring taps of ONE channel.  The block's eval-BatchNorm is then applied as
  // x * scale only (bn_shift = zeros: a zero activation stays an exact zero in LDS) and its shift enters through
  // bias_tab [27][16]: bias[co] + sum over the taps INSIDE the grid and all input channels of shift[c] * W[tap][c][co],
  // one row per border class of the output voxel (3 x 3 x 3: first / interior / last plane per axis).
  int korder;
  const float *bias_tab;
  // profiling only (mi_scorer_enable_profile): [kMfmaCountSlots] counters the zero-skipping kernels add the number of MFMA
  // instructions they EXECUTED to (slot = workgroup % kMfmaCountSlots); nullptr otherwise
  unsigned long long *mfma_count;
  // M-tile geometry: 0 = the four cells of an M-tile are consecutive cells of the tile in (x, y, z) raster order;
  // 1 = they are stacked along x (tcx % 4 == 0); 2 (conv3d_h2_kernel only) = a 2 x 2 square in (x, y), tcx == tcy == 2.  With a 4 x 4 x 2-cell tile (halo 10 x 10 x 6, odd quad stride) the
  // sixteen lanes of every ds_read_b128 lane group then hit sixteen different 16-byte LDS slots (no bank conflict;
  // the raster order costs 3 LDS cycles per group) -- see LAB.md §3.1.
  int mt_x;
  // split-fp16 kernels (conv3d_h2.hip) only: cc4 counts OCTETS per K chunk, ccs fp16 elements per halo voxel in LDS, cin4
  // stays the input's channel QUADS, wp is the packed [chunk][pair][2][coutp][h0..h7 | l0..l7] fp16 array of the weights
  // times 1 / h2_unscale (a power of two); sparse != 0 = skip the MFMAs of an all-zero A operand
  float h2_unscale;
  float h2_post_unscale;  // fused 1x1x1 conv on the split-fp16 kernel: post_w = its packed [pair][2][coutp][h | l] weights, post_cc4 its OCTETS
  int h2_pad_y, h2_pad_x;  // split-fp16 kernels: 16-byte pad slots behind every z-row / x-plane of the LDS halo tile
  // 32-wide 3x3x3 split-fp16 kernel (conv3d_h2_kernel): activation tensors may live in HBM already split by their producer
  // ("split format": [pose][octet of 8 channels][x][y][z][h0..h7 | l0..l7] fp16, h = RN_f16(a), l = RN_f16(a - h) -- the bytes
  // of an fp32 tensor of cs = 8 * octets channels, octet-major so that the halo tile of a K chunk, one octet, is a box of a
  // dense array).  in_split: `in` is such a tensor (in_cs = 8 * octets) and the halo tile is staged by LDS-DMA, no
  // registers, no arithmetic; out_split: the epilogue writes one.
  int in_split, out_split;
  // occupancy of a split-format voxel grid, written by the voxelizer (voxelize.h VoxArgs::occ): [pose][occ_nt]^3[8] bytes,
  // byte o of a 4 x 4 x 4-cell block = some cell of the block is non-zero in octet o.  The zero-skipping first convolution
  // reads the blocks its halo tile touches and stages the octets that have something in them only.  nullptr = not known.
  const unsigned char *in_occ;
  int occ_nt;
  int h2_wlds;  // conv3d_h2_kernel variant: the chunk's weights go through LDS, shared by the workgroup's waves; one halo-tile
                // buffer.  2 = and two consecutive poses per workgroup on one copy of the weights (split-format input)
  int nposes;   // poses of the launch (set by the launcher)
  int h2_prefetch;  // weights-in-LDS variant: touch the next (chunk, pose) item's tile towards L2 under this item's K loop
  // sticky flag (one word per scorer): an activation left the fp16 range (|a| > 65504, or NaN) where a split-fp16 kernel
  // produced or consumed it -- the call's scores are then recomputed on the fp32-MFMA kernels (engine.cpp)
  unsigned *h2_overflow;
  // Transposed convs of the gradient pass on the split-fp16 kernel (conv3d_h2_kernel, fp32 tensors in and out).  A gradient
  // has no fixed range: every producer of one records the per-pose maximum of |g| (out_amax: atomicMax of the float's bits),
  // and the consumer stages g * 2^(14 - exponent(amax)) -- exact, largest staged value in [2^14, 2^15), absolute error of a
  // split value <= 2^-25 = 2^-39 of the pose's largest -- and un-scales its accumulators by the inverse.  The ReLU mask of
  // the layer whose output gradient this is, is applied by the PRODUCER of that gradient (out_mask: the forward activation
  // at the same voxel and channel, idempotent) so that the staging has nothing to look up but, behind a fused max pool, the
  // arg-max (in_mode 2 on this kernel: `in`, `in_argmax` at S / 2, already masked).
  const unsigned *in_amax;  // [pose] or nullptr
  unsigned *out_amax;       // [pose] or nullptr (pool == 0 epilogues: conv3d_mfma_kernel, conv3d_h2_kernel; fc_backward)
  const float *out_mask;    // channels-last activation tensor of the OUTPUT's shape, stride out_mask_cs; nullptr = no mask
  int out_mask_cs;
  // output channels [out_mask_c0, out_mask_c1) the mask and the maximum apply to (Dense blocks: the launch that writes the
  // final value of the NEXT layer's 16-channel slice of the concat buffer's gradient prepares that slice, after its own
  // out_scale / accumulate; the other channels are written as ever)
  int out_mask_c0, out_mask_c1;
  // conv3d_h2_d16_kernel (conv3d_h2_dense.hip; Dense-block layers on split-format tensors): byte g of d16_taps[s] = the tap
  // (dx * 9 + dy * 3 + dz; 27 = the zero-weight filler) lane group g feeds in step s -- the order the weights are packed in
  unsigned d16_taps[7];
  // persistent launches (conv3d_h2_dense.hip): n_items = (pose pair, tile) items of the launch, set by the launcher;
  // h2_persist = workgroups per CU the launch is capped at, each walking items blockIdx.x, + gridDim.x, ... (0 = one
  // workgroup per item)
  int n_items, h2_persist;
  // conv3d_h2_d16_kernel, GRP variants (launches that leave a workgroup alone on its CU: per-pose calls): the K chunks of a
  // tile are DMA'd d16_group at a time -- one L2 round trip for the group instead of one per chunk -- into as many
  // [tile | weights] sets in LDS; same MFMAs in the same order.  Set by the launcher.
  int d16_group;
  // conv3d_h2_16_ring_kernel (conv3d_h2.hip): ring slots of a launch of few small workgroups; set by the launcher
  int h16_ring;
  // MI_PRECISION_FP16 (the reduced-precision forward mode, BASELINE config 5): the split-fp16 kernels issue the h * h MFMA
  // only -- one of the three per product: activations and weights rounded to fp16, fp32 accumulation
  int h2_honly;
  // conv3d_h2_ws_kernel (conv3d_h2_ws.hip): > 0 = the launch may take the stationary-weights kernel with a ring of this many
  // halo-tile buffers (2 .. 5) where it covers the layer (conv_h2_ws_covers) and the batch is large enough; 0 = never
  int h2_ws;
  int h2_dbg;  // timing experiments only (MI_GNINA_H2_DBG; wrong results): 1 = no chunk-level live test, 2 = no K loop, 4 = no staging, 8 = no weight loads, 16 = no A-operand reads, 32 = the tile's DMA sources are one contiguous run, 64 = no epilogue (d16 / k1s), 128 = no chunk groups in conv3d_h2_d16_kernel (right results)
};

constexpr int kMfmaCountSlots = 1024;

enum { CONV_CFG_4x1_1x3 = 1 /* 1x1 96 -> 96 with one M-tile per wave: <= 128 VGPRs, four waves per SIMD */, CONV_CFG_4x1_2x1 = 0, CONV_CFG_1x4_7x1 = 2, CONV_CFG_N16_TM4 = 3, CONV_CFG_N16_TM3 = 4,
       CONV_CFG_4x1_2x3 = 5, CONV_CFG_4x1_1x5 = 6, CONV_CFG_2x2_3x1 = 7,
       CONV_CFG_4x1_1x1 = 8, CONV_CFG_N16_TM1 = 9, CONV_CFG_N16_TM2 = 10 /* latency variants: one M-tile per wave */ };

// MI355X dispatches workgroups round-robin over its 8 XCDs (workgroup i -> XCD i % 8), each with a private
// 4 MB L2.  Tiles of one pose share halo voxels, K chunks and the candidate list, so a launch re-numbers its
// workgroups such that every XCD works on a CONTIGUOUS range of (pose, tile) ids: XCD x gets ids
// [x * n/8, (x+1) * n/8).  Ids past the last multiple of 8 keep their number.
#if defined(__HIPCC__)
__device__ __forceinline__ int xcd_contiguous_id(int wg, int n) {
  const int per = n >> 3;
  return wg >= (per << 3) ? wg : (wg & 7) * per + (wg >> 3);
}
#endif

// i-th tap (dx * 9 + dy * 3 + dz) of the boustrophedon walk of the 3 x 3 x 3 taps: consecutive taps are face neighbours
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int conv_snake_tap(int i) {
  const int dx = i / 9;
  int r = i - 9 * dx;
  if (dx & 1) r = 8 - r;
  const int dy = r / 3, k = r - 3 * dy;
  return dx * 9 + dy * 3 + ((dy & 1) ? 2 - k : k);
}

size_t conv_lds_bytes(const ConvArgs &p);
void conv_cfg_shape(int cfg, int *wm, int *wn, int *tm, int *tn);
void launch_conv(const ConvArgs &p, int cfg, int B, hipStream_t s);

size_t conv_bf16_lds_bytes(const ConvArgs &p);
void launch_conv_bf16(const ConvArgs &p, int cfg, int B, hipStream_t s);
void launch_gmax_bf16(const void *in, float *out, int B, int C, int in_cs, int out_cs, int S, hipStream_t s);
void launch_gmax_backward_bf16(const void *act, const float *g_out, float *g_in, int B, int C, int in_cs, int out_cs,
                               int S, hipStream_t s);

size_t conv_h2_lds_bytes(const ConvArgs &p);
bool conv_h2_has_cfg(int cfg);
int conv_h2_mt_mask(int cfg);
bool conv_h2_has_bwd_k1(int cfg);  // ... of conv3d_h2_k1_kernel (1x1x1 behind a fused max pool)
bool conv_h2_has_bwd(int cfg);  // the gradient-pass variant of conv3d_h2_kernel exists for this tile shape  // bit m set: conv3d_h2_kernel is compiled with M-tile geometry m (ConvArgs::mt_x) for this shape
void conv_h2_planar_geo(const ConvArgs &p, int *sy, int *sx, int *pl);
void launch_conv_h2(const ConvArgs &p, int cfg, int B, hipStream_t s);
// conv3d_h2_ws.hip: the first 3x3x3 convolution with stationary weights and a ring of halo tiles
bool conv_h2_ws_covers(const ConvArgs &p, int B);
size_t conv_h2_ws_lds_bytes(int ring);
void launch_conv_h2_ws(ConvArgs p, int B, int ring, hipStream_t s);
// conv3d_h2_dense.hip: Dense-block layers / 1x1x1 transitions on split-format tensors
void conv_d16_tap_order(unsigned taps4[7], unsigned char order[28]);
bool conv_d16_layout_conflict_free(int SY, int SX);
size_t conv_h2_d16_lds_bytes(const ConvArgs &p);
void launch_conv_h2_d16(ConvArgs p, int B, hipStream_t s);
size_t conv_h2_k1s_lds_bytes(const ConvArgs &p);
void launch_conv_h2_k1s(ConvArgs p, int B, hipStream_t s);

void launch_zero_cell_probe(const float *in, int B, int C, int cs, int S, unsigned *out, hipStream_t s);
void launch_pool_input(const float *in, float *out, int B, int C, int Cp, int N, int mode, hipStream_t s);
void launch_pool_cl(const float *in, float *out, int B, int C, int in_cs, int out_cs, int S, int mode,
                    hipStream_t s);
void launch_gmax(const float *in, float *out, int B, int C, int in_cs, int out_cs, int S, hipStream_t s);
// global max pool backward: g_in [B][S]^3[in_cs] = g_out[b][c] at the first arg-max voxel of channel c, 0 elsewhere
void launch_gmax_backward(const float *act, const float *g_out, float *g_in, int B, int C, int in_cs, int out_cs,
                          int S, hipStream_t s);
void launch_fc_heads(const float *in, const float *w, const float *bias, int n_in, int skip_softmax,
                     int logistic_loss, float *pose, float *aff, float *loss, float *raw3, int B, hipStream_t s);
void launch_overlap_forward(const float *grid, int B, long N3, float *pose, float *aff, float *loss, float *ave_out,
                            hipStream_t s);
void launch_overlap_backward(const float *grid, const float *ave, int B, long N3, float *gg, hipStream_t s);
// in place over a channel slice: g = act > 0 ? g : 0, amax[b] = max |g| (float bits) -- see conv3d.hip
void launch_grad_mask_amax(float *g, const float *act, int C, int g_cs, int act_cs, long vox_per_pose, int B, unsigned *amax,
                           hipStream_t s);
// (mask: the FC input's forward activation, g_in = mask > 0 ? g : 0; amax [B]: per-pose max |g_in| bits -- both optional)
void launch_fc_backward(const float *raw3, const float *w, int n_in, float *g_in, int B, hipStream_t s,
                        const float *mask = nullptr, unsigned *amax = nullptr);
void launch_unpool_avg(const float *g_pooled, float *g_full, int B, int C, int in_cs, int out_cs, int S,
                       hipStream_t s);
bool gmax_heads_covers(int C);
void launch_gmax_heads(const float *in, float *gmax_out, int B, int C, int in_cs, int out_cs, int S, const float *w, const float *bias,
                       int skip_softmax, int logistic_loss, float *pose, float *aff, float *loss, hipStream_t s);
void launch_zero_u32(unsigned *p, size_t n, hipStream_t s);
void launch_sum_models(const float *src, int n_models, size_t n, float *dst, hipStream_t s);
void launch_ensemble_reduce(const float *pose_m, const float *aff_m, const float *loss_m, int n_models, int B,
                            float *pose, float *aff, float *loss, float *var, hipStream_t s, unsigned *ovf_in = nullptr, unsigned *ovf_out = nullptr);

}  // namespace mig
