// engine.cpp -- device-resident models, scorers and the C ABI (include/mi_gnina.h).
//
// Host-side restatement of what gnina does around the kernels:
//   TorchModel ctor        gninasrc/lib/torch_model.cpp:49-118   -> Model (weights packed for the MFMA kernel)
//   TorchModel::forward    gninasrc/lib/torch_model.cpp:153-224  -> Scorer::run_model (batched, no host syncs)
//   CNNTorchScorer::score  gninasrc/lib/cnn_torch_scorer.cpp:105-198 -> Scorer::score_batch (ensemble mean/variance)
//   DLScorer::setReceptor  gninasrc/lib/dl_scorer.cpp:93-193     -> Scorer::set_receptor (typed once, kept in HBM)
// Differences by design: B poses per call instead of 1, the receptor is typed and
// uploaded once instead of per pose, models with identical type maps share one voxelization,
// results leave the device once per batch instead of three .item() syncs per pose.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <numeric>

#include <mutex>
#include <set>

#include "../../include/mi_gnina.h"
#include "common.h"
#include "conv3d.h"
#include "model.h"
#include "typer.h"
#include "voxelize.h"
#include "options.h"

namespace mig {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }

// The one environment variable this library WRITES, once per process and -- when a pool starts worker threads -- on the
// caller's thread before they exist (mi_pool_create): glibc does not make setenv safe against getenv in other threads.
// The library's own MI_* switches are read from the environment here as well, once (options.h): nothing in the engine calls
// getenv afterwards.
void process_env_once() {
  static std::once_flag once;
  std::call_once(once, [] {
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    (void)option(OPT_MI_GNINA_NO_H2);
  });
}

// Scoring calls of different scorers on one device run side by side (each scorer on its own stream, driven by its own host
// thread) -- which is how gnina drives the seam: fresh_copy() hands every worker thread and every Monte-Carlo task a NEW
// CNNTorchScorer with a NEW mutex (cnn_torch_scorer.h:54, dl_scorer.h:43-44, main.cpp:1436-1438, parallel_mc.cpp:145-146).
//
// Rounds 5 took a per-device lock here because two scorers on two hardware queues did not reproduce their single-thread
// bits (~5 % of B = 1 calls, up to 6e-2 in the affinity).  Round 6 found what deviated and why the lock only hid it (DESIGN
// "concurrency"): voxelize_tiles' squared distances, formed with PACKED-fp32 instructions (v_pk_add_f32 / v_pk_mul_f32), came
// out wrong in the upper half of a wavefront (lanes 32-63) in ~2 % of its launches whenever the Dense family's f16-MFMA conv
// K loops shared the SIMDs -- lanes outside an atom's support then passed the range test and added e^-2 (2 d/r - 3)^2.
// Evidence: the voxelizer alone as the victim (mi_debug_vox_stress, every launch compared on the device), an in-kernel
// trap that cleared the lists / the scalar loads / the LDS / the exec mask, and builds of the kernel that differ in one thing:
// without packed-fp32 instructions 0 of 50,000 launches and 0 of 9,000 two-thread calls deviate, with them (any operand
// form, scalar load in flight or not, wait states added) ~1.4-2 %.  voxelize.hip is therefore compiled without them
// (gnina_amd/build.py, tests/test_cabi_cpu.py checks the device code).  MI_GNINA_CALL_LOCK=1 brings the old serialisation
// back for A/B measurements; nothing depends on it.
static std::recursive_mutex &device_call_lock(int device) {
  static std::recursive_mutex locks[64];
  if (!option(OPT_MI_GNINA_CALL_LOCK) || atoi(option(OPT_MI_GNINA_CALL_LOCK)) == 0) {
    static thread_local std::recursive_mutex own;  // (no serialisation: a lock nobody else takes)
    return own;
  }
  return locks[device >= 0 && device < 64 ? device : 0];
}

// Lanes or not?  A call of at most eight poses may spread an ensemble's models over the device's lane streams (priority streams
// shared by all scorers of the device).  That is the fastest thing to do for a scorer that has the device to itself; when other
// scorers are at work on it as well -- gnina's worker threads, each with its fresh_copy() -- the lanes of several scorers queue
// up behind each other on the shared streams, and every call stays on its scorer's own stream.  Measured on gnina's default
// ensemble at B = 1, poses/s from 1 / 2 / 4 host threads (bench.py's seam_b1, round 6): this rule 1,661 / 1,703 / 3,300; lanes
// always 1,640 / 2,062 / 1,066; never 881 / 1,734 / 1,985.  (A mixed population -- some calls on lanes, some not -- is the worst
// regime, 1,010 from two threads: hence a rule all scorers of a device agree on, "did another scorer begin a call here within
// the last 20 ms".  Which hardware queue a stream sits on matters as much as the rule: Scorer::lane_offset.)
struct DeviceActivity {  // the scorers seen lately on a device: (scorer, start of its last call), a handful of slots
  static constexpr int kSlots = 16;
  std::atomic<const void *> scorer[kSlots];
  std::atomic<long long> last_ns[kSlots];
};
static DeviceActivity g_activity[64];
static bool scorer_alone_on_device(int device, const void *scorer) {
  DeviceActivity &a = g_activity[device >= 0 && device < 64 ? device : 0];
  const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  int others = 0;
  int mine = -1, oldest = 0;
  for (int i = 0; i < DeviceActivity::kSlots; i++) {
    const void *sc = a.scorer[i].load(std::memory_order_relaxed);
    const long long t = a.last_ns[i].load(std::memory_order_relaxed);
    if (sc == scorer) mine = i;
    else if (sc != nullptr && now - t < 20000000ll) others++;
    if (t < a.last_ns[oldest].load(std::memory_order_relaxed)) oldest = i;
  }
  if (mine < 0) mine = oldest;  // (a race between two new scorers for one slot costs one of them a stale entry for one call)
  a.scorer[mine].store(scorer, std::memory_order_relaxed);
  a.last_ns[mine].store(now, std::memory_order_relaxed);
  return others <= (option(OPT_MI_GNINA_LANES_OTHERS) ? atoi(option(OPT_MI_GNINA_LANES_OTHERS)) : 0);  // (MI_GNINA_LANES_OTHERS: A/B switch)
}

// The lane streams of a device (see Scorer::lane_streams), shared by all its scorers: a set per scorer is harmful -- four
// scorers x three priority streams oversubscribe the hardware
// queues and the runtime time-slices them (gnina's default ensemble from four threads: 777 -> 110 poses/s).  One stream
// priority per lane, cycling through the device's range: the HIP runtime keeps a pool of hardware queues per priority, so
// lanes of different priority never share a hardware queue.  (With plain streams and the default four queues per pool, two
// of the three lanes of gnina's default ensemble landed on one queue and ran one after the other: 1,289 us per B = 1 call
// against 726 us with GPU_MAX_HW_QUEUES=8, 717 us with priorities.)  Thread-safe; never destroyed.
static hipStream_t device_lane_stream(int device, int k) {
  static std::mutex mu;
  static std::vector<hipStream_t> pool[64];
  std::lock_guard<std::mutex> lock(mu);
  std::vector<hipStream_t> &v = pool[device >= 0 && device < 64 ? device : 0];
  // (a stream belongs to the device that is current when it is created: a caller thread that is on another device --
  // mi_gnina_init not called in it -- must not put the lanes there)
  struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int want) {
      int cur = -1;
      if (hipGetDevice(&cur) == hipSuccess && cur != want && want >= 0 && hipSetDevice(want) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() {
      if (prev >= 0) (void)hipSetDevice(prev);
    }
  };
  while ((int)v.size() <= k) {
    DeviceGuard on_device(device);
    int least = 0, greatest = 0;
    MIG_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int span = least - greatest + 1;
    hipStream_t st = nullptr;
    MIG_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, span > 1 ? greatest + (int)v.size() % span : 0));
    v.push_back(st);
  }
  return v[k];
}

void ensure_max_lds(const void *kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void *>> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  if (done.insert({dev, kernel}).second)
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// -------------------------------------------------------------------------------------------
// Model: parsed program + device weights
// -------------------------------------------------------------------------------------------
struct ConvPlan {
  int cfg = 0;
  int cin = 0;           // real (unpadded) input channels, for FLOP accounting
  ConvArgs a{};          // device pointers to weights filled at load; in/out patched per run
  int src = -1, dst = -1;
  // latency variant for small batches (a per-pose DLScorer::score call is B = 1): same K chunking and packed
  // weights, smaller spatial tiles and one M-tile per wave, so that one pose still spreads over the chip
  bool has_lat = false;
  int lat_cfg = 0;
  int lat_tc[3] = {0, 0, 0};
  // a coarser latency tile for launches the finest one would cut into more than 512 workgroups (6^3 Dense layers)
  bool has_lat2 = false;
  int lat2_cfg = 0;
  int lat2_tc[3] = {0, 0, 0};
  // the same layer on the split-fp16 kernels (conv3d_h2.hip; forward, same tiles, own K chunking and packed weights)
  bool has_h2 = false;
  ConvArgs h2{};
  // 3x3x3 layers on conv3d_h2_kernel (planar, double-buffered halo tile; may read / write split-format tensors): M-tile
  // geometry and LDS pad slots of the throughput tile live in h2 (mt_x, h2_pad_*), those of the latency tile here
  bool h2_planar = false;
  int h2_lat_mt = 0, h2_lat_pad[2] = {0, 0};
  int h2_cfg = -1;  // >= 0: the throughput launch takes this kernel shape and h2's own tile (tc*, nt*) instead of the fp32 plan's
  // Dense family on split-format tensors (conv3d_h2_dense.hip), taken when the layer's buffers are split (Model::buf_split):
  // a Dense-block layer's own plan (BatchNorm folded into the weights / a border-class bias table, one octet per K chunk,
  // its own tile), and "the 1x1x1 plan in h2 can run on conv3d_h2_k1s_kernel" (same packed weights, split-format input)
  bool has_d16 = false;
  ConvArgs d16{};
  // the d16 plan's latency tile (2 x 2 x 2 cells = four M-tiles, one per wave) for launches its throughput tile would cut
  // into fewer workgroups than the chip has CUs; same packed weights, same K order
  int d16_lat_tc[3] = {0, 0, 0};
  bool has_k1s = false;
};

// Tile geometry of a launch of `nb` poses: the throughput plan, or the latency variant when the throughput
// plan would leave most of the 256 CUs idle.
static void pick_tile(const ConvPlan &cp, int nb, ConvArgs &a, int &cfg) {
  cfg = cp.cfg;
  if (!cp.has_lat) return;
  int wm, wn, tm, tn;
  conv_cfg_shape(cp.cfg, &wm, &wn, &tm, &tn);
  const long groups = cdiv(a.coutp / (cp.cfg == CONV_CFG_N16_TM4 || cp.cfg == CONV_CFG_N16_TM3 ? 16 : 32), wn * tn);
  const long blocks = (long)nb * a.ntx * a.nty * a.ntz * groups;
  if (blocks >= 512 || option(OPT_MI_GNINA_NO_LAT)) return;
  if (a.sparse == 1) return;  // the zero-quad skipping pairs up surviving quads per tile: keep one tiling so results do not depend on the batch size
  if (a.post_w) {  // the fused 1x1 conv needs every mid channel inside one workgroup
    int lwm, lwn, ltm, ltn;
    conv_cfg_shape(cp.lat_cfg, &lwm, &lwn, &ltm, &ltn);
    if (a.coutp / 32 > lwn * ltn) return;
    a.post_rows = lwm * ltm * 32;
  }
  cfg = cp.lat_cfg;
  a.tcx = cp.lat_tc[0], a.tcy = cp.lat_tc[1], a.tcz = cp.lat_tc[2];
  a.mt_x = 0;  // (the latency tiles keep the raster order of cells)
  const int cells = a.S / 2;
  if (cp.has_lat2 && (long)nb * cdiv(cells, a.tcx) * cdiv(cells, a.tcy) * cdiv(cells, a.tcz) * groups > 512) {
    cfg = cp.lat2_cfg;
    a.tcx = cp.lat2_tc[0], a.tcy = cp.lat2_tc[1], a.tcz = cp.lat2_tc[2];
  }
  a.ntx = cdiv(cells, a.tcx);
  a.nty = cdiv(cells, a.tcy);
  a.ntz = cdiv(cells, a.tcz);
}

struct Step {
  OpKind kind;
  ConvPlan conv;         // Conv
  ConvPlan bwd;          // Conv: transposed twin for the gradient pass (grad program only)
  bool has_bwd = false;
  // The transposed FIRST conv computes d loss / d (pooled grid); voxel_backward then reads the channels of the atoms that
  // move.  With a rigid receptor those are the ligand's channels only: bwd_lig is the same transposed conv restricted to
  // output channels [lig_c0, C) -- for Default2017 (16 + 19 channels) one 32-wide N tile instead of two (35 -> 64).
  ConvPlan bwd_lig;
  bool has_bwd_lig = false;
  // The transposed conv on the split-fp16 kernel (3x3x3 behind a ReLU; its own plan: the one tile shape the gradient-pass
  // variant is compiled for).  bwd_h2_kind 1: the conv's private output -- the producer of that buffer's gradient applies the
  // ReLU mask and records the per-pose maximum (Model::buf_bwd_h2); 2: a channel slice of a concat buffer (Dense blocks) --
  // a small kernel does both ahead of the launch (launch_grad_mask_amax).  bwd_lig_h2: the ligand-channel restriction.
  ConvPlan bwd_h2;
  int bwd_h2_kind = 0;
  ConvPlan bwd_lig_h2;
  bool has_bwd_lig_h2 = false;
  int lig_c0 = 0;
  bool has_bn = false;
  int src = -1, dst = -1;
  int pool_mode = 0;     // Pool
  int C = 0;             // channels moved by Pool/GMax
  long w_off = 0, b_off = 0;  // Fc (offsets into dev_data)
  int n_in = 0;
  int post_cout = 0;      // Conv with a fused 1x1x1 conv behind it: that conv's output channels
  bool src_bf16 = false;  // bf16 program: GMax reads a bf16 tensor
  // ReLU'd conv layer: is its input sparse enough for the per-MFMA zero test to pay?  0 = not measured yet, 1 = yes,
  // 2 = no.  Measured once, on the first launch of >= 32 poses (launch_zero_cell_probe); either answer gives the same
  // bits (ConvArgs::sparse 2 / 3), so the plain int shared by the scorers of a model is a benign race.
  mutable int relu_skip = 0;
};

struct Model {
  std::atomic<int> refs{1};
  ModelDesc d;
  int N = 0;             // grid points per side
  int C = 0, Cp = 0;     // input channels / padded
  int input_pool = 0;    // 1 max, 2 avg: first op, fused into the voxelizer
  int input_dst = -1;    // buffer receiving the pooled grid
  std::vector<int> buf_cp;          // padded channel stride per buffer (0 = never materialised)
  std::vector<Step> steps;          // executable program after fusion (split-fp16 kernels where planned)
  std::vector<Step> steps32;        // the same with fp32 MFMA in every layer and the fused 3x3x3 + 1x1x1 pairs (MI_PRECISION_FP32_MFMA)
  std::vector<Step> gsteps;         // gradient-capable program (avg pools unfused, transposed convs planned)
  std::vector<char> op_h2;          // per op of the description: the forward program runs it on the split-fp16 kernels
  // Split-format tensors of the forward program `steps` (ConvArgs::in_split): buf_split[id] = activation buffer id is
  // written split by its producer (a conv3d_h2_kernel epilogue) because every layer reading it is a conv3d_h2_kernel layer
  // without BatchNorm; pooled_split_ok = the same holds for the pooled voxel grid, which the voxelizer then writes split
  // with a channel stride of Cp8 (whole octets).  Every other program (fp32 MFMA, gradient, bf16) keeps fp32 tensors.
  // gradient program: the buffer is the private, ReLU'd output of ONE conv whose transposed twin has a split-fp16 plan -- the
  // producer of its gradient then masks by the activation and records the per-pose maximum (ConvArgs::out_mask / out_amax)
  std::vector<char> buf_bwd_h2;
  std::vector<char> buf_split;
  bool pooled_split_ok = false;
  int Cp8 = 0;
  std::vector<Step> hsteps;         // bf16-MFMA forward program (built on first use, mi_scorer_set_precision)
  std::vector<Step> hgsteps;        // bf16-MFMA gradient program (max-pool networks: Default2017, Dense)
  std::once_flag hsteps_once, hgsteps_once;
  std::string hsteps_error, hgsteps_error;
  bool grad_supported = false;
  std::string grad_unsupported_reason;
  bool overlap = false;             // the parameter-free Overlap toy (mean of rec * lig density over the full grid)
  DevBuf<float> dev_data;           // fc weights, biases, bn params (canonical payload)
  std::vector<std::unique_ptr<DevBuf<float>>> packed;  // packed conv weights / padded bias / bn
  float qa, qb, qc;                 // quadratic tail coefficients
  DensityConsts dens[kNumSminaTypes];
};

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static const float *push_dev(Model &m, const std::vector<float> &host) {
  m.packed.emplace_back(new DevBuf<float>());
  auto &buf = *m.packed.back();
  buf.ensure(host.size());
  MIG_HIP(hipMemcpy(buf.p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  return buf.p;
}

// Transposed ("backward-data") twin of a forward conv: same geometry, channels swapped, taps flipped.
static Op make_bwd_op(const Op &o) {
  Op b = o;
  b.cin = o.cout;
  b.cout = o.cin;
  b.relu = 0;
  b.bn_scale_off = b.bn_shift_off = -1;
  b.dst_c0 = 0;
  return b;
}

// (force_std: the 4 x 1 waves x 2 M-tiles shape whatever the layer -- the only one conv3d_h2_kernel's gradient-pass variant
// is compiled for)
static void plan_conv(Model &m, const Op &o, ConvPlan &cp, int pool_mode, int dst_buf, int dst_c0,
                      bool backward = false, bool n16_backward = false, bool force_std = false, bool k1_wide = false) {
  const int S = m.d.bufs[o.src].S;
  MIG_CHECK(S % 2 == 0, 2, "conv spatial size must be even");
  const int cells = S / 2;
  const int NT = cdiv(o.cout, 32);
  ConvArgs &a = cp.a;
  a.S = S;
  a.cout = o.cout;
  a.coutp = NT * 32;
  a.ksize = o.ksize;
  a.relu = o.relu;
  a.pool = pool_mode;
  a.out_c0 = dst_c0;
  // workgroup shape
  // Dense-block convs: 16-wide MFMA tiles -- and (n16_backward) a transposed first conv that only has to produce the <= 16
  // ligand channels of the grid gradient (ReLU-masked input, in_mode 1, which the 16-wide kernel stages; not the max-unpool)
  const bool n16 = (o.cout == 16 && pool_mode == 0 && !backward) || (backward && n16_backward && o.cout <= 16);
  if (n16) {
    a.coutp = 16;
    if (cells == 6) {
      cp.cfg = CONV_CFG_N16_TM3;  // 4 waves x 3 M-tiles x 2 cells = 24 cells = 2x2x6
      a.tcx = 2, a.tcy = 2, a.tcz = 6;
    } else {
      cp.cfg = CONV_CFG_N16_TM4;  // 4 waves x 4 M-tiles x 2 cells = 32 cells
      if (cells == 3) a.tcx = 3, a.tcy = 3, a.tcz = 3;
      else a.tcx = 2, a.tcy = 4, a.tcz = 4;
    }
  } else if (o.ksize == 1 && NT == 3 && cells % 4 == 0 && (!backward || k1_wide)) {
    // 1x1 bottleneck 96 -> 96: all output channels in one workgroup (input read once); one M-tile per wave keeps the
    // kernel at 128 VGPRs = four waves per SIMD (two M-tiles: 191 VGPRs, two waves, 3.54 against 3.32 ms at 24^3)
    cp.cfg = CONV_CFG_4x1_1x3;
    a.tcx = 2, a.tcy = 2, a.tcz = 4;
  } else if (o.ksize == 1 && NT == 5 && (!backward || k1_wide)) {
    cp.cfg = CONV_CFG_4x1_1x5;  // 1x1 bottleneck 160 -> 160: 4 waves x 1 M-tile = 16 cells
    if (cells == 6) a.tcx = 2, a.tcy = 2, a.tcz = 3;  // 12 cells used of 16
    else a.tcx = 2, a.tcy = 2, a.tcz = 4;
  } else if (cells == 3 && NT % 4 == 0 && !force_std) {
    cp.cfg = CONV_CFG_1x4_7x1;
    a.tcx = a.tcy = a.tcz = 3;
  } else if (cells == 6 && NT % 2 == 0 && !force_std) {
    cp.cfg = CONV_CFG_2x2_3x1;  // 24 cells = 6 M-tiles: 2 x 2 waves x 3 M-tiles, one wave per SIMD (a 3 x 2 = 6-wave
                                // workgroup loads two SIMDs twice as much as the other two: measured 2.08 -> 1.71 ms)
    a.tcx = 2, a.tcy = 2, a.tcz = 6;
  } else {
    cp.cfg = CONV_CFG_4x1_2x1;  // 8 M-tiles = 32 cells
    if (cells % 4 == 0 && o.ksize == 3 && !option(OPT_MI_GNINA_NO_MT_X))
      a.tcx = 4, a.tcy = 4, a.tcz = 2, a.mt_x = 1;  // M-tiles stacked along x: conflict-free A-operand reads (conv3d.h)
    else if (cells % 4 == 0) a.tcx = 2, a.tcy = 4, a.tcz = 4;
    else if (cells == 6) a.tcx = 2, a.tcy = 2, a.tcz = 6;
    else if (cells == 3) a.tcx = 3, a.tcy = 3, a.tcz = 3;
    else a.tcx = 2, a.tcy = 4, a.tcz = 4;
  }
  if (const char *ev = option(OPT_MI_GNINA_K1_TILE)) {  // experiment: tile of the 1x1x1 transitions, "x,y,z" cells (<= 16)
    int tx_ = 0, ty_ = 0, tz_ = 0;
    if (o.ksize == 1 && !backward && (cp.cfg == CONV_CFG_4x1_1x3 || cp.cfg == CONV_CFG_4x1_1x5) && sscanf(ev, "%d,%d,%d", &tx_, &ty_, &tz_) == 3 &&
        tx_ > 0 && ty_ > 0 && tz_ > 0 && tx_ * ty_ * tz_ <= 16 && cells % tz_ == 0)
      a.tcx = tx_, a.tcy = ty_, a.tcz = tz_;
  }
  a.ntx = cdiv(cells, a.tcx);
  a.nty = cdiv(cells, a.tcy);
  a.ntz = cdiv(cells, a.tcz);
  // K chunking: largest divisor of cin4 whose halo tile fits the LDS budget
  const int cin4 = cdiv(o.cin, 4);
  if (!backward) MIG_CHECK(cin4 * 4 <= round_up(m.d.bufs[o.src].C, 4), 2, "conv input channels exceed buffer");
  if (backward) MIG_CHECK(o.cin % 4 == 0, 2, "backward conv needs the forward Cout to be a multiple of 4");
  const int halo = o.ksize == 3 ? 1 : 0;
  const size_t HV = (size_t)(2 * a.tcx + 2 * halo) * (2 * a.tcy + 2 * halo) * (2 * a.tcz + 2 * halo);
  // K chunking over input-channel quads: the halo tile of one chunk must fit the LDS budget.  Convs that
  // read the sparse pooled voxel grid get a small budget (more workgroups per CU, finer zero-skipping);
  // the last chunk may be partial (its missing quads are zero-filled and, on the sparse path, skipped).
  // 40 KB per workgroup = 3-4 workgroups per CU whose staging and MFMA phases overlap (measured on the dense
  // layers: 72 KB 20.3k, 52 KB 20.8k, 36 KB 20.8k poses/s; MI_GNINA_LDS_KB overrides for tuning)
  const bool sparse_budget = o.src == m.input_dst && !backward && o.bn_scale_off < 0;
  size_t budget = 40 * 1024;
  // (the 12^3 Dense-block layers -- 27 workgroups per pose, K loops of a few hundred MFMAs -- gain 10 % from a fifth
  // workgroup per CU: 28 KB measured 0.98 -> 0.87 ms for 96 -> 16, while the 24^3 layers lose 3-5 % with it)
  if (n16 && cells == 6) budget = 28 * 1024;
  if (!sparse_budget && option(OPT_MI_GNINA_LDS_KB) && atoi(option(OPT_MI_GNINA_LDS_KB)) > 0)
    budget = (size_t)atoi(option(OPT_MI_GNINA_LDS_KB)) * 1024;
  if (sparse_budget && option(OPT_MI_GNINA_SPARSE_LDS_KB) && atoi(option(OPT_MI_GNINA_SPARSE_LDS_KB)) > 0)
    budget = (size_t)atoi(option(OPT_MI_GNINA_SPARSE_LDS_KB)) * 1024;
  int best = 1;
  for (int c = 1; c <= cin4; c++) {
    if (!sparse_budget && cin4 % c) continue;  // dense layers: equal chunks only (no wasted MFMAs)
    int ccs = 4 * (c | 1);
    if (HV * ccs * 4 + 27 * c * 4 <= budget) best = c;
  }
  a.cc4 = best;
  a.ccs = 4 * (best | 1);  // odd quad stride: consecutive voxels land on different 16-byte LDS slots
  a.nchunks = cdiv(cin4, best);
  a.cin4 = cin4;
  // pack weights [chunk][pair][2][coutp][4]
  const int taps = o.ksize * o.ksize * o.ksize;
  const int kstep = n16 ? 4 : 2;  // quads per MFMA step: pairs for 32x32x2, quartets for 16x16x4
  // (the 32x32x2 kernel wants an all-zero row behind the last quad of every chunk: see ConvArgs::wrows)
  const int Q = taps * a.cc4, P = n16 ? (Q + kstep - 1) / kstep + 1 : Q / kstep + 1;  // (+ a zero quartet / zero rows)
  a.wrows = P * kstep;
  std::vector<float> wp((size_t)a.nchunks * P * kstep * a.coutp * 4, 0.f);
  // forward: canonical [tap][cin][cout]; backward: W'[tap][co][ci] = W[taps-1-tap][ci][co] (flipped, transposed)
  std::vector<float> wT;
  const float *w = m.d.data.data() + o.w_off;
  if (backward) {
    const int fci = o.cout, fco = o.cin;  // forward cin / cout
    wT.resize((size_t)taps * o.cin * o.cout);
    for (int t = 0; t < taps; t++)
      for (int ci = 0; ci < fci; ci++)
        for (int co = 0; co < fco; co++)
          wT[((size_t)t * fco + co) * fci + ci] = w[((size_t)(taps - 1 - t) * fci + ci) * fco + co];
    w = wT.data();
  }
  // Dense-block convs (N = 16 kernel, BatchNorm on the input): channel-major K order + per-MFMA zero test, the BN shift
  // through a border-class bias table (ConvArgs::korder / bias_tab)
  const bool n16_skip = n16 && o.ksize == 3 && o.bn_scale_off >= 0 && !option(OPT_MI_GNINA_NO_RELU_SKIP);
  a.korder = n16_skip ? 1 : 0;
  for (int ch = 0; ch < a.nchunks; ch++)
    for (int pr = 0; pr < P; pr++)
      for (int kh = 0; kh < kstep; kh++) {
        int q = kstep * pr + kh;
        if (q >= Q) continue;
        int tap = q / a.cc4, c4 = q % a.cc4;
        if (a.korder) c4 = q / taps, tap = taps == 27 ? conv_snake_tap(q % taps) : q % taps;
        for (int j = 0; j < 4; j++) {
          int c = (ch * a.cc4 + c4) * 4 + j;
          if (c >= o.cin) continue;
          for (int n = 0; n < o.cout; n++)
            wp[((((size_t)ch * P + pr) * kstep + kh) * a.coutp + n) * 4 + j] = w[((size_t)tap * o.cin + c) * o.cout + n];
        }
      }
  a.wp = push_dev(m, wp);
  std::vector<float> bias(a.coutp, 0.f);
  if (!backward) std::copy(m.d.data.begin() + o.b_off, m.d.data.begin() + o.b_off + o.cout, bias.begin());
  a.bias = push_dev(m, bias);
  a.bn_scale = a.bn_shift = nullptr;
  a.bias_tab = nullptr;
  a.in_mode = 0;
  a.sparse = 0;
  a.in_argmax = nullptr;
  a.in_act = nullptr;
  a.in_act_cs = 0;
  a.argmax_out = nullptr;
  a.out_scale = nullptr;
  a.accumulate = 0;
  a.post_w = nullptr;
  a.post_bias = nullptr;
  a.post_relu = 0;
  a.post_rows = 0;
  a.post_cc4 = a.post_wrows = 0;
  if (o.bn_scale_off >= 0) {
    std::vector<float> sc(cin4 * 4, 0.f), sh(cin4 * 4, 0.f);
    std::copy(m.d.data.begin() + o.bn_scale_off, m.d.data.begin() + o.bn_scale_off + o.cin, sc.begin());
    std::copy(m.d.data.begin() + o.bn_shift_off, m.d.data.begin() + o.bn_shift_off + o.cin, sh.begin());
    if (n16_skip) {
      // y = sum (x * sc + sh) * W = sum (x * sc) * W + sum_{taps inside the grid} sh * W: the second sum depends on the
      // output voxel only through which of its 27 taps fall outside the zero-padded grid (padding is 0 AFTER BatchNorm)
      std::vector<float> tab(27 * 16, 0.f);
      for (int cls = 0; cls < 27; cls++) {
        const int e[3] = {cls / 9, (cls / 3) % 3, cls % 3};  // per axis: 0 first plane, 1 interior, 2 last plane
        for (int n = 0; n < o.cout; n++) {
          double acc = bias[n];
          for (int t = 0; t < 27; t++) {
            const int d[3] = {t / 9 - 1, (t / 3) % 3 - 1, t % 3 - 1};
            bool inside = true;
            for (int ax = 0; ax < 3; ax++)
              if ((e[ax] == 0 && d[ax] < 0) || (e[ax] == 2 && d[ax] > 0)) inside = false;
            if (!inside) continue;
            for (int c = 0; c < o.cin; c++) acc += (double)sh[c] * (double)w[((size_t)t * o.cin + c) * o.cout + n];
          }
          tab[cls * 16 + n] = (float)acc;
        }
      }
      a.bias_tab = push_dev(m, tab);
      std::fill(sh.begin(), sh.end(), 0.f);
      a.sparse = 2;
    }
    a.bn_scale = push_dev(m, sc);
    a.bn_shift = push_dev(m, sh);
  }
  cp.src = o.src;
  cp.dst = dst_buf;
  cp.cin = o.cin;
  MIG_CHECK(conv_lds_bytes(a) <= 160 * 1024, 2, "conv tile exceeds LDS");
  // latency variant: 4 waves x 1 M-tile (32-wide kernel: 16 cells; 16-wide kernel: 8 cells, or 16 at 6^3)
  cp.has_lat = true;
  if (n16) {
    if (cells % 2 == 0) {
      cp.lat_cfg = CONV_CFG_N16_TM1;
      cp.lat_tc[0] = 2, cp.lat_tc[1] = 2, cp.lat_tc[2] = 2;
    } else {
      if (cells == 3) {
        // 6^3: nine workgroups of 1 x 1 x 3 cells per pose (two waves with cells).  The K loop of the 16-wide kernel is bound by
        // LDS reads (every A byte feeds 16 output channels only): three workgroups of nine cells ran a chunk's 14 steps in
        // 2.4 us, and the way to more LDS bandwidth is more CUs
        cp.lat_cfg = CONV_CFG_N16_TM1;
        cp.lat_tc[0] = 1, cp.lat_tc[1] = 1, cp.lat_tc[2] = 3;
        cp.has_lat2 = true;  // 4 waves x 2 M-tiles x 2 cells = 16 cells >= 1 x 3 x 3: three workgroups per pose
        cp.lat2_cfg = CONV_CFG_N16_TM2;
        cp.lat2_tc[0] = 1, cp.lat2_tc[1] = 3, cp.lat2_tc[2] = 3;
      } else {
        cp.lat_cfg = CONV_CFG_N16_TM2;  // 4 waves x 2 M-tiles x 2 cells = 16 cells
        cp.lat_tc[0] = 1, cp.lat_tc[1] = 2, cp.lat_tc[2] = 4;
      }
    }
  } else {
    cp.lat_cfg = CONV_CFG_4x1_1x1;  // 4 M-tiles = 16 cells, one 32-wide N tile per workgroup
    if (cells % 4 == 0) cp.lat_tc[0] = 2, cp.lat_tc[1] = 2, cp.lat_tc[2] = 4;
    else if (cells == 6) cp.lat_tc[0] = 2, cp.lat_tc[1] = 2, cp.lat_tc[2] = 3;
    else if (cells == 3) cp.lat_tc[0] = 1, cp.lat_tc[1] = 3, cp.lat_tc[2] = 3;
    else cp.lat_tc[0] = 2, cp.lat_tc[1] = 2, cp.lat_tc[2] = 4;
  }
}

static void plan_conv_h2(Model &m, const Op &o, ConvPlan &cp, bool fused_post = false, bool backward = false);

// ---- split-fp16 twin of a forward conv plan (conv3d_h2.hip) ----
static unsigned short host_f2h(float f) {  // fp32 -> fp16, round to nearest even, subnormals and overflow handled
  unsigned u;
  memcpy(&u, &f, 4);
  const unsigned sign = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u >= 0x7f800000u) return (unsigned short)(sign | (u > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (u >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);  // rounds to >= 2^16: infinity
  if (u < 0x33000001u) return (unsigned short)sign;               // <= 2^-25: rounds to zero
  int e = (int)(u >> 23) - 127;
  unsigned man = (u & 0x7fffffu) | 0x800000u;  // 24 significant bits
  int shift = e >= -14 ? 13 : 13 + (-14 - e);  // bits dropped (subnormal results drop more)
  unsigned q = man >> shift, rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) q++;
  // q holds the implicit bit for normals (bit 10): adding the biased exponent minus one folds a carry in correctly
  const unsigned out = e >= -14 ? ((unsigned)(e + 15 - 1) << 10) + q : q;
  return (unsigned short)(sign | out);
}
static float host_h2f(unsigned short h) {
  const unsigned sign = (unsigned)(h & 0x8000u) << 16;
  const int e = (h >> 10) & 0x1f;
  const unsigned man = h & 0x3ffu;
  float v;
  if (e == 0) v = ldexpf((float)man, -24);
  else if (e == 31) v = man ? NAN : INFINITY;
  else v = ldexpf((float)(man | 0x400u), e - 25);
  return sign ? -v : v;
}

// per-layer power of two that lifts max |w| into [2^13, 2^14): the weights' low halves stay normal fp16 numbers
static float h2_weight_scale(const float *w, size_t n) {
  float wmax = 0.f;
  for (size_t i = 0; i < n; i++) wmax = std::max(wmax, fabsf(w[i]));
  return wmax > 0.f ? ldexpf(1.f, 14 - (int)floorf(log2f(wmax)) - 1) : 1.f;
}

// LDS layout of conv3d_h2_kernel's planar halo tile for a tile of tc cells and a workgroup of wm x tm M-tiles: which four
// cells form an M-tile (0 raster, 1 stacked along x, 2 a 2 x 2 square in (x, y)) and how many 16-byte pad slots follow every
// z-row / x-plane, chosen with the bank model of the A-operand reads (MI355X: a ds_read_b128 is served in four groups of 16
// lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- and a group takes as many LDS cycles as the
// largest number of its lanes that hit one slot modulo 16 at different addresses).  All lanes of a half-wave add the same
// tap offset, so the model only needs the lanes' base slots.  Smallest mean cycles, then smallest tile.
static void h2_choose_layout(const int tc[3], int n_mtiles, int mt_mask, int &mt_out, int &pad_y, int &pad_x, double *cycles_out = nullptr) {
  const long max_slots = 8 * 256;  // the kernel's staging: at most eight wave-DMAs per thread and chunk (conv3d_h2.hip kH2NS)
  static const int kGroup[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
  const int tcx = tc[0], tcy = tc[1], tcz = tc[2], NC = tcx * tcy * tcz;
  const int HX = 2 * tcx + 2, HY = 2 * tcy + 2, HZ = 2 * tcz + 2;
  double best = 1e30;
  long best_size = 0;
  mt_out = 0, pad_y = 0, pad_x = 0;
  for (int mt = 0; mt < 3; mt++) {
    if (!((mt_mask >> mt) & 1)) continue;  // (geometries compiled for the kernel shape: conv_h2_mt_mask)
    if (mt == 1 && (tcx % 4 != 0 || option(OPT_MI_GNINA_NO_MT_X))) continue;
    if (mt == 2 && !(tcx == 2 && tcy == 2)) continue;
    const int n_mt = mt == 1 ? tcx / 4 * tcy * tcz : mt == 2 ? tcz : cdiv(NC, 4);
    if (n_mt > n_mtiles) continue;
    for (int py = 0; py < 16; py++)
      for (int px = 0; px < 16; px++) {
        const int SY = HZ + py, SX = HY * SY + px;
        const long size = (long)((HX * SX + 31) & ~31);
        if (2 * size > max_slots) continue;
        double tot = 0;
        for (int mtile = 0; mtile < n_mt; mtile++) {
          int base[32];
          for (int row = 0; row < 32; row++) {
            const int oz = row & 1, oy = (row >> 1) & 1, ox = (row >> 3) & 1, cim = ((row >> 2) & 1) + 2 * ((row >> 4) & 1);
            int cx, cy, cz;
            if (mt == 1) cz = mtile % tcz, cy = (mtile / tcz) % tcy, cx = 4 * (mtile / (tcz * tcy)) + cim;
            else if (mt == 2) cz = mtile, cy = cim & 1, cx = cim >> 1;
            else {
              const int cell = std::min(mtile * 4 + cim, NC - 1);
              cz = cell % tcz, cy = (cell / tcz) % tcy, cx = cell / (tcz * tcy);
            }
            base[row] = (2 * cx + ox) * SX + (2 * cy + oy) * SY + (2 * cz + oz);
          }
          for (int g = 0; g < 2; g++) {
            int worst = 0;
            for (int i = 0; i < 16; i++) {
              int distinct = 0;  // different addresses of this group in lane i's slot class, counted once (at their first lane)
              bool first = true;
              for (int j = 0; j < i; j++)
                if (base[kGroup[g][j]] == base[kGroup[g][i]]) first = false;
              if (!first) continue;
              for (int j = 0; j < 16; j++) {
                if ((base[kGroup[g][j]] - base[kGroup[g][i]]) % 16 != 0) continue;
                bool seen = false;
                for (int k = 0; k < j; k++)
                  if (base[kGroup[g][k]] == base[kGroup[g][j]]) seen = true;
                if (!seen) distinct++;
              }
              worst = std::max(worst, distinct);
            }
            tot += worst;
          }
        }
        const double mean = tot / (2.0 * n_mt);
        if (mean < best - 1e-9 || (mean < best + 1e-9 && size < best_size)) best = mean, best_size = size, mt_out = mt, pad_y = py, pad_x = px;
      }
  }
  if (cycles_out) *cycles_out = best;
}

// (backward: `o` is make_bwd_op of a forward conv and `cp` its transposed plan -- weights flipped and transposed as in
// plan_conv; the launch stages an fp32 gradient tensor with ConvArgs::in_amax scaling)
static void plan_conv_h2(Model &m, const Op &o, ConvPlan &cp, bool fused_post, bool backward) {
  if (option(OPT_MI_GNINA_NO_H2) || !conv_h2_has_cfg(cp.cfg)) return;
  if (cp.has_lat && !conv_h2_has_cfg(cp.lat_cfg)) return;
  ConvArgs a = cp.a;  // geometry, tiles, bias, BatchNorm, ReLU / pool, output slice
  const int taps = o.ksize * o.ksize * o.ksize;
  const int halo = o.ksize == 3 ? 1 : 0;
  const int cin8 = cdiv(o.cin, 8);
  // 3x3x3 layers with 32-wide output tiles: conv3d_h2_kernel -- one octet per K chunk, planar double-buffered halo tile
  const bool planar = o.ksize == 3 && a.coutp != 16;
  // K chunking over octets.  The throughput tile and the latency tile share the packed weights, i.e. the chunking: sized
  // for the larger halo
  size_t HV = (size_t)(2 * a.tcx + 2 * halo) * (2 * a.tcy + 2 * halo) * (2 * a.tcz + 2 * halo);
  if (cp.has_lat)
    HV = std::max(HV, (size_t)(2 * cp.lat_tc[0] + 2 * halo) * (2 * cp.lat_tc[1] + 2 * halo) * (2 * cp.lat_tc[2] + 2 * halo));
  size_t budget = 52 * 1024;
  if (const char *ev = option(OPT_MI_GNINA_H2_LDS_KB))
    if (atoi(ev) > 0) budget = (size_t)atoi(ev) * 1024;
  // the kernel's staging registers (conv3d_h2.hip: VPT halo voxels per thread x NQ channel quads per chunk -- 3 x 4 under a
  // 3x3x3 conv's halo, 1 x 12 for a 1x1x1 conv)
  int vpt = 3, max_c = 2;
  if (a.coutp != 16 && o.ksize == 1) vpt = 1, max_c = 6;
  if (planar) max_c = 1;
  {
    int wm, wn, tm, tn;
    conv_cfg_shape(cp.cfg, &wm, &wn, &tm, &tn);
    if (HV > (size_t)(planar ? 4 : vpt) * 64 * wm * wn) return;  // (HV covers the latency tile too; conv3d_h2_kernel checks its own limits at launch)
    if (a.coutp != 16 && o.ksize != 1 && tn >= 3) return;  // tile shapes compiled for the 1x1x1 bottlenecks only
    if (a.coutp != 16 && o.ksize == 1 && a.mt_x) return;
    if (planar) {
      if (wm * wn != 4) return;
      int cfg_here = cp.cfg;
      // the 6^3 layers (1 x 4 waves x 7 M-tiles, one workgroup per pose): on the split-fp16 kernel the SAME tile as 4 x 1
      // waves x 2 M-tiles of ONE 32-channel group (the other groups along blockIdx.y: the small input tile is staged once
      // per group) -- every wave of a workgroup then wants the same B operands, which go through LDS once per workgroup
      // (and per two poses) instead of through L1 / L2 once per wave, and the launch has four times the workgroups:
      // 0.35 -> 0.27 ms for 64 -> 128 at 6^3.  (The 12^3 layers, 2 x 2 waves x 3 M-tiles of a 2 x 2 x 6-cell tile, lose
      // with it -- 0.49 -> 0.58 ms: six M-tiles leave two of the eight slots idle and the tile is staged twice;
      // MI_GNINA_H2_WN1=all tries them.)  Not with a fused 1x1x1 conv behind it (all its input channels must sit in one
      // workgroup).
      const char *wn1 = option(OPT_MI_GNINA_H2_WN1);
      const bool wn1_all = wn1 && !strcmp(wn1, "all"), wn1_off = wn1 && !strcmp(wn1, "0");
      if (!fused_post && !wn1_off && (cp.cfg == CONV_CFG_1x4_7x1 || (wn1_all && cp.cfg == CONV_CFG_2x2_3x1))) {
        const int n_mt_raster = cdiv(a.tcx * a.tcy * a.tcz, 4);
        if (n_mt_raster <= 8) {
          cp.h2_cfg = cfg_here = CONV_CFG_4x1_2x1;
          wm = 4, wn = 1, tm = 2;
        }
      }
      const int tc[3] = {a.tcx, a.tcy, a.tcz};
      int mt, py, px;
      h2_choose_layout(tc, wm * tm, conv_h2_mt_mask(cfg_here), mt, py, px);
      a.mt_x = mt, a.h2_pad_y = py, a.h2_pad_x = px;
      if (cp.has_lat) {
        int lwm, lwn, ltm, ltn;
        conv_cfg_shape(cp.lat_cfg, &lwm, &lwn, &ltm, &ltn);
        if (lwm * lwn != 4) return;
        h2_choose_layout(cp.lat_tc, lwm * ltm, conv_h2_mt_mask(cp.lat_cfg), mt, py, px);
        cp.h2_lat_mt = mt;
        cp.h2_lat_pad[0] = py, cp.h2_lat_pad[1] = px;
      }
    }
  }
  int best = 1;
  for (int c = 1; c <= cin8 && c <= max_c; c++)
    if (HV * (16 * c + 8) * 2 + (size_t)(taps * c + 8) * 4 + HV * 4 <= budget) best = c;
  if (planar) best = 1;
  const int nchunks = cdiv(cin8, best);
  best = cdiv(cin8, nchunks);  // equal chunks (the last one may still be an octet short)
  a.cc4 = best;
  a.ccs = 16 * best + 8;  // odd number of 16-byte slots per voxel: neighbouring voxels land on different slots
  a.nchunks = nchunks;
  a.cin4 = cdiv(o.cin, 4);
  // octets per MFMA step: two for 32x32x16 (rows [chunk][pair][2][coutp]), four for the 16-wide 16x16x32 ([chunk][step][4][16])
  const bool n16 = a.coutp == 16;
  const int kstep = n16 ? 4 : 2;
  const int Pmax = (taps * best + kstep - 1) / kstep;
  const float *w = m.d.data.data() + o.w_off;  // canonical [tap][cin][cout]
  std::vector<float> wT;
  if (backward) {  // W'[tap][co][ci] = W[taps-1-tap][ci][co]
    const int fci = o.cout, fco = o.cin;  // forward cin / cout
    wT.resize((size_t)taps * o.cin * o.cout);
    for (int t = 0; t < taps; t++)
      for (int ci = 0; ci < fci; ci++)
        for (int co = 0; co < fco; co++)
          wT[((size_t)t * fco + co) * fci + ci] = w[((size_t)(taps - 1 - t) * fci + ci) * fco + co];
    w = wT.data();
  }
  const float sw = h2_weight_scale(w, (size_t)taps * o.cin * o.cout);
  std::vector<unsigned short> wp((size_t)nchunks * Pmax * kstep * a.coutp * 16, 0);
  for (int ch = 0; ch < nchunks; ch++)
    for (int c8 = 0; c8 < best && ch * best + c8 < cin8; c8++)
      for (int tap = 0; tap < taps; tap++) {
        const int q = c8 * taps + tap, pr = q / kstep, kh = q % kstep;
        for (int j = 0; j < 8; j++) {
          const int c = (ch * best + c8) * 8 + j;
          if (c >= o.cin) continue;
          for (int n = 0; n < o.cout; n++) {
            const float v = w[((size_t)tap * o.cin + c) * o.cout + n] * sw;
            const unsigned short hi = host_f2h(v), lo = host_f2h(v - host_h2f(hi));
            if (planar) {
              // conv3d_h2_kernel: [chunk][step][h | l][half-wave][cout][8] -- the 64 lanes of a B-operand load read 1 KB of
              // consecutive bytes (eight full cache lines; interleaved h | l rows cost sixteen half-used ones per load)
              const size_t row = (((size_t)ch * Pmax + pr) * 2 * kstep + kh) * a.coutp + n;
              wp[row * 8 + j] = hi;
              wp[(row + (size_t)kstep * a.coutp) * 8 + j] = lo;
              continue;
            }
            const size_t idx = ((((size_t)ch * Pmax + pr) * kstep + kh) * a.coutp + n) * 16;
            wp[idx + j] = hi;
            wp[idx + 8 + j] = lo;
          }
        }
      }
  if (o.bn_scale_off >= 0) {  // the BatchNorm itself (the fp32 plan of a Dense-block layer may carry the zero-shift variant)
    std::vector<float> sc(a.cin4 * 4, 0.f), sh(a.cin4 * 4, 0.f);
    std::copy(m.d.data.begin() + o.bn_scale_off, m.d.data.begin() + o.bn_scale_off + o.cin, sc.begin());
    std::copy(m.d.data.begin() + o.bn_shift_off, m.d.data.begin() + o.bn_shift_off + o.cin, sh.begin());
    a.bn_scale = push_dev(m, sc);
    a.bn_shift = push_dev(m, sh);
  }
  std::vector<float> wpf((wp.size() + 1) / 2, 0.f);
  memcpy(wpf.data(), wp.data(), wp.size() * sizeof(unsigned short));
  a.wp = push_dev(m, wpf);
  a.h2_unscale = 1.f / sw;
  a.sparse = 0;
  a.korder = 0;
  a.bias_tab = nullptr;
  a.mfma_count = nullptr;
  MIG_CHECK(conv_h2_lds_bytes(a) <= 160 * 1024, 2, "split-fp16 conv tile exceeds LDS");
  cp.h2 = a;
  cp.has_h2 = true;
  cp.h2_planar = planar;
}

// ---- a Dense-block layer (BatchNorm -> 3x3x3 conv, c_in -> 16 -> ReLU, appended to the concat buffer it reads) on
// conv3d_h2_d16_kernel: split-format concat buffer in and out, BatchNorm folded.  `cp.a` (the fp32 plan) supplies S, bias ----
static void plan_conv_d16(Model &m, const Op &o, ConvPlan &cp) {
  if (o.ksize != 3 || o.cout != 16 || o.src != o.dst || o.cin % 8 || o.dst_c0 % 8 || cp.a.pool != 0) return;
  const int S = m.d.bufs[o.src].S, cells = S / 2;
  ConvArgs a = cp.a;
  // tiles whose A-operand reads are conflict-free without pad slots (conv3d_h2_dense.hip): 2 x 4 x 4 cells at 24^3 / 48^3
  // (sixteen M-tiles of 4 x 2 x 2 voxels: four per wave), 2 x 2 x 6 cells at 12^3 (twelve: three per wave)
  if (cells % 4 == 0) a.tcx = 2, a.tcy = 4, a.tcz = 4;
  else if (cells == 6) a.tcx = 2, a.tcy = 2, a.tcz = 6;
  else return;  // (6^3 blocks: 0.13 ms per layer, left on conv3d_h2_16_kernel and an fp32 buffer)
  if (const char *ev = option(OPT_MI_GNINA_D16_TILE))  // (experiment: "xyz" cells of the throughput tile where it divides the grid, e.g. 224)
    if (strlen(ev) == 3) {
      const int tx = ev[0] - '0', ty = ev[1] - '0', tz = ev[2] - '0';
      if (tx >= 2 && tx % 2 == 0 && ty >= 1 && tz >= 1 && cells % tx == 0 && cells % ty == 0 && cells % tz == 0 && (tx / 2) * ty * tz <= 16)
        a.tcx = tx, a.tcy = ty, a.tcz = tz;
    }
  a.ntx = cdiv(cells, a.tcx), a.nty = cdiv(cells, a.tcy), a.ntz = cdiv(cells, a.tcz);
  a.mt_x = 0;
  a.h2_pad_y = a.h2_pad_x = 0;
  {
    const int HY = 2 * a.tcy + 2, HZ = 2 * a.tcz + 2;
    const int SY = HZ, SX = HY * SY;
    if (!conv_d16_layout_conflict_free(SY, SX)) {  // (not the case for the tiles above; kept for other grids)
      bool found = false;
      for (int py = 0; py < 4 && !found; py++)
        for (int px = 0; px < 8 && !found; px++)
          if (conv_d16_layout_conflict_free(HZ + py, HY * (HZ + py) + px)) a.h2_pad_y = py, a.h2_pad_x = px, found = true;
      if (!found) return;
    }
  }
  unsigned char order[28];
  conv_d16_tap_order(a.d16_taps, order);
  const int cin8 = o.cin / 8;
  a.nchunks = cin8;
  a.cc4 = 1;
  a.cin4 = o.cin / 4;
  a.coutp = 16;
  // folded weights W'[t][c][n] = scale[c] W[t][c][n] (canonical [tap][cin][cout]); bias table per border class
  const float *w = m.d.data.data() + o.w_off;
  std::vector<float> sc(o.cin, 1.f), sh(o.cin, 0.f);
  if (o.bn_scale_off >= 0) {
    std::copy(m.d.data.begin() + o.bn_scale_off, m.d.data.begin() + o.bn_scale_off + o.cin, sc.begin());
    std::copy(m.d.data.begin() + o.bn_shift_off, m.d.data.begin() + o.bn_shift_off + o.cin, sh.begin());
  }
  std::vector<float> wf((size_t)27 * o.cin * 16);
  for (int t = 0; t < 27; t++)
    for (int c = 0; c < o.cin; c++)
      for (int n = 0; n < 16; n++) wf[((size_t)t * o.cin + c) * 16 + n] = (float)((double)sc[c] * (double)w[((size_t)t * o.cin + c) * o.cout + n]);
  const float sw = h2_weight_scale(wf.data(), wf.size());
  // packed [chunk = octet][step][h | l][lane group][cout][8 fp16]
  std::vector<unsigned short> wp((size_t)cin8 * 7 * 2 * 4 * 16 * 8, 0);
  for (int ch = 0; ch < cin8; ch++)
    for (int st = 0; st < 7; st++)
      for (int g = 0; g < 4; g++) {
        const int tap = order[4 * st + g];
        if (tap > 26) continue;  // the filler: zero rows
        for (int n = 0; n < 16; n++)
          for (int j = 0; j < 8; j++) {
            const float v = wf[((size_t)tap * o.cin + ch * 8 + j) * 16 + n] * sw;
            const unsigned short hi = host_f2h(v), lo = host_f2h(v - host_h2f(hi));
            const size_t rowh = ((((size_t)ch * 7 + st) * 2 + 0) * 4 + g) * 16 + n, rowl = ((((size_t)ch * 7 + st) * 2 + 1) * 4 + g) * 16 + n;
            wp[rowh * 8 + j] = hi;
            wp[rowl * 8 + j] = lo;
          }
      }
  std::vector<float> wpf((wp.size() + 1) / 2, 0.f);
  memcpy(wpf.data(), wp.data(), wp.size() * sizeof(unsigned short));
  a.wp = push_dev(m, wpf);
  a.h2_unscale = 1.f / sw;
  // y = sum (x sc + sh) W = sum x (sc W) + sum over the taps INSIDE the grid of sh W: one bias row per border class of the
  // output voxel (3 x 3 x 3: first / interior / last plane per axis; the zero padding is zero after the BatchNorm)
  std::vector<float> tab(27 * 16, 0.f);
  for (int cls = 0; cls < 27; cls++) {
    const int e[3] = {cls / 9, (cls / 3) % 3, cls % 3};
    for (int n = 0; n < 16; n++) {
      double acc = m.d.data[o.b_off + n];
      for (int t = 0; t < 27; t++) {
        const int d[3] = {t / 9 - 1, (t / 3) % 3 - 1, t % 3 - 1};
        bool inside = true;
        for (int ax = 0; ax < 3; ax++)
          if ((e[ax] == 0 && d[ax] < 0) || (e[ax] == 2 && d[ax] > 0)) inside = false;
        if (!inside) continue;
        for (int c = 0; c < o.cin; c++) acc += (double)sh[c] * (double)w[((size_t)t * o.cin + c) * o.cout + n];
      }
      tab[cls * 16 + n] = (float)acc;
    }
  }
  a.bias_tab = push_dev(m, tab);
  a.bn_scale = a.bn_shift = nullptr;
  a.sparse = 0;
  a.korder = 0;
  a.mfma_count = nullptr;
  a.in_split = a.out_split = 1;
  a.h2_wlds = 2;
  if (conv_h2_d16_lds_bytes(a) > 160 * 1024) return;
  cp.d16 = a;
  cp.has_d16 = true;
  // The K loop of this kernel is bound by LDS reads (an A byte feeds 16 output channels: 40 KB of ds_read_b128 per step and
  // CU against 192 cycles of MFMA with four M-tiles per wave), so a per-pose call -- 54 workgroups at 24^3, 9 at 12^3 -- is
  // faster on MORE CUs with one M-tile per wave: 2 x 2 x 2 cells, conflict-free without pads (z-row stride 6, x-plane 36 slots)
  if (cells % 2 == 0 && conv_d16_layout_conflict_free(6, 36) && !option(OPT_MI_GNINA_NO_LAT)) cp.d16_lat_tc[0] = cp.d16_lat_tc[1] = cp.d16_lat_tc[2] = 2;
}

// ---- bf16 program (conv3d_bf16.hip): octets of 8 channels, bf16 activations, fp32 accumulation ----
static unsigned short host_f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

static void plan_conv_bf16(Model &m, const Op &o, ConvPlan &cp, int pool_mode, int dst_buf, int dst_c0,
                           bool backward = false) {
  const int S = m.d.bufs[o.src].S;
  MIG_CHECK(S % 2 == 0, 2, "conv spatial size must be even");
  const int cells = S / 2;
  const int NT = cdiv(o.cout, 32);
  ConvArgs &a = cp.a;
  a = ConvArgs{};
  a.S = S;
  a.cout = o.cout;
  a.coutp = NT * 32;
  a.ksize = o.ksize;
  a.relu = o.relu;
  a.pool = pool_mode;
  a.out_c0 = dst_c0;
  if (o.ksize == 1 && NT == 3 && cells % 4 == 0) {
    cp.cfg = CONV_CFG_4x1_2x3;
    a.tcx = 2, a.tcy = 4, a.tcz = 4;
  } else if (o.ksize == 1 && NT == 5) {
    cp.cfg = CONV_CFG_4x1_1x5;
    if (cells == 6) a.tcx = 2, a.tcy = 2, a.tcz = 3;
    else a.tcx = 2, a.tcy = 2, a.tcz = 4;
  } else if (cells == 3 && NT % 4 == 0) {
    cp.cfg = CONV_CFG_1x4_7x1;
    a.tcx = a.tcy = a.tcz = 3;
  } else if (cells == 6 && NT % 2 == 0) {
    cp.cfg = CONV_CFG_2x2_3x1;
    a.tcx = 2, a.tcy = 2, a.tcz = 6;
  } else {
    cp.cfg = CONV_CFG_4x1_2x1;
    if (cells % 4 == 0) a.tcx = 2, a.tcy = 4, a.tcz = 4;
    else if (cells == 6) a.tcx = 2, a.tcy = 2, a.tcz = 6;
    else if (cells == 3) a.tcx = 3, a.tcy = 3, a.tcz = 3;
    else a.tcx = 2, a.tcy = 4, a.tcz = 4;
  }
  a.ntx = cdiv(cells, a.tcx);
  a.nty = cdiv(cells, a.tcy);
  a.ntz = cdiv(cells, a.tcz);
  const int cin8 = (cdiv(o.cin, 8) + 1) & ~1;  // octets, rounded up to even: the two k-halves of an MFMA share a tap
  const int halo = o.ksize == 3 ? 1 : 0;
  const size_t HV = (size_t)(2 * a.tcx + 2 * halo) * (2 * a.tcy + 2 * halo) * (2 * a.tcz + 2 * halo);
  const int taps = o.ksize * o.ksize * o.ksize;
  size_t budget = 36 * 1024;  // small tiles, 4 workgroups per CU: staging and MFMA phases of different workgroups overlap (measured best of 24..72 KB)
  if (const char *ev = option(OPT_MI_GNINA_BF16_LDS_KB))
    if (atoi(ev) > 0) budget = (size_t)atoi(ev) * 1024;
  int best = 2;
  for (int c = 2; c <= cin8; c += 2) {
    if (cin8 % c) continue;
    if (HV * 8 * (c | 1) * 2 + (size_t)taps * c * 4 + 64 * c + HV * 4 <= budget) best = c;
  }
  a.cc4 = best;             // octets per chunk
  a.ccs = 8 * (best | 1);   // odd octet stride: consecutive voxels land on different 16-byte LDS slots
  a.nchunks = cin8 / best;
  a.cin4 = cin8;
  const int Q = taps * a.cc4, P = (Q + 1) / 2;
  std::vector<unsigned short> wp((size_t)a.nchunks * P * 2 * a.coutp * 8, 0);
  const float *w = m.d.data.data() + o.w_off;  // canonical [tap][cin][cout]
  std::vector<float> wT;
  if (backward) {  // W'[tap][co][ci] = W[taps-1-tap][ci][co] (flipped, transposed); o is the swapped-channel twin
    const int fci = o.cout, fco = o.cin;
    wT.resize((size_t)taps * o.cin * o.cout);
    for (int t = 0; t < taps; t++)
      for (int ci = 0; ci < fci; ci++)
        for (int co = 0; co < fco; co++)
          wT[((size_t)t * fco + co) * fci + ci] = w[((size_t)(taps - 1 - t) * fci + ci) * fco + co];
    w = wT.data();
  }
  for (int ch = 0; ch < a.nchunks; ch++)
    for (int pr = 0; pr < P; pr++)
      for (int kh = 0; kh < 2; kh++) {
        const int q = 2 * pr + kh;
        if (q >= Q) continue;
        const int tap = q / a.cc4, c8 = q % a.cc4;
        for (int j = 0; j < 8; j++) {
          const int c = (ch * a.cc4 + c8) * 8 + j;
          if (c >= o.cin) continue;
          for (int n = 0; n < o.cout; n++)
            wp[((((size_t)ch * P + pr) * 2 + kh) * a.coutp + n) * 8 + j] =
                host_f2bf(w[((size_t)tap * o.cin + c) * o.cout + n]);
        }
      }
  std::vector<float> wpf((wp.size() + 1) / 2, 0.f);
  memcpy(wpf.data(), wp.data(), wp.size() * sizeof(unsigned short));
  a.wp = push_dev(m, wpf);
  std::vector<float> bias(a.coutp, 0.f);
  if (!backward) std::copy(m.d.data.begin() + o.b_off, m.d.data.begin() + o.b_off + o.cout, bias.begin());
  a.bias = push_dev(m, bias);
  if (o.bn_scale_off >= 0) {
    std::vector<float> sc(cin8 * 8, 0.f), sh(cin8 * 8, 0.f);
    std::copy(m.d.data.begin() + o.bn_scale_off, m.d.data.begin() + o.bn_scale_off + o.cin, sc.begin());
    std::copy(m.d.data.begin() + o.bn_shift_off, m.d.data.begin() + o.bn_shift_off + o.cin, sh.begin());
    a.bn_scale = push_dev(m, sc);
    a.bn_shift = push_dev(m, sh);
  }
  cp.src = o.src;
  cp.dst = dst_buf;
  cp.cin = o.cin;
  MIG_CHECK(conv_bf16_lds_bytes(a) <= 160 * 1024, 2, "bf16 conv tile exceeds LDS");
}

// The forward program on the bf16 kernels.  Tensors that stay fp32: the pooled voxel grid (written by the
// voxelizer), whatever feeds the fully connected heads, the global-max output.
static void build_bf16_program(Model &m, bool grad) {
  const ModelDesc &d = m.d;
  if (grad) MIG_CHECK(m.grad_supported, 2, m.grad_unsupported_reason);
  std::vector<char> f32(d.bufs.size(), 0);
  f32[m.input_dst] = 1;
  for (const Op &o : d.ops) {
    if (o.kind == OpKind::Fc) f32[o.src] = 1;
    if (o.kind == OpKind::GMax) f32[o.dst] = 1;
  }
  std::vector<Step> out;
  for (size_t i = 1; i < d.ops.size(); i++) {
    const Op &o = d.ops[i];
    Step st;
    st.kind = o.kind;
    if (o.kind == OpKind::Conv) {
      int pool_mode = 0, dst = o.dst;
      if (i + 1 < d.ops.size() && d.ops[i + 1].kind == OpKind::Pool && d.ops[i + 1].src == o.dst && o.src != o.dst &&
          o.dst_c0 == 0 && o.cout == d.bufs[o.dst].C) {
        bool used_elsewhere = false;
        for (size_t j = i + 2; j < d.ops.size(); j++)
          if (d.ops[j].src == o.dst) used_elsewhere = true;
        if (!used_elsewhere) {
          // (gradient program: an avg pool would have to stay un-fused, which the bf16 path does not do)
          MIG_CHECK(!(grad && d.ops[i + 1].pool_mode == 2), 2, "bf16 gradient: average-pooling networks run in fp32");
          pool_mode = d.ops[i + 1].pool_mode;
          dst = d.ops[i + 1].dst;
          i++;
        }
      }
      MIG_CHECK(f32[o.src] || m.buf_cp[o.src] % 8 == 0, 2, "bf16 path: channel stride must be a multiple of 8");
      MIG_CHECK(f32[dst] || (m.buf_cp[dst] % 8 == 0 && o.dst_c0 % 8 == 0), 2, "bf16 path: unaligned output slice");
      plan_conv_bf16(m, o, st.conv, pool_mode, dst, o.dst_c0);
      st.conv.a.in_f32 = f32[o.src];
      st.conv.a.out_f32 = f32[dst];
      st.has_bn = o.bn_scale_off >= 0;
      if (grad) {
        // transposed twin: reads the fp32 gradient of this conv's output slice (masked by the bf16 / fp32
        // activation), writes / accumulates the fp32 gradient of its input
        MIG_CHECK(o.cout % 8 == 0, 2, "bf16 gradient: conv output channels must be a multiple of 8");
        plan_conv_bf16(m, make_bwd_op(o), st.bwd, 0, o.src, 0, true);
        st.has_bwd = true;
        st.bwd.a.in_f32 = 1;
        st.bwd.a.out_f32 = 1;
        st.bwd.a.act_f32 = f32[dst];
        if (st.has_bn) {
          std::vector<float> sc(st.bwd.a.coutp, 0.f);
          std::copy(d.data.begin() + o.bn_scale_off, d.data.begin() + o.bn_scale_off + o.cin, sc.begin());
          st.bwd.a.out_scale = push_dev(m, sc);
        }
        st.bwd.a.accumulate = o.src == o.dst ? 1 : 0;
      }
    } else if (o.kind == OpKind::GMax) {
      st.src = o.src;
      st.dst = o.dst;
      st.C = d.bufs[o.src].C;
      st.src_bf16 = !f32[o.src];
    } else if (o.kind == OpKind::Fc) {
      st.src = o.src;
      st.w_off = o.w_off;
      st.b_off = o.b_off;
      st.n_in = o.n_in;
    } else {
      throw Error(2, "bf16 path: stand-alone pooling layers are not supported");
    }
    out.push_back(st);
  }
  (grad ? m.hgsteps : m.hsteps) = std::move(out);
}

static Model *build_model(ModelDesc &&desc) {
  std::unique_ptr<Model> m(new Model());
  m->d = std::move(desc);
  ModelDesc &d = m->d;
  m->N = d.grid_points();
  m->C = d.n_channels();
  m->Cp = round_up(m->C, 4);
  // density constants exactly as the oracle computes them (fp32)
  {
    float e = expf(-2.0f);
    m->qa = e * 4.0f;
    m->qb = -e * 12.0f;
    m->qc = e * 9.0f;
    for (int t = 0; t < kNumSminaTypes; t++) m->dens[t] = density_consts(smina_xs_radius(t), d.radius_scaling);
  }
  m->buf_cp.assign(d.bufs.size(), 0);
  if (d.ops[0].kind == OpKind::Overlap) {
    // no layer program: the full-resolution two-channel grid [B][2][N^3] (reference layout) is the input
    m->overlap = true;
    m->input_pool = 0;
    m->input_dst = 0;
    m->buf_cp[0] = 2;
    m->grad_supported = true;
    return m.release();
  }
  // op 0 = pool of the voxel grid -> fused into the voxelizer
  m->input_pool = d.ops[0].pool_mode;
  m->input_dst = d.ops[0].dst;
  m->buf_cp[m->input_dst] = round_up(d.bufs[m->input_dst].C, 4);
  m->op_h2.assign(d.ops.size(), 0);
  // f32only: no split-fp16 plans, and "conv3 -> ReLU -> conv1" pairs fused into one fp32 kernel (the intermediate tensor
  // never reaches HBM).  The default forward program runs such a pair as two split-fp16 kernels instead: at the f16 MFMA
  // rate the round trip of the intermediate tensor costs less than the fused fp32 kernel's MFMAs.
  auto build_steps = [&](bool grad, bool f32only) {
    const bool no_h2 = f32only || option(OPT_MI_GNINA_NO_H2);
    std::vector<Step> out;
    for (size_t i = 1; i < d.ops.size(); i++) {
      const Op &o = d.ops[i];
      Step st;
      st.kind = o.kind;
      if (o.kind == OpKind::Conv) {
        // fuse "conv -> pool" when the pool directly consumes this conv's private output
        // (gradient program: only max pools are fused -- their arg-max is saved; avg pools keep the
        // pre-pool activation, which the ReLU backward needs)
        int pool_mode = 0, dst = o.dst, dst_c0 = o.dst_c0;
        const size_t conv_op_index = i;
        // fuse "conv3 -> ReLU -> conv1 (same width) -> ..." (Default2018's unit pairs): the 1x1x1 conv runs as a
        // second MFMA pass on the LDS-transposed tile inside the first conv's kernel (forward program only --
        // the gradient program needs the intermediate activation)
        const Op *post = nullptr;
        // (the default forward program fuses too -- conv3d_h2.hip carries the pair on the f16 pipe -- unless
        // MI_GNINA_H2_NO_FUSE1X1 asks for two launches)
        if (!grad && (no_h2 || !option(OPT_MI_GNINA_H2_NO_FUSE1X1)) && o.ksize == 3 && i + 1 < d.ops.size() &&
            d.ops[i + 1].kind == OpKind::Conv) {
          const Op &o2 = d.ops[i + 1];
          bool used_elsewhere = false;
          for (size_t j = i + 2; j < d.ops.size(); j++)
            if (d.ops[j].src == o.dst || (d.ops[j].kind == OpKind::Conv && d.ops[j].dst == o.dst)) used_elsewhere = true;
          if (o2.ksize == 1 && o2.src == o.dst && o2.dst != o.dst && o.src != o.dst && o.dst_c0 == 0 && o2.dst_c0 == 0 &&
              o.cout == d.bufs[o.dst].C && o2.cin == o.cout && o2.cout == o.cout && o.cout % 32 == 0 && o.cout <= 64 &&
              o2.bn_scale_off < 0 && !used_elsewhere && !option(OPT_MI_GNINA_NO_FUSE1X1)) {
            post = &o2;
            dst = o2.dst;
            dst_c0 = 0;
            i++;  // swallow the 1x1 conv
          }
        }
        const Op &last = post ? *post : o;  // the op whose output a following pool would consume
        if (i + 1 < d.ops.size() && d.ops[i + 1].kind == OpKind::Pool && d.ops[i + 1].src == last.dst &&
            last.src != last.dst && last.dst_c0 == 0 && last.cout == d.bufs[last.dst].C &&
            !(grad && d.ops[i + 1].pool_mode == 2)) {
          bool used_elsewhere = false;
          for (size_t j = i + 2; j < d.ops.size(); j++)
            if (d.ops[j].src == last.dst) used_elsewhere = true;
          if (!used_elsewhere) {
            pool_mode = d.ops[i + 1].pool_mode;
            dst = d.ops[i + 1].dst;
            i++;  // swallow the pool
          }
        }
        plan_conv(*m, o, st.conv, pool_mode, dst, dst_c0);
        // the split-fp16 twin (not with a fused 1x1 conv).  The gradient
        // program's forward pass takes it for exactly the layers the forward program does: a pose scores the same bits
        // with and without its gradient (eval vs eval_deriv energies are compared inside the search)
        if (!no_h2 && (!grad || m->op_h2[conv_op_index])) plan_conv_h2(*m, o, st.conv, post != nullptr);
        // Dense family on split-format tensors (forward program only; which of the two plans runs follows the buffers' format)
        if (!no_h2 && !grad && !post && !option(OPT_MI_GNINA_NO_DENSE_SPLIT)) {
          plan_conv_d16(*m, o, st.conv);
          const ConvArgs &h = st.conv.h2;
          st.conv.has_k1s = st.conv.has_h2 && !st.conv.h2_planar && o.ksize == 1 && o.bn_scale_off < 0 && o.cin % 8 == 0 &&
                            (st.conv.cfg == CONV_CFG_4x1_1x3 || st.conv.cfg == CONV_CFG_4x1_1x5) && h.coutp == (st.conv.cfg == CONV_CFG_4x1_1x3 ? 96 : 160) &&
                            h.tcx * h.tcy * h.tcz <= 16 && h.cc4 * h.nchunks == o.cin / 8 && (size_t)8 * h.tcx * h.tcy * h.tcz * (2 * h.cc4 + 1) <= 7 * 256;
        }
        if (post && st.conv.has_h2) {
          // the 1x1x1 conv's own split-fp16 plan supplies the packed weights of the fused second pass: its K chunks must
          // be consecutive whole octet pairs ([chunk][pairs][2][coutp][h | l] is then one [pair][2][coutp][h | l] array)
          int wm_, wn_, tm_, tn_;
          conv_cfg_shape(st.conv.cfg, &wm_, &wn_, &tm_, &tn_);
          ConvPlan p2;
          plan_conv(*m, *post, p2, 0, post->dst, 0);
          plan_conv_h2(*m, *post, p2);
          const bool ok = p2.has_h2 && tm_ <= 3 && tn_ == 1 && st.conv.a.coutp / 32 <= wn_ && p2.h2.cc4 % 2 == 0 &&
                          p2.h2.cc4 * p2.h2.nchunks * 8 == st.conv.a.coutp;
          if (ok) {
            st.conv.h2.post_w = p2.h2.wp;
            st.conv.h2.post_cc4 = p2.h2.cc4 * p2.h2.nchunks;  // octets
            st.conv.h2.post_bias = p2.h2.bias;
            st.conv.h2.post_relu = post->relu;
            st.conv.h2.h2_post_unscale = p2.h2.h2_unscale;
            st.conv.h2.post_rows = wm_ * tm_ * 32;
            if (conv_h2_lds_bytes(st.conv.h2) > 160 * 1024) st.conv.has_h2 = false;
          } else {
            st.conv.has_h2 = false;  // (this pair stays on the fused fp32 kernel)
          }
        }
        if (!grad && !f32only) {
          m->op_h2[conv_op_index] = st.conv.has_h2 ? 1 : 0;
          if (post) m->op_h2[conv_op_index + 1] = st.conv.has_h2 ? 1 : 0;
        }
        if (post) {
          {
            int wm_, wn_, tm_, tn_;
            conv_cfg_shape(st.conv.cfg, &wm_, &wn_, &tm_, &tn_);
            MIG_CHECK(tm_ <= 3 && tn_ == 1, 2, "fused 1x1 conv planned on a kernel shape that does not carry it");
          }
          ConvPlan p2;
          plan_conv(*m, *post, p2, 0, post->dst, 0);  // for its packed weights [pair][2][coutp][4]
          MIG_CHECK(p2.a.cc4 % 2 == 0 && p2.a.cc4 * p2.a.nchunks * 4 == st.conv.a.coutp, 2,
                    "fused 1x1 conv: its K chunks must hold an even number of channel quads and cover coutp exactly");
          st.conv.a.post_w = p2.a.wp;
          st.conv.a.post_cc4 = p2.a.cc4;
          st.conv.a.post_wrows = p2.a.wrows;
          st.conv.a.post_bias = p2.a.bias;
          st.conv.a.post_relu = post->relu;
          st.post_cout = post->cout;
          int wm_, wn_, tm_, tn_;
          conv_cfg_shape(st.conv.cfg, &wm_, &wn_, &tm_, &tn_);
          st.conv.a.post_rows = wm_ * tm_ * 32;
          MIG_CHECK(conv_lds_bytes(st.conv.a) <= 160 * 1024, 2, "fused conv tile exceeds LDS");
        }
        st.has_bn = o.bn_scale_off >= 0;
        if (grad && o.cout % 4 == 0) {
          plan_conv(*m, make_bwd_op(o), st.bwd, 0, o.src, 0, true);
          st.has_bwd = true;
          // transposed 3x3x3 convs on the split-fp16 kernel (run_backward decides per call)
          const bool bwd_h2 = !no_h2 && o.ksize == 3 && o.relu && pool_mode != 2 && dst_c0 % 4 == 0 && !option(OPT_MI_GNINA_NO_H2_BWD);
          auto shape_ok = [](const ConvPlan &cp) { return conv_h2_has_bwd(cp.cfg) && (!cp.has_lat || conv_h2_has_bwd(cp.lat_cfg)); };
          // 1x1x1 convs behind a fused max pool (Dense transitions) on conv3d_h2_k1_kernel's gradient-pass variant: all
          // output channels in one workgroup, as their forward twin
          const bool bwd_h2_k1 = !no_h2 && o.ksize == 1 && o.relu && pool_mode == 1 && dst_c0 % 4 == 0 && o.src != o.dst &&
                                 !option(OPT_MI_GNINA_NO_H2_BWD) && !option(OPT_MI_GNINA_NO_H2_BWD_K1);
          if (bwd_h2_k1) {
            plan_conv(*m, make_bwd_op(o), st.bwd_h2, 0, o.src, 0, true, false, false, true);
            if (conv_h2_has_bwd_k1(st.bwd_h2.cfg) && (!st.bwd_h2.has_lat || conv_h2_has_bwd_k1(st.bwd_h2.lat_cfg)))
              plan_conv_h2(*m, make_bwd_op(o), st.bwd_h2, false, true);
            if (st.bwd_h2.has_h2 && !st.bwd_h2.h2_planar && !option(OPT_MI_GNINA_NO_H2_BWD_DENSE)) st.bwd_h2_kind = 2;
          }
          if (bwd_h2) {
            plan_conv(*m, make_bwd_op(o), st.bwd_h2, 0, o.src, 0, true, false, true);
            if (shape_ok(st.bwd_h2)) plan_conv_h2(*m, make_bwd_op(o), st.bwd_h2, false, true);
            const bool is_private = o.bn_scale_off < 0 && o.src != o.dst && dst_c0 == 0 && o.cout == d.bufs[dst].C;
            if (st.bwd_h2.has_h2) st.bwd_h2_kind = is_private ? 1 : (pool_mode == 0 && !option(OPT_MI_GNINA_NO_H2_BWD_DENSE) ? 2 : 0);
          }
          if (st.has_bn) {  // d(BN x)/dx: the transposed conv's output is scaled per (forward-input) channel
            std::vector<float> sc(st.bwd.a.coutp, 0.f);
            std::copy(d.data.begin() + o.bn_scale_off, d.data.begin() + o.bn_scale_off + o.cin, sc.begin());
            st.bwd.a.out_scale = push_dev(*m, sc);
          }
          // a Dense-block layer reads and extends the same concat buffer: its input gradient accumulates
          st.bwd.a.accumulate = o.src == o.dst ? 1 : 0;
          const int c0 = d.recmap.n_channels;
          // (fewer 32-wide N tiles -- Default2017: 35 -> 19 channels, two tiles -> one -- or few enough channels for the
          // 16-wide kernel: Default2018 / Dense, 28 -> 14; that kernel masks by the ReLU but does not un-pool, so not behind
          // a fused max pool)
          const bool fewer_tiles = cdiv(o.cin - c0, 32) < cdiv(o.cin, 32);
          const bool n16_bwd = o.cin - c0 <= 16 && pool_mode != 1 && o.ksize == 3 && o.relu;
          if (o.src == m->input_dst && !st.has_bn && o.src != o.dst && c0 > 0 && c0 < o.cin && (fewer_tiles || n16_bwd) &&
              !option(OPT_MI_GNINA_NO_LIG_BWD)) {
            // forward weights restricted to the ligand's input channels, appended to the payload: [tap][cin - c0][cout]
            const int taps = o.ksize * o.ksize * o.ksize, csub = o.cin - c0;
            Op osub = o;
            osub.cin = csub;
            osub.w_off = (long)d.data.size();
            d.data.resize(d.data.size() + (size_t)taps * csub * o.cout);
            for (int tp = 0; tp < taps; tp++)
              for (int ci = 0; ci < csub; ci++)
                for (int co = 0; co < o.cout; co++)
                  d.data[(size_t)osub.w_off + ((size_t)tp * csub + ci) * o.cout + co] = d.data[(size_t)o.w_off + ((size_t)tp * o.cin + c0 + ci) * o.cout + co];
            plan_conv(*m, make_bwd_op(osub), st.bwd_lig, 0, o.src, c0, true, n16_bwd && !fewer_tiles);
            st.has_bwd_lig = true;
            st.lig_c0 = c0;
            if (st.bwd_h2_kind) {
              plan_conv(*m, make_bwd_op(osub), st.bwd_lig_h2, 0, o.src, c0, true, false, true);
              if (shape_ok(st.bwd_lig_h2)) plan_conv_h2(*m, make_bwd_op(osub), st.bwd_lig_h2, false, true);
              st.has_bwd_lig_h2 = st.bwd_lig_h2.has_h2;
            }
          }
        }
        m->buf_cp[dst] = round_up(d.bufs[dst].C, 4);
      } else if (o.kind == OpKind::Pool) {
        st.src = o.src;
        st.dst = o.dst;
        st.pool_mode = o.pool_mode;
        st.C = d.bufs[o.src].C;
        m->buf_cp[o.dst] = round_up(d.bufs[o.dst].C, 4);
      } else if (o.kind == OpKind::GMax) {
        st.src = o.src;
        st.dst = o.dst;
        st.C = d.bufs[o.src].C;
        m->buf_cp[o.dst] = d.bufs[o.dst].C;
      } else {
        st.src = o.src;
        st.w_off = o.w_off;
        st.b_off = o.b_off;
        st.n_in = o.n_in;
        MIG_CHECK(m->buf_cp[o.src] == d.bufs[o.src].C, 2, "fc input buffer must not be channel padded");
        MIG_CHECK(o.n_in % 4 == 0, 2, "fc input size must be a multiple of 4");
      }
      out.push_back(st);
    }
    return out;
  };
  m->steps = build_steps(false, false);
  m->steps32 = option(OPT_MI_GNINA_NO_H2) ? m->steps : build_steps(false, true);
  // tensor formats of the forward program (see Model::buf_split)
  m->Cp8 = round_up(m->C, 8);
  m->buf_split.assign(d.bufs.size(), 0);
  if (!option(OPT_MI_GNINA_NO_H2) && !option(OPT_MI_GNINA_H2_NO_SPLIT_TENSORS)) {
    // A buffer is split when every layer reading it can stage the split format and every layer writing it can produce it.
    // What a layer can write may depend on what it reads (conv3d_h2_k1s_kernel is the 1x1x1 kernel with the split epilogue
    // AND the split staging): iterate down from "everything that could be" to the fixed point.
    std::vector<int> readers(d.bufs.size(), 0);
    for (const Step &st : m->steps) {
      const int src = st.kind == OpKind::Conv ? st.conv.src : st.src;
      if (src >= 0) readers[src]++;
    }
    std::vector<char> split(d.bufs.size(), 1);
    auto src_split = [&](int id) { return id == m->input_dst ? true : (bool)split[id]; };
    for (int it = 0; it < (int)d.bufs.size() + 2; it++) {
      std::vector<char> next(d.bufs.size(), 1);
      for (size_t id = 0; id < d.bufs.size(); id++)
        if (readers[id] == 0 || m->buf_cp[id] % 8 != 0) next[id] = 0;
      for (const Step &st : m->steps) {
        if (st.kind != OpKind::Conv) {
          if (st.src >= 0) next[st.src] = 0;
          if (st.dst >= 0) next[st.dst] = 0;
          continue;
        }
        const bool k3 = st.conv.has_h2 && st.conv.h2_planar;
        const ConvArgs &h = st.conv.h2;
        const bool can_read = (k3 && !st.has_bn) || st.conv.has_d16 || st.conv.has_k1s;
        // (whole octets from channel 0: conv3d_h2_kernel's epilogue; any octet-aligned slice: the Dense kernels', which run
        // only on a split input)
        const bool can_write = (k3 && h.out_c0 == 0 && h.cout == h.coutp && h.cout % 8 == 0) ||
                               (st.conv.has_d16 && src_split(st.conv.src)) ||
                               (st.conv.has_k1s && src_split(st.conv.src) && h.out_c0 % 8 == 0 && h.cout % 8 == 0);
        if (!can_read) next[st.conv.src] = 0;
        if (!can_write) next[st.conv.dst] = 0;
      }
      for (size_t id = 0; id < d.bufs.size(); id++) next[id] = next[id] && split[id];
      if (next == split) break;
      split = next;
    }
    for (size_t id = 0; id < d.bufs.size(); id++)
      if ((int)id != m->input_dst) m->buf_split[id] = split[id];
    // the pooled voxel grid: written by the voxelizer (either format), read by the first layers
    std::vector<int> split_readers(d.bufs.size(), 0);
    for (const Step &st : m->steps)
      if (st.kind == OpKind::Conv && st.conv.has_h2 && st.conv.h2_planar && !st.has_bn) split_readers[st.conv.src]++;
    m->pooled_split_ok = readers[m->input_dst] > 0 && readers[m->input_dst] == split_readers[m->input_dst] && m->input_pool != 0;
  }
  // gradient program: conv / pool stacks (Default2017 / Default2018 families) and the Dense family
  // (BatchNorm-on-input convs growing a concat buffer, global max pool)
  m->grad_supported = !d.skip_softmax && !d.apply_logistic_loss;
  if (!m->grad_supported) m->grad_unsupported_reason = "skip_softmax / apply_logistic_loss models";
  for (const Op &o : d.ops) {
    if (o.kind == OpKind::Conv && o.cout % 4 != 0) {
      m->grad_supported = false;
      m->grad_unsupported_reason = "conv output channels not a multiple of 4";
    }
    if (o.kind == OpKind::Pool && o.pool_mode == 1 && &o != &d.ops[0]) {
      // a max pool must be fusable into the conv that feeds it
      bool fed_by_conv = false;
      for (const Op &c : d.ops)
        if (c.kind == OpKind::Conv && c.dst == o.src && c.src != c.dst) fed_by_conv = true;
      if (!fed_by_conv) {
        m->grad_supported = false;
        m->grad_unsupported_reason = "stand-alone max pooling layer";
      }
    }
  }
  if (m->grad_supported) m->gsteps = build_steps(true, false);
  m->buf_bwd_h2.assign(d.bufs.size(), 0);
  for (size_t id = 0; id < d.bufs.size(); id++) {
    int producers = 0, fit = 0;
    for (const Step &st : m->gsteps) {
      if (st.kind == OpKind::Conv && st.conv.dst == (int)id) producers++, fit += (st.has_bwd && st.bwd_h2_kind == 1) ? 1 : 0;
      else if (st.kind != OpKind::Conv && st.kind != OpKind::Fc && st.dst == (int)id) producers++;
    }
    m->buf_bwd_h2[id] = producers == 1 && fit == 1;
  }
  for (const Step &st : m->steps) {
    int s = st.kind == OpKind::Conv ? st.conv.src : st.src;
    MIG_CHECK(s >= 0 && m->buf_cp[s] > 0, 2, "layer program reads a buffer nothing produced");
  }
  m->dev_data.ensure(d.data.size());
  MIG_HIP(hipMemcpy(m->dev_data.p, d.data.data(), d.data.size() * sizeof(float), hipMemcpyHostToDevice));
  return m.release();
}

// -------------------------------------------------------------------------------------------
// Scorer
// -------------------------------------------------------------------------------------------
struct TypedReceptor {  // one per distinct (recmap, radius_scaling)
  TypeMap map;
  float radius_scaling = 1.f;
  int n = 0;
  DevBuf<AtomRec> rec;
  DevBuf<int> chan;
  std::vector<int> row;       // input row of every typed atom (same order as rec)
  std::vector<int> smt_of;    // its smina type
  // flexible rows (mi_scorer_set_flex): slot per typed atom for the gather kernel, and the typed flexible
  // atoms as an (index, constants, channel) list for the voxelizer backward
  int n_flex_typed = 0;
  DevBuf<int> flex_slot, flex_perm, flex_chan;
  DevBuf<LigConsts> flex_consts;
};

struct VoxGroup {  // models sharing one voxelization: same maps, geometry, radius scale, input pool
  int rec_idx = -1;
  int first_model = -1;
  std::vector<int> models;
};

struct Scorer {
  std::vector<Model *> models;
  hipStream_t stream = nullptr;
  int precision = 0;  // 0 = fp32 (parity path), 1 = bf16-MFMA forward (mi_scorer_set_precision)
  int h2_honly = 0;   // MI_PRECISION_FP16: scoring calls' split-fp16 kernels issue the h * h MFMA only (ConvArgs::h2_honly)
  // fp32 forward convolutions: 1 = on the split-fp16 kernels where a layer has that plan (conv3d_h2.hip), 0 = fp32 MFMA only
  // (MI_PRECISION_FP32_MFMA, or MI_GNINA_CONV_PATH=f32 in the environment)
  int conv_path = (option(OPT_MI_GNINA_CONV_PATH) && !strcmp(option(OPT_MI_GNINA_CONV_PATH), "f32")) ? 0 : 1;
  int cap = 1024;    // poses per launch of the call in flight: min(chunk, B, what the activation budget allows)
  int chunk = 1024;  // poses per launch: fewer, larger launches win (93.6k vs 87.1k poses/s at 256); 2.4 MB/pose of HBM
  bool have_receptor = false;
  int n_rec_atoms_in = 0;
  std::vector<std::unique_ptr<TypedReceptor>> receptors;
  std::vector<VoxGroup> groups;
  // per-call ligand staging
  DevBuf<float> d_lig, d_centers_in, d_centers;
  DevBuf<int> d_lig_perm, d_lig_chan, d_cand_chan, d_cand_n, d_pose_rows, d_pose_nlig;
  DevBuf<LigConsts> d_lig_consts;
  DevBuf<unsigned char> d_lig_typed;
  DevBuf<AtomRec> d_cand, d_cand2;
  DevBuf<int> d_cand_chan2, d_cand_n2;
  // Small calls of an ensemble (gnina scores ONE pose per DLScorer::score call, torch_model.cpp:179; the default ensemble is
  // three models): every model's layer program runs on its own stream ("lane") off the voxelization on the main stream, with
  // its own set of activation buffers (act_lane) -- the launches of a B = 1 program are latency, not throughput, and three
  // independent chains of ~20 dependent kernels overlap almost entirely.  Results are the bits of the serial order.
  std::vector<hipStream_t> lane_streams;  // (the device's shared set: device_lane_stream)
  // Which of the device's eight lane streams the models take: model mi runs on stream (lane_offset + mi) % 8.  The hardware
  // queue -- and with it the command-processor pipe -- a stream sits on is the runtime's choice from what existed when the
  // stream was created, and an ensemble call of ~60 dependent launches is 0.59 ms on one choice of three streams and 0.95-1.26
  // ms on others (round 6, tools/experiments: MI_GNINA_LANE_OFFSET=0..5: 983 / 948 / 588 / 643 / 1,264 / 1,001 us).  So a scorer
  // TRIES the eight offsets on its first synchronous lane calls (kLaneTunePer calls each, the fastest of the later ones
  // counts) and keeps the best.
  int lane_offset = 0, lane_streams_offset = -1;
  int lane_tune_calls = 0, lane_tune_best = 0, lane_tune_B = 0;
  double lane_tune_best_us = 1e30, lane_tune_cur_us = 1e30;
  bool last_call_lanes = false;
  std::vector<hipEvent_t> lane_done;
  std::vector<hipEvent_t> lane_start;   // per voxelization group
  int act_lane = 0;                     // run_program: activation buffer set in use (0 = the shared set)
  bool alone_on_device = true;          // no other scorer has scored on the device lately (scorer_alone_on_device)
  size_t centers_stride = 0;            // lanes: voxel group 1 writes its grid centres behind group 0's (voxelize_chunk)
  // gradient calls on lanes (score_batch_grad_once): the pooled-grid slot of the group whose model run_program / run_backward
  // is working on, and the per-model gradient accumulators the lanes write before they are summed in model order
  size_t grad_pooled_slot = 0;
  DevBuf<float> d_lig_grad_m, d_flex_grad_m;
  std::vector<int> last_lane;           // per model: the set its last forward program wrote (mi_debug_read_activation)
  // mi_debug_vox_stress: the quiet run's pooled grid, the mismatch log, the trap ring of a -DMI_VOX_TRAP build
  DevBuf<unsigned> d_dbg_ref, d_dbg_trap;
  DevBuf<int> d_dbg_log;
  unsigned *dbg_trap = nullptr;         // handed to voxelize_tiles (VoxArgs::trap) while a stress run is on
  int dbg_cap = 0, dbg_nslab = 0;       // geometry of the last voxelize_chunk's candidate lists (mi_debug_read_candidates)
  int device = 0;                       // the HIP device the scorer was created on
  // activations: one set of buffers sized for `chunk` poses, shared by all models (max size per id)
  std::vector<std::unique_ptr<DevBuf<float>>> act;
  std::vector<std::unique_ptr<DevBuf<float>>> gact;            // gradients w.r.t. the activation buffers
  std::vector<std::unique_ptr<DevBuf<unsigned char>>> argm;    // arg-max of fused max pools
  DevBuf<float> d_raw3, d_lig_grad, d_ave;
  DevBuf<unsigned> d_probe;             // zero-cell counters of launch_zero_cell_probe
  // split-fp16 range flag (ConvArgs::h2_overflow): cleared when a call starts, raised by any split-fp16 kernel (or the
  // voxelizer writing a split tensor) that meets an activation beyond +-65504 or a NaN; a call that finds it raised
  // recomputes its scores on the fp32-MFMA kernels (score_batch / score_batch_grad)
  DevBuf<unsigned char> d_occ[2];       // occupancy bytes of the split-format pooled grid (VoxArgs::occ), per buffer set
  DevBuf<unsigned> d_ovf;
  DevBuf<unsigned> d_gamax;  // gradient pass: [buffer][cap] per-pose max |g| bits of the buffers in Model::buf_bwd_h2
  unsigned *h_ovf = nullptr;            // pinned copy
  bool ovf_clean = false;               // the device flag is known to be zero (cleared by the last call's reduction kernel)
  int h2_fallbacks = 0;                 // calls recomputed because of it (mi_scorer_h2_fallbacks)
  bool ovf_pending = false;             // device-output calls in flight whose flag mi_scorer_synchronize still has to read
  // setup_ligand cache, one per voxel group: the same ligand is scored call after call (poses of one docking run), and an
  // ensemble with two groups (gnina's default: the Dense pair + a Default2018) would otherwise re-upload four arrays and
  // synchronise twice per call (~90 us of a B = 1 call)
  struct LigDev {
    DevBuf<int> perm, chan;
    DevBuf<LigConsts> consts;
    DevBuf<unsigned char> typed;
    std::vector<int32_t> smt;
    int n = 0;
    bool valid = false;
  };
  std::vector<std::unique_ptr<LigDev>> lig_dev;
  std::vector<int> flex_rows;           // receptor rows with per-pose coordinates
  DevBuf<float> d_flex, d_flex_grad;    // [B][n_flex][3]
  const float *cur_flex = nullptr;      // device flex coordinates of the call in flight (or nullptr)
  // per-pose rotations of the next scoring call (mi_scorer_set_rotations; TorchModel::forward's `rotate`)
  std::vector<float> next_rot;          // host [B][4], consumed by one call
  DevBuf<float> d_rot;
  const float *cur_rot = nullptr;
  // outputs per model [n_models][B] and reduced
  DevBuf<float> d_pose_m, d_aff_m, d_loss_m, d_pose, d_aff, d_loss, d_var, d_out4;
  float *h_out4 = nullptr;  // pinned staging of the four output arrays
  float *h_lig_pin = nullptr;  // pinned block a small synchronous scoring call hands its poses over in (read by gather_pose_atoms)
  size_t h_out4_n = 0;
  int last_B = 0;
  // per-kernel profiling (mi_scorer_enable_profile): HIP events around every launch on `stream`
  struct ProfRec {
    std::string name;
    double flops, bytes;
    int poses;
    hipEvent_t e0, e1;
    int cnt_slot = -1;  // index of this launch's block of executed-MFMA counters in d_mfma_cnt (zero-skipping convs)
  };
  DevBuf<unsigned long long> d_mfma_cnt;  // [kProfCntLaunches][kMfmaCountSlots], profile mode only
  int cnt_used = 0;
  bool profile = false;
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  std::string prof_json;
  bool timing = false;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  float last_ms[3] = {0, 0, 0};
  ~Scorer() {
    for (auto *m : models)
      if (m && --m->refs == 0) delete m;
    for (auto &e : ev)
      if (e) (void)hipEventDestroy(e);
    for (auto &r : prof) {
      (void)hipEventDestroy(r.e0);
      (void)hipEventDestroy(r.e1);
    }
    for (auto &e : ev_pool) (void)hipEventDestroy(e);
    for (auto &e : lane_done)
      if (e) (void)hipEventDestroy(e);
    for (auto &e : lane_start)
      if (e) (void)hipEventDestroy(e);
    if (h_out4) (void)hipHostFree(h_out4);
    if (h_lig_pin) (void)hipHostFree(h_lig_pin);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

static hipEvent_t prof_event(Scorer &s) {
  hipEvent_t e;
  if (!s.ev_pool.empty()) {
    e = s.ev_pool.back();
    s.ev_pool.pop_back();
  } else {
    MIG_HIP(hipEventCreate(&e));
  }
  return e;
}

// RAII: records an event pair around one kernel launch when profiling is on
struct ProfScope {
  Scorer &s;
  bool on;
  Scorer::ProfRec r;
  hipStream_t st;
  ProfScope(Scorer &sc, const std::string &name, double flops, double bytes, int poses, hipStream_t stream = nullptr)
      : s(sc), on(sc.profile), st(stream ? stream : sc.stream) {
    if (!on) return;
    r.name = name;
    r.flops = flops;
    r.bytes = bytes;
    r.poses = poses;
    r.e0 = prof_event(s);
    r.e1 = prof_event(s);
    (void)hipEventRecord(r.e0, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.e1, st);
    s.prof.push_back(r);
  }
};

constexpr int kProfCntLaunches = 256;  // launches with an executed-MFMA counter block per profile drain

// a zeroed counter block for the launch `ps` brackets (nullptr when profiling is off or the blocks are used up)
static unsigned long long *prof_counter(Scorer &s, ProfScope &ps) {
  if (!ps.on || s.cnt_used >= kProfCntLaunches) return nullptr;
  s.d_mfma_cnt.ensure((size_t)kProfCntLaunches * kMfmaCountSlots);
  unsigned long long *p = s.d_mfma_cnt.p + (size_t)s.cnt_used * kMfmaCountSlots;
  (void)hipMemsetAsync(p, 0, kMfmaCountSlots * sizeof(unsigned long long), ps.st);
  ps.r.cnt_slot = s.cnt_used++;
  return p;
}

constexpr size_t kPooledSlot = 0;  // buffer id 0 (the full grid) is never materialised, reuse its slot
constexpr size_t kPooledSlot2 = 4096;  // pooled grid of an ensemble's second voxel group when its models run on lanes
constexpr size_t kGamaxLaneSlots = 256;  // gradient-maximum slots (Scorer::d_gamax) per lane of a gradient call on lanes
constexpr size_t kLaneSlots = 64;      // activation buffer ids per lane (Scorer::act_lane): set l owns slots [64 l, 64 l + 64)

// Poses per launch for a call on B poses: the user's chunk, clipped to B and to an activation-memory budget
// (96 GB of the 288 GB by default, MI_GNINA_ACT_GB overrides) -- a 96^3 Dense pose keeps ~110 MB of
// activations (x2.25 with gradients), a 48^3 Default2017 pose 2.4 MB.
// channel stride the pooled voxel grid's buffer is sized for: the fp32 layout (buf_cp) or the split layout (whole octets)
static int pooled_stride(const Model *m) { return std::max(m->buf_cp[m->input_dst], m->Cp8); }

static void set_call_capacity(Scorer &s, int B, bool grad) {
  double per_pose = 0;
  size_t max_bufs = 0;
  for (Model *m : s.models) max_bufs = std::max(max_bufs, m->d.bufs.size());
  for (size_t id = 0; id < max_bufs; id++) {
    double worst = 0;
    for (Model *m : s.models)
      if (id < m->d.bufs.size() && m->buf_cp[id] > 0) {
        const BufDecl &bd = m->d.bufs[id];
        worst = std::max(worst, (double)bd.S * bd.S * bd.S * ((int)id == m->input_dst ? pooled_stride(m) : m->buf_cp[id]) * 4.0);
      }
    per_pose += worst;
  }
  if (grad) per_pose *= 2.25;  // gradient buffers + arg-max bytes
  double budget_gb = 96.0;
  if (const char *ev = option(OPT_MI_GNINA_ACT_GB))
    if (atof(ev) > 0) budget_gb = atof(ev);
  const double fit = budget_gb * 1073741824.0 / std::max(per_pose, 1.0);
  int cap = std::min(s.chunk, std::max(B, 1));
  if ((double)cap > fit) cap = std::max(1, (int)fit);
  s.cap = cap;
}

static float *act_buf(Scorer &s, size_t id, size_t floats) {
  if (s.act.size() <= id) s.act.resize(id + 1);
  if (!s.act[id]) s.act[id].reset(new DevBuf<float>());
  s.act[id]->ensure(floats);
  return s.act[id]->p;
}

static float *gact_buf(Scorer &s, size_t id, size_t floats) {
  if (s.gact.size() <= id) s.gact.resize(id + 1);
  if (!s.gact[id]) s.gact[id].reset(new DevBuf<float>());
  s.gact[id]->ensure(floats);
  return s.gact[id]->p;
}

static unsigned char *argm_buf(Scorer &s, size_t id, size_t n) {
  if (s.argm.size() <= id) s.argm.resize(id + 1);
  if (!s.argm[id]) s.argm[id].reset(new DevBuf<unsigned char>());
  s.argm[id]->ensure(n);
  return s.argm[id]->p;
}

static void build_groups(Scorer &s) {
  s.groups.clear();
  for (size_t mi = 0; mi < s.models.size(); mi++) {
    Model *m = s.models[mi];
    bool placed = false;
    for (auto &g : s.groups) {
      Model *f = s.models[g.first_model];
      if (f->d.recmap == m->d.recmap && f->d.ligmap == m->d.ligmap && f->d.resolution == m->d.resolution &&
          f->d.dimension == m->d.dimension && f->d.radius_scaling == m->d.radius_scaling &&
          f->input_pool == m->input_pool && f->Cp == m->Cp) {
        g.models.push_back((int)mi);
        placed = true;
        break;
      }
    }
    if (!placed) {
      VoxGroup g;
      g.first_model = (int)mi;
      g.models.push_back((int)mi);
      s.groups.push_back(g);
    }
  }
}

static void set_receptor(Scorer &s, const float *xyz, const int32_t *smt, int n) {
  s.receptors.clear();
  for (auto &g : s.groups) {
    Model *m = s.models[g.first_model];
    int found = -1;
    for (size_t r = 0; r < s.receptors.size(); r++)
      if (s.receptors[r]->map == m->d.recmap && s.receptors[r]->radius_scaling == m->d.radius_scaling) found = (int)r;
    if (found < 0) {
      std::unique_ptr<TypedReceptor> tr(new TypedReceptor());
      tr->map = m->d.recmap;
      tr->radius_scaling = m->d.radius_scaling;
      // type (make_coordset, torch_model.cpp:120-142), drop untyped atoms, stable sort by channel
      std::vector<int> idx;
      for (int i = 0; i < n; i++) {
        int t = smt[i];
        MIG_CHECK(t >= 0 && t < kNumSminaTypes, 1, "receptor smina type out of range");
        if (tr->map.chan_of_smt[t] >= 0) idx.push_back(i);
      }
      std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
        return tr->map.chan_of_smt[smt[a]] < tr->map.chan_of_smt[smt[b]];
      });
      std::vector<AtomRec> recs(idx.size());
      std::vector<int> chans(idx.size());
      for (size_t k = 0; k < idx.size(); k++) {
        int i = idx[k];
        const DensityConsts &dc = m->dens[smt[i]];
        recs[k] = AtomRec{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], dc.ar, dc.t2, dc.g2, dc.kexp, dc.inv_ar};
        chans[k] = tr->map.chan_of_smt[smt[i]];
      }
      tr->n = (int)idx.size();
      tr->row = idx;
      tr->smt_of.resize(idx.size());
      for (size_t k = 0; k < idx.size(); k++) tr->smt_of[k] = smt[idx[k]];
      tr->rec.upload(recs.data(), recs.size(), s.stream);
      tr->chan.upload(chans.data(), chans.size(), s.stream);
      MIG_HIP(hipStreamSynchronize(s.stream));
      s.receptors.push_back(std::move(tr));
      found = (int)s.receptors.size() - 1;
    }
    g.rec_idx = found;
  }
  s.have_receptor = true;
  s.n_rec_atoms_in = n;
  s.flex_rows.clear();
}

// Rotations given with mi_scorer_set_rotations apply to exactly one scoring call: upload at entry, forget at exit.
struct RotScope {
  Scorer &s;
  RotScope(Scorer &sc, int B) : s(sc) {
    if (s.next_rot.empty()) return;
    std::vector<float> r;
    r.swap(s.next_rot);  // consumed even if the call fails
    MIG_CHECK((int)r.size() == 4 * B, 1, "mi_scorer_set_rotations was given a different number of poses than this call");
    s.d_rot.upload(r.data(), r.size(), s.stream);
    s.cur_rot = s.d_rot.p;
  }
  ~RotScope() { s.cur_rot = nullptr; }
};

// Declare the receptor rows that move with every pose (flexible side chains).  In the reference these are
// the first num_flex rows of receptor_coords (dl_scorer.cpp:150-193) and receptor_map sends their
// gradients back to the model (cnn_torch_scorer.cpp:216-224).
static void set_flex(Scorer &s, const int32_t *rows, int n_flex) {
  MIG_CHECK(s.have_receptor, 4, "mi_scorer_set_receptor must be called before mi_scorer_set_flex");
  MIG_CHECK(n_flex >= 0 && (n_flex == 0 || rows), 1, "bad flex arguments");
  std::vector<int> slot_of_row(s.n_rec_atoms_in, -1);
  for (int j = 0; j < n_flex; j++) {
    MIG_CHECK(rows[j] >= 0 && rows[j] < s.n_rec_atoms_in, 1, "flex row out of range");
    MIG_CHECK(slot_of_row[rows[j]] < 0, 1, "duplicate flex row");
    slot_of_row[rows[j]] = j;
  }
  for (auto &g : s.groups) {
    TypedReceptor &tr = *s.receptors[g.rec_idx];
    Model *m = s.models[g.first_model];
    std::vector<int> slot(tr.n, -1), perm, chan;
    std::vector<LigConsts> lc;
    for (int k = 0; k < tr.n; k++) {
      const int fs = slot_of_row[tr.row[k]];
      slot[k] = fs;
      if (fs < 0) continue;
      const DensityConsts &dc = m->dens[tr.smt_of[k]];
      perm.push_back(fs);
      lc.push_back(LigConsts{dc.ar, dc.t2, dc.g2, dc.kexp, dc.inv_ar});
      chan.push_back(tr.map.chan_of_smt[tr.smt_of[k]]);
    }
    tr.n_flex_typed = (int)perm.size();
    tr.flex_slot.upload(slot.data(), slot.size(), s.stream);
    tr.flex_perm.upload(perm.data(), perm.size(), s.stream);
    tr.flex_chan.upload(chan.data(), chan.size(), s.stream);
    tr.flex_consts.upload(lc.data(), lc.size(), s.stream);
    MIG_HIP(hipStreamSynchronize(s.stream));
  }
  s.flex_rows.assign(rows, rows + n_flex);
}

struct LigSetup {
  int n_lig = 0;  // typed ligand atoms (ragged: the largest count of any pose)
  bool ragged = false;
  // the device-side description gather_pose_atoms / voxel_backward read (the group's cached arrays, or the per-pose ones)
  const int *perm = nullptr, *chan = nullptr;
  const LigConsts *consts = nullptr;
  const unsigned char *typed = nullptr;
};

// Ragged batch (virtual screening): pose b has its own ligand -- rows [0, rows_b) of
// lig_smt[b][0..L) are real atoms (smt >= 0), the rest padding (-1).  Same typing / stable channel sort as
// setup_ligand, per pose.
static LigSetup setup_ligand_ragged(Scorer &s, const VoxGroup &g, const int32_t *lig_smt, int B, int L) {
  Model *m = s.models[g.first_model];
  std::vector<int> perm((size_t)B * L, 0), chan((size_t)B * L, 0), rows(B), nl(B);
  std::vector<LigConsts> lc((size_t)B * L);
  std::vector<unsigned char> typed((size_t)B * L, 0);
  std::vector<int> idx;
  int max_n = 0;
  for (int b = 0; b < B; b++) {
    const int32_t *t = lig_smt + (size_t)b * L;
    int r = 0;
    while (r < L && t[r] >= 0) r++;
    for (int i = r; i < L; i++) MIG_CHECK(t[i] < 0, 1, "ragged ligand rows: padding (-1) must follow the real atoms");
    rows[b] = r;
    idx.clear();
    for (int i = 0; i < r; i++) {
      MIG_CHECK(t[i] < kNumSminaTypes, 1, "ligand smina type out of range");
      if (m->d.ligmap.chan_of_smt[t[i]] >= 0) {
        idx.push_back(i);
        typed[(size_t)b * L + i] = 1;
      }
    }
    std::stable_sort(idx.begin(), idx.end(),
                     [&](int a, int c) { return m->d.ligmap.chan_of_smt[t[a]] < m->d.ligmap.chan_of_smt[t[c]]; });
    nl[b] = (int)idx.size();
    max_n = std::max(max_n, nl[b]);
    for (size_t k = 0; k < idx.size(); k++) {
      const DensityConsts &dc = m->dens[t[idx[k]]];
      perm[(size_t)b * L + k] = idx[k];
      lc[(size_t)b * L + k] = LigConsts{dc.ar, dc.t2, dc.g2, dc.kexp, dc.inv_ar};
      chan[(size_t)b * L + k] = m->d.ligmap.chan_of_smt[t[idx[k]]] + m->d.recmap.n_channels;
    }
  }
  s.d_lig_perm.upload(perm.data(), perm.size(), s.stream);
  s.d_lig_chan.upload(chan.data(), chan.size(), s.stream);
  s.d_lig_consts.upload(lc.data(), lc.size(), s.stream);
  s.d_lig_typed.upload(typed.data(), typed.size(), s.stream);
  s.d_pose_rows.upload(rows.data(), rows.size(), s.stream);
  s.d_pose_nlig.upload(nl.data(), nl.size(), s.stream);
  MIG_HIP(hipStreamSynchronize(s.stream));
  LigSetup ls;
  ls.n_lig = max_n;
  ls.ragged = true;
  ls.perm = s.d_lig_perm.p, ls.chan = s.d_lig_chan.p, ls.consts = s.d_lig_consts.p, ls.typed = s.d_lig_typed.p;
  return ls;
}

// Type the ligand rows with the group's ligand map, upload permutation / constants.
static LigSetup setup_ligand(Scorer &s, const VoxGroup &g, const int32_t *lig_smt, int L) {
  Model *m = s.models[g.first_model];
  const size_t gi = (size_t)(&g - s.groups.data());
  while (s.lig_dev.size() < s.groups.size()) s.lig_dev.emplace_back(new Scorer::LigDev);
  Scorer::LigDev &d = *s.lig_dev[gi < s.lig_dev.size() ? gi : 0];
  auto made = [&]() {
    LigSetup ls;
    ls.n_lig = d.n;
    ls.perm = d.perm.p, ls.chan = d.chan.p, ls.consts = d.consts.p, ls.typed = d.typed.p;
    return ls;
  };
  // the same ligand is scored call after call (poses of one docking run): keep its device-side description
  if (d.valid && (int)d.smt.size() == L && std::equal(d.smt.begin(), d.smt.end(), lig_smt)) return made();
  d.valid = false;
  std::vector<int> idx;
  std::vector<unsigned char> typed(L, 0);
  for (int i = 0; i < L; i++) {
    int t = lig_smt[i];
    MIG_CHECK(t >= 0 && t < kNumSminaTypes, 1, "ligand smina type out of range");
    if (m->d.ligmap.chan_of_smt[t] >= 0) {
      idx.push_back(i);
      typed[i] = 1;
    }
  }
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
    return m->d.ligmap.chan_of_smt[lig_smt[a]] < m->d.ligmap.chan_of_smt[lig_smt[b]];
  });
  std::vector<int> chan(idx.size());
  std::vector<LigConsts> lc(idx.size());
  for (size_t k = 0; k < idx.size(); k++) {
    const DensityConsts &dc = m->dens[lig_smt[idx[k]]];
    lc[k] = LigConsts{dc.ar, dc.t2, dc.g2, dc.kexp, dc.inv_ar};
    chan[k] = m->d.ligmap.chan_of_smt[lig_smt[idx[k]]] + m->d.recmap.n_channels;  // torch_model.cpp:168
  }
  // (a pending device-output call may still read the arrays being replaced: drain the stream first)
  MIG_HIP(hipStreamSynchronize(s.stream));
  d.perm.upload(idx.data(), idx.size(), s.stream);
  d.chan.upload(chan.data(), chan.size(), s.stream);
  d.consts.upload(lc.data(), lc.size(), s.stream);
  d.typed.upload(typed.data(), typed.size(), s.stream);
  // the uploads read from stack vectors: make them complete before those die
  MIG_HIP(hipStreamSynchronize(s.stream));
  d.n = (int)idx.size();
  d.smt.assign(lig_smt, lig_smt + L);
  d.valid = true;
  return made();
}

// gather + voxelize poses [b0, b0+nb) of the batch for one group. mode: 0 full grid, 1/2 pooled.
static void voxelize_chunk(Scorer &s, const VoxGroup &g, const LigSetup &ls, const float *d_lig_xyz, int L,
                           const float *d_centers_in, unsigned flags, int b0, int nb, int mode, float *out,
                           unsigned char *argmax_out = nullptr, hipStream_t vs = nullptr, int set = 0, bool split = false,
                           bool skip_gather = false) {
  if (!vs) vs = s.stream;
  DevBuf<AtomRec> &cand = set ? s.d_cand2 : s.d_cand;
  DevBuf<int> &cand_chan = set ? s.d_cand_chan2 : s.d_cand_chan;
  DevBuf<int> &cand_n = set ? s.d_cand_n2 : s.d_cand_n;
  Model *m = s.models[g.first_model];
  TypedReceptor &tr = *s.receptors[g.rec_idx];
  const int cap = tr.n + ls.n_lig + 1;
  const int nt_axis = cdiv(cdiv(m->N, 2), 4);
  const int n_slab = nt_axis <= kMaxSlabs ? nt_axis : 1;  // one candidate list per x-slab of tiles
  cand.ensure((size_t)s.cap * n_slab * cap);
  cand_chan.ensure((size_t)s.cap * n_slab * cap);
  cand_n.ensure((size_t)s.cap * n_slab);
  GatherArgs ga{};
  ga.rec = tr.rec.p;
  ga.rec_chan = tr.chan.p;
  ga.n_rec = tr.n;
  if (s.cur_flex && !s.flex_rows.empty()) {
    ga.rec_flex_slot = tr.flex_slot.p;
    ga.n_flex = (int)s.flex_rows.size();
    ga.flex_xyz = s.cur_flex + (size_t)b0 * ga.n_flex * 3;
  }
  ga.lig_xyz = d_lig_xyz + (size_t)b0 * L * 3;
  ga.L = L;
  ga.lig_perm = ls.perm;
  ga.lig_consts = ls.consts;
  ga.lig_chan = ls.chan;
  ga.n_lig = ls.n_lig;
  ga.lig_typed = ls.typed;
  if (ls.ragged) {  // per-pose ligand descriptions
    ga.lig_perm += (size_t)b0 * L;
    ga.lig_consts += (size_t)b0 * L;
    ga.lig_chan += (size_t)b0 * L;
    ga.lig_typed += (size_t)b0 * L;
    ga.pose_rows = s.d_pose_rows.p + b0;
    ga.pose_n_lig = s.d_pose_nlig.p + b0;
  }
  ga.centers_in = d_centers_in ? d_centers_in + (size_t)b0 * 3 : nullptr;
  ga.rot = s.cur_rot ? s.cur_rot + (size_t)b0 * 4 : nullptr;
  ga.center_typed_only = (flags & MI_CENTER_TYPED_ONLY) ? 1 : 0;
  ga.half_dim = m->d.dimension / 2.0f;
  ga.centers_out = s.d_centers.p + (size_t)set * s.centers_stride + (size_t)b0 * 3;
  ga.cand = cand.p;
  ga.cand_chan = cand_chan.p;
  ga.cand_n = cand_n.p;
  ga.cap = cap;
  ga.n_slab = n_slab;
  s.dbg_cap = cap, s.dbg_nslab = n_slab;
  ga.res = m->d.resolution;
  if (!skip_gather) {  // (skipped only by mi_debug_vox_stress: the lists of the previous launch, untouched)
    ProfScope ps(s, "gather_pose_atoms", 0.0, (double)nb * (tr.n + ls.n_lig) * 36.0, nb, vs);
    launch_gather(ga, nb, vs);
  }
  VoxArgs va{};
  va.cand = cand.p;
  va.cand_chan = cand_chan.p;
  va.cand_n = cand_n.p;
  va.cap = cap;
  va.n_slab = n_slab;
  va.centers = ga.centers_out;
  va.N = m->N;
  va.tiles_per_axis = cdiv(cdiv(m->N, 2), 4);
  va.C = m->C;
  va.Cp = m->Cp;
  va.res = m->d.resolution;
  va.half_dim = ga.half_dim;
  va.qa = m->qa;
  va.qb = m->qb;
  va.qc = m->qc;
  va.out = out;
  va.argmax_out = argmax_out;
  va.trap = s.dbg_trap;
  if (split) {  // the pooled grid in split format (Model::pooled_split_ok): whole octets, h | l halves
    MIG_CHECK(mode != 0 && !argmax_out, 2, "split-format voxel grid: pooled forward output only");
    va.split = 1;
    va.Cp = m->Cp8;
    s.d_ovf.ensure(1);
    va.overflow = s.d_ovf.p;
    const size_t nt3 = (size_t)va.tiles_per_axis * va.tiles_per_axis * va.tiles_per_axis;
    s.d_occ[set].ensure((size_t)s.cap * nt3 * 8);
    va.occ = s.d_occ[set].p;
  }
  {
    // algorithmic bytes: the un-fused figure C*N^3*4 written once per pose
    ProfScope ps(s, mode == 0 ? "voxelize_tiles<full>" : "voxelize_tiles<pooled>", 0.0,
                 (double)nb * m->C * m->N * m->N * m->N * 4.0, nb, vs);
    launch_voxelize(va, nb, mode, vs);
  }
  MIG_HIP(hipGetLastError());
}

// Does this call run on the bf16 kernels?  Forward: whenever the scorer asks for it (an unsupported layer
// program is an error).  Gradient: max-pool networks (Default2017, Dense); average-pooling networks keep
// their gradient calls in fp32.
static bool use_bf16(Scorer &s, Model &m, bool grad) {
  if (s.precision != 1) return false;
  if (!grad) {
    std::call_once(m.hsteps_once, [&] {
      try {
        build_bf16_program(m, false);
      } catch (const std::exception &e) {
        m.hsteps_error = e.what();
      }
    });
    MIG_CHECK(m.hsteps_error.empty(), 2, "bf16 program of " + m.d.name + ": " + m.hsteps_error);
    return true;
  }
  std::call_once(m.hgsteps_once, [&] {
    try {
      build_bf16_program(m, true);
    } catch (const std::exception &e) {
      m.hgsteps_error = e.what();
    }
  });
  return m.hgsteps_error.empty();
}

// Launch arguments of a layer's split-fp16 twin: the tensors of `a`, the tile pick_tile chooses for `nb` poses (or the
// twin's own throughput tile), conv3d_h2_kernel's M-tile geometry, LDS pads and weights-in-LDS variant for that tile.
static void h2_launch_args(const ConvPlan &cp, const ConvArgs &a, int nb, ConvArgs &h, int &cfg) {
  h = cp.h2;
  h.in = a.in, h.in_cs = a.in_cs, h.out = a.out, h.out_cs = a.out_cs;
  ConvArgs geo = a;
  pick_tile(cp, nb, geo, cfg);
  h.tcx = geo.tcx, h.tcy = geo.tcy, h.tcz = geo.tcz, h.ntx = geo.ntx, h.nty = geo.nty, h.ntz = geo.ntz, h.mt_x = geo.mt_x;
  if (h.post_w) h.post_rows = geo.post_rows;  // (pick_tile: rows of the tile it chose)
  if (!cp.h2_planar) {
    // conv3d_h2_16_kernel: weights through LDS for the 6^3 layers (1), every layer the buffers fit (2), never (0)
    h.h2_wlds = 0;
    if (h.coutp == 16 && h.ksize == 3) h.h2_wlds = option(OPT_MI_GNINA_H16_WLDS) ? atoi(option(OPT_MI_GNINA_H16_WLDS)) : 1;
    return;
  }
  // conv3d_h2_kernel: its own M-tile geometry and LDS pads per tile
  const bool lat = cfg != cp.cfg;
  if (!lat && cp.h2_cfg >= 0 && !h.post_w) {  // its own throughput tile
    cfg = cp.h2_cfg;
    h.tcx = cp.h2.tcx, h.tcy = cp.h2.tcy, h.tcz = cp.h2.tcz;
    h.ntx = cp.h2.ntx, h.nty = cp.h2.nty, h.ntz = cp.h2.ntz;
  }
  h.mt_x = lat ? cp.h2_lat_mt : cp.h2.mt_x;
  h.h2_pad_y = lat ? cp.h2_lat_pad[0] : cp.h2.h2_pad_y;
  h.h2_pad_x = lat ? cp.h2_lat_pad[1] : cp.h2.h2_pad_x;
  // weights through LDS where the kernel shape has that variant (conv3d_h2.hip launch_h2_k3)
  int wm_, wn_, tm_, tn_;
  conv_cfg_shape(cfg, &wm_, &wn_, &tm_, &tn_);
  // (default 2: weights through LDS, two poses per workgroup -- measured 1.76 against 1.94 (one pose) and
  // 1.9-2.0 ms (weights from L1 / L2, double-buffered tile) on the headline's first conv; MI_GNINA_H2_WLDS=0/1/2)
  const char *ev = option(OPT_MI_GNINA_H2_WLDS);
  h.h2_wlds = (wn_ == 1 && tm_ <= 2) ? (ev ? atoi(ev) : 2) : 0;
  // (L2 prefetch of the next item's tile under this item's K loop: measured, no gain -- 1.722 vs 1.717 ms; opt-in)
  h.h2_prefetch = (option(OPT_MI_GNINA_H2_PF) && atoi(option(OPT_MI_GNINA_H2_PF)) != 0) ? 1 : 0;
  // stationary weights + a ring of halo tiles for the first convolution where the kernel covers it (conv3d_h2_ws.hip);
  // MI_GNINA_H2_WS = ring size 2 .. 5, 0 = conv3d_h2_kernel
  h.h2_ws = option(OPT_MI_GNINA_H2_WS) ? atoi(option(OPT_MI_GNINA_H2_WS)) : 0;
}

// Run the layer program of model mi on `nb` poses whose pooled grid already sits in
// act[input_dst]; writes pose/aff/loss at out offsets.
// The layer program a call runs for model m (forward or gradient, bf16 / split-fp16 / fp32-MFMA).
static const std::vector<Step> &program_steps(Scorer &s, Model &m, bool grad) {
  const bool bf16 = use_bf16(s, m, grad);
  return bf16 ? (grad ? m.hgsteps : m.hsteps) : (grad ? m.gsteps : (s.conv_path != 0 ? m.steps : m.steps32));
}

// (step_lo, step_hi: the steps [step_lo, step_hi) of the program only -- the lanes enqueue their programs a few steps at a time)
static void run_program(Scorer &s, int mi, int nb, float *pose, float *aff, float *loss, bool grad = false,
                        size_t pooled_slot = 0, bool pooled_split = false, int step_lo = 0, int step_hi = 1 << 30) {
  Model *m = s.models[mi];
  if (m->overlap) {
    if (step_lo > 0) return;
    const long N3 = (long)m->N * m->N * m->N;
    const float *grid = act_buf(s, pooled_slot, (size_t)s.cap * 2 * N3);
    s.d_ave.ensure(s.cap);
    ProfScope ps(s, "overlap_forward", 2.0 * nb * N3, (double)nb * 2 * N3 * 4.0, nb);
    launch_overlap_forward(grid, nb, N3, pose, aff, loss, s.d_ave.p, s.stream);
    MIG_HIP(hipGetLastError());
    return;
  }
  const bool bf16 = use_bf16(s, *m, grad);
  const std::vector<Step> &steps = program_steps(s, *m, grad);
  if (s.last_lane.size() < s.models.size()) s.last_lane.resize(s.models.size(), 0);
  s.last_lane[mi] = s.act_lane;
  auto arg_ptr = [&](int id) -> unsigned char * {
    const BufDecl &bd = m->d.bufs[id];
    const size_t slot = id == m->input_dst ? pooled_slot : (size_t)id + (size_t)s.act_lane * kLaneSlots;
    return argm_buf(s, slot, (size_t)s.cap * bd.S * bd.S * bd.S * m->buf_cp[id]);
  };
  auto buf_ptr = [&](int id) -> float * {
    const BufDecl &bd = m->d.bufs[id];
    // the pooled voxel grid lives in a dedicated slot shared by all models of a voxelization group
    const size_t slot = id == m->input_dst ? pooled_slot : (size_t)id + (size_t)s.act_lane * kLaneSlots;
    return act_buf(s, slot, (size_t)s.cap * bd.S * bd.S * bd.S * (id == m->input_dst ? pooled_stride(m) : m->buf_cp[id]));
  };
  // split-format tensors (Model::buf_split) exist in the split-fp16 forward program only
  const bool fwd_h2 = !bf16 && !grad && s.conv_path != 0;
  MIG_CHECK(!pooled_split || (fwd_h2 && m->pooled_split_ok), 2, "pooled grid written split for a program that reads fp32");
  auto is_split = [&](int id) { return fwd_h2 && (id == m->input_dst ? pooled_split : (bool)m->buf_split[id]); };
  // per-pose scoring calls: a whole-grid max pool and the heads behind it run as one launch (conv3d.hip gmax_heads_kernel);
  // a pure function of the program and the call, so that a lane's slices agree on it wherever they are cut
  auto gmax_fused = [&](int si) {
    if (si < 0 || si + 1 >= (int)steps.size()) return false;
    const Step &g = steps[si], &fc = steps[si + 1];
    return g.kind == OpKind::GMax && fc.kind == OpKind::Fc && !grad && !g.src_bf16 && nb <= 8 && !s.profile &&
           !option(OPT_MI_GNINA_NO_GMAX_FUSE) && fc.src == g.dst && fc.n_in == g.C && m->buf_cp[g.dst] == g.C && gmax_heads_covers(g.C);
  };
  int si0 = std::max(step_lo, 0);
  if (si0 > 0 && gmax_fused(si0 - 1)) si0++;  // (the previous slice's max pool took this slice's first step along)
  for (int si = si0; si < std::min(step_hi, (int)steps.size()); si++) {
    const Step &st = steps[si];
    switch (st.kind) {
      case OpKind::Conv: {
        ConvArgs a = st.conv.a;
        a.in = buf_ptr(st.conv.src);
        a.in_cs = m->buf_cp[st.conv.src];
        a.out = buf_ptr(st.conv.dst);
        a.out_cs = m->buf_cp[st.conv.dst];
        if (grad && a.pool == 1) a.argmax_out = arg_ptr(st.conv.dst);
        a.sparse = (st.conv.src == m->input_dst && !st.has_bn && !bf16) ? 1 : 0;  // the pooled voxel grid is ~12 % dense
        // ReLU'd activations (Default2017 / Default2018 convs behind the first one): channel-major K order with the
        // per-MFMA zero test, no per-tile quad dropping (ConvArgs::sparse)
        // (not on the split-fp16 kernels: three 32-cycle MFMAs per test and an LDS round trip per dead step -- conv2 / conv3 of
        // Default2017 run 7 % / 14 % faster without it)
        const bool use_h2 = !bf16 && st.conv.has_h2 && s.conv_path != 0;
        if (!use_h2 && !a.sparse && !st.has_bn && !bf16 && a.ksize == 3 && st.conv.src != m->input_dst && !option(OPT_MI_GNINA_NO_RELU_SKIP)) {
          if (st.relu_skip == 0 && nb >= 32) {
            s.d_probe.ensure(2);
            MIG_HIP(hipMemsetAsync(s.d_probe.p, 0, 2 * sizeof(unsigned), s.stream));
            launch_zero_cell_probe(a.in, 32, st.conv.cin, a.in_cs, a.S, s.d_probe.p, s.stream);
            unsigned cnt[2] = {0u, 0u};
            MIG_HIP(hipMemcpyAsync(cnt, s.d_probe.p, sizeof cnt, hipMemcpyDeviceToHost, s.stream));
            MIG_HIP(hipStreamSynchronize(s.stream));
            // measured break-even: Default2017 (cells 50 % zero) gains 25 %, Default2018 (13-20 %) loses 1-2 %
            st.relu_skip = (cnt[1] && (double)cnt[0] >= 0.30 * (double)cnt[1]) ? 1 : 2;
          }
          a.korder = 1;  // either way the channel-major K order: same bits with and without the test
          a.sparse = st.relu_skip == 2 ? 0 : 2;
        }
        if (st.conv.a.korder) a.sparse = 2;  // Dense-block conv planned with the channel-major K order: per-MFMA test on
        if (option(OPT_MI_GNINA_NO_SPARSE)) a.sparse = 0;
        {
          const double S3 = (double)a.S * a.S * a.S;
          const int taps = a.ksize * a.ksize * a.ksize;
          char nm[96];
          snprintf(nm, sizeof nm, "conv%d_s%d_%dto%d%s%s", a.ksize, a.S, st.conv.cin, a.cout,
                   a.post_w ? "+conv1" : "", a.pool ? "_pool" : "");
          if (bf16) strncat(nm, "_bf16", sizeof nm - strlen(nm) - 1);
          if (use_h2 && is_split(st.conv.src) && (st.conv.has_d16 || st.conv.has_k1s)) strncat(nm, "_sp", sizeof nm - strlen(nm) - 1);
          if (use_h2) strncat(nm, "_h2", sizeof nm - strlen(nm) - 1);
          ProfScope ps(s, nm, 2.0 * nb * S3 * (taps * st.conv.cin * a.cout + (a.post_w ? (double)a.cout * st.post_cout : 0.0)),
                       (double)nb * S3 * 4.0 * (st.conv.cin + a.cout / (a.pool ? 8.0 : 1.0)), nb);
          if (bf16) {
            launch_conv_bf16(a, st.conv.cfg, nb, s.stream);
          } else if (use_h2 && is_split(st.conv.src) && (st.conv.has_d16 || st.conv.has_k1s)) {
            // Dense family on split-format tensors (conv3d_h2_dense.hip): a block layer, or the 1x1x1 transition behind a block
            ConvArgs h = st.conv.has_d16 ? st.conv.d16 : st.conv.h2;
            h.in = a.in, h.in_cs = a.in_cs, h.out = a.out, h.out_cs = a.out_cs;
            h.in_split = 1;
            h.out_split = is_split(st.conv.dst) ? 1 : 0;
            h.argmax_out = nullptr;
            s.d_ovf.ensure(1);
            h.h2_overflow = s.d_ovf.p;
            h.mfma_count = nullptr;
            h.h2_honly = (!grad && s.h2_honly) ? 1 : 0;  // (MI_PRECISION_FP16: scoring calls only)
            // (timing experiments, wrong results: conv3d_h2_dense.hip's h2_dbg bits)
            if (const char *ev = option(st.conv.has_d16 ? OPT_MI_GNINA_D16_DBG : OPT_MI_GNINA_K1S_DBG)) h.h2_dbg = atoi(ev);
            if (st.conv.has_d16) {
              MIG_CHECK(h.out_split, 2, "Dense-block layer planned on split tensors writes a buffer that is not split");
              if (st.conv.d16_lat_tc[0] > 0 && (long)nb * h.ntx * h.nty * h.ntz < 256) {
                const int cells = h.S / 2, *tc = st.conv.d16_lat_tc;
                if ((long)nb * cdiv(cells, tc[0]) * cdiv(cells, tc[1]) * cdiv(cells, tc[2]) <= 512) {
                  h.tcx = tc[0], h.tcy = tc[1], h.tcz = tc[2];
                  h.ntx = cdiv(cells, tc[0]), h.nty = cdiv(cells, tc[1]), h.ntz = cdiv(cells, tc[2]);
                  h.h2_pad_y = h.h2_pad_x = 0;
                }
              }
              if (const char *ev = option(OPT_MI_GNINA_D16_NP)) h.h2_wlds = atoi(ev) >= 2 ? 2 : 1;
              h.d16_group = option(OPT_MI_GNINA_D16_GROUP_MAX) ? atoi(option(OPT_MI_GNINA_D16_GROUP_MAX)) : 0;  // (cap on the chunk groups of per-pose launches; 0 = the launcher's)
              h.h2_persist = -1;  // (persistent launch, as many workgroups as the chip holds; MI_GNINA_D16_PERSIST=0: one per item, n: at most n per CU)
              if (const char *ev = option(OPT_MI_GNINA_D16_PERSIST)) h.h2_persist = atoi(ev);
              launch_conv_h2_d16(h, nb, s.stream);
            } else {
              h.h2_persist = -1;
              if (const char *ev = option(OPT_MI_GNINA_K1S_PERSIST)) h.h2_persist = atoi(ev);
              launch_conv_h2_k1s(h, nb, s.stream);
            }
          } else if (use_h2) {
            // split-fp16 kernel: same tensors, same tiles (pick_tile), own K chunking and weights
            ConvArgs h;
            int cfg;
            h2_launch_args(st.conv, a, nb, h, cfg);
            h.argmax_out = a.argmax_out;
            h.sparse = a.sparse == 1 ? 1 : 0;  // the zero test pays on the pooled voxel grid only
            if (option(OPT_MI_GNINA_H2_NO_SKIP)) h.sparse = 0;
            if (st.conv.h2_planar) {  // tensor formats
              h.in_split = is_split(st.conv.src) ? 1 : 0;
              h.out_split = is_split(st.conv.dst) ? 1 : 0;
              if (h.in_split && st.conv.src == m->input_dst) {
                h.in_cs = m->Cp8;
                // the voxelizer's occupancy bytes of this buffer set -- opt-in (MI_GNINA_H2_OCC=1): on the headline workload a
                // quarter of the (tile, octet) chunks is empty, and reading the bytes ahead of the first DMA costs as much
                // (conv3d_h2_ws_kernel builds its round list from them: always, when that kernel may take the launch)
                if ((option(OPT_MI_GNINA_H2_OCC) && atoi(option(OPT_MI_GNINA_H2_OCC))) || h.h2_ws > 0) {
                  h.in_occ = s.d_occ[pooled_slot == kPooledSlot2 ? 1 : 0].p;
                  h.occ_nt = cdiv(cdiv(m->N, 2), 4);
                }
              }
            } else {
              MIG_CHECK(!is_split(st.conv.src) && !is_split(st.conv.dst), 2, "split-format tensor at a layer that cannot take it");
            }
            s.d_ovf.ensure(1);
            h.h2_overflow = s.d_ovf.p;
            h.h2_honly = (!grad && s.h2_honly) ? 1 : 0;  // (MI_PRECISION_FP16: scoring calls only)
            if (h.h2_honly) h.h2_ws = 0;
            if (const char *ev = option(OPT_MI_GNINA_H2_DBG)) h.h2_dbg = atoi(ev);
            h.mfma_count = (h.sparse && h.coutp != 16) ? prof_counter(s, ps) : nullptr;
            launch_conv_h2(h, cfg, nb, s.stream);
          } else {
            int cfg;
            pick_tile(st.conv, nb, a, cfg);
            a.mfma_count = (a.sparse && !a.bias_tab) ? prof_counter(s, ps) : nullptr;  // (32-wide zero-skipping kernels)
            launch_conv(a, cfg, nb, s.stream);
          }
        }
        break;
      }
      case OpKind::Pool:
        launch_pool_cl(buf_ptr(st.src), buf_ptr(st.dst), nb, st.C, m->buf_cp[st.src], m->buf_cp[st.dst],
                       m->d.bufs[st.src].S, st.pool_mode, s.stream);
        break;
      case OpKind::GMax:
        // per-pose scoring calls: the whole-grid max pool and the heads behind it in one launch (conv3d.hip gmax_heads_kernel)
        if (gmax_fused(si)) {
          const Step &fc = steps[si + 1];
          launch_gmax_heads(buf_ptr(st.src), buf_ptr(st.dst), nb, st.C, m->buf_cp[st.src], m->buf_cp[st.dst], m->d.bufs[st.src].S,
                            m->dev_data.p + fc.w_off, m->dev_data.p + fc.b_off, m->d.skip_softmax, m->d.apply_logistic_loss, pose, aff,
                            loss, s.stream);
          si++;  // (the Fc step is done)
          break;
        }
        if (st.src_bf16)
          launch_gmax_bf16(buf_ptr(st.src), buf_ptr(st.dst), nb, st.C, m->buf_cp[st.src], m->buf_cp[st.dst],
                           m->d.bufs[st.src].S, s.stream);
        else
          launch_gmax(buf_ptr(st.src), buf_ptr(st.dst), nb, st.C, m->buf_cp[st.src], m->buf_cp[st.dst],
                      m->d.bufs[st.src].S, s.stream);
        break;
      case OpKind::Fc: {
        ProfScope ps(s, "fc_heads", 2.0 * nb * 3.0 * st.n_in, (double)nb * st.n_in * 4.0, nb);
        launch_fc_heads(buf_ptr(st.src), m->dev_data.p + st.w_off, m->dev_data.p + st.b_off, st.n_in,
                        m->d.skip_softmax, m->d.apply_logistic_loss, pose, aff, loss,
                        grad ? (s.d_raw3.ensure((size_t)3 * s.cap * (s.act_lane + 1)), s.d_raw3.p + (size_t)3 * s.cap * s.act_lane) : nullptr, nb, s.stream);
        break;
      }
      case OpKind::Overlap:  // handled above: the Overlap model has no layer program
        break;
    }
  }
  MIG_HIP(hipGetLastError());
}

// Gradient of the loss w.r.t. the pooled voxel grid for `nb` poses whose forward pass (gradient
// program) has just run: fc backward -> transposed convs (max-unpool / ReLU mask applied while
// staging) -> avg un-pool.  Returns the device pointer of dL/d(pooled grid).
static float *run_backward(Scorer &s, int mi, int nb) {
  Model *m = s.models[mi];
  if (m->overlap) {  // d loss / d (full grid) from the grid itself and the mean run_program left in d_ave
    const long N3 = (long)m->N * m->N * m->N;
    const float *grid = act_buf(s, kPooledSlot, (size_t)s.cap * 2 * N3);
    float *gg = gact_buf(s, kPooledSlot, (size_t)s.cap * 2 * N3);
    ProfScope ps(s, "overlap_backward", 0.0, 0.0, nb);
    launch_overlap_backward(grid, s.d_ave.p, nb, N3, gg, s.stream);
    MIG_HIP(hipGetLastError());
    return gg;
  }
  const bool bf16 = use_bf16(s, *m, true);
  const std::vector<Step> &gsteps = bf16 ? m->hgsteps : m->gsteps;
  // (lanes: a model's buffers live in its lane's slot set; the pooled grid and its arg-max bytes belong to the model's voxel
  // group -- Scorer::grad_pooled_slot --, the GRADIENT of the pooled grid to the model: slot 0 of its lane's set)
  const size_t lane_base = (size_t)s.act_lane * kLaneSlots;
  auto slot_of = [&](int id) { return id == m->input_dst ? s.grad_pooled_slot : (size_t)id + lane_base; };
  auto gslot_of = [&](int id) { return id == m->input_dst ? lane_base : (size_t)id + lane_base; };
  auto count_of = [&](int id) {
    const BufDecl &bd = m->d.bufs[id];
    return (size_t)s.cap * bd.S * bd.S * bd.S * m->buf_cp[id];
  };
  auto act_ptr = [&](int id) { return act_buf(s, slot_of(id), count_of(id)); };
  auto g_ptr = [&](int id) { return gact_buf(s, gslot_of(id), count_of(id)); };
  // conv + ReLU -> stand-alone average pool (Default2018): the transposed conv un-pools the half-resolution gradient
  // while it stages (ConvArgs::in_mode 3) and no full-resolution gradient tensor is written in between.  Generic fp32
  // kernel only; the pool must be the only reader of the conv's output.
  auto avg_unpool_fused = [&](int ip) {
    if (bf16 || ip < 1 || ip >= (int)gsteps.size() || option(OPT_MI_GNINA_NO_UNPOOL_FUSE)) return false;
    const Step &pl = gsteps[ip], &cv = gsteps[ip - 1];
    if (pl.kind != OpKind::Pool || pl.pool_mode != 2 || cv.kind != OpKind::Conv || !cv.has_bwd) return false;
    if (cv.conv.dst != pl.src || !cv.conv.a.relu || cv.conv.a.pool != 0 || cv.conv.a.out_c0 != 0) return false;
    auto n16 = [](int c) { return c == CONV_CFG_N16_TM1 || c == CONV_CFG_N16_TM2 || c == CONV_CFG_N16_TM3 || c == CONV_CFG_N16_TM4; };
    if (cv.has_bwd_lig || n16(cv.bwd.cfg) || (cv.bwd.has_lat && n16(cv.bwd.lat_cfg))) return false;
    for (int j = 0; j < (int)gsteps.size(); j++)
      if (j != ip && gsteps[j].kind != OpKind::Fc && (gsteps[j].kind == OpKind::Conv ? gsteps[j].conv.src : gsteps[j].src) == pl.src)
        return false;
    return true;
  };
  // transposed convs on the split-fp16 kernel (default precision): the producer of a gradient such a conv will read masks
  // it by the ReLU and records its per-pose maximum; `ready` = that has happened for the buffer's gradient in this call
  const bool h2_bwd = !bf16 && s.conv_path != 0 && !option(OPT_MI_GNINA_NO_H2_BWD);
  const size_t nbufs = m->d.bufs.size();
  std::vector<char> ready(nbufs, 0), slice_ready(gsteps.size(), 0);
  if (h2_bwd) {
    const size_t slots = nbufs + gsteps.size();  // one per buffer (kind 1) and one per step (kind 2)
    MIG_CHECK(s.act_lane == 0 || slots <= kGamaxLaneSlots, 2, "gradient lanes: model with more gradient-maximum slots than a lane's share");
    s.d_gamax.ensure(s.act_lane == 0 ? slots * (size_t)s.cap : (size_t)(s.act_lane + 1) * kGamaxLaneSlots * s.cap);
    launch_zero_u32(s.d_gamax.p + (size_t)s.act_lane * kGamaxLaneSlots * s.cap, slots * (size_t)s.cap, s.stream);
  }
  auto amax_of = [&](int id) { return s.d_gamax.p + (size_t)s.act_lane * kGamaxLaneSlots * s.cap + (size_t)id * s.cap; };
  auto wants_mask = [&](int id) { return h2_bwd && id != m->input_dst && m->buf_bwd_h2[id]; };
  for (int i = (int)gsteps.size() - 1; i >= 0; i--) {
    const Step &st = gsteps[i];
    switch (st.kind) {
      case OpKind::Fc: {
        ProfScope ps(s, "fc_backward", 2.0 * nb * 2.0 * st.n_in, 0.0, nb);
        const bool mk = wants_mask(st.src);
        launch_fc_backward(s.d_raw3.p + (size_t)3 * s.cap * s.act_lane, m->dev_data.p + st.w_off, st.n_in, g_ptr(st.src), nb, s.stream,
                           mk ? act_ptr(st.src) : nullptr, mk ? amax_of(st.src) : nullptr);
        ready[st.src] = mk;
        break;
      }
      case OpKind::Conv: {
        MIG_CHECK(st.has_bwd, 1, "gradient not supported for this layer");
        // (rigid receptor: the first conv's transposed twin computes the ligand's channels of the grid gradient only)
        const bool lig_only = st.has_bwd_lig && !bf16 && (s.cur_flex == nullptr || s.flex_rows.empty());
        const int dst = st.conv.dst, src = st.conv.src;
        // the split-fp16 kernel: the gradient it reads must have been prepared by its producer
        const int h2_kind = (lig_only && !st.has_bwd_lig_h2) ? 0 : st.bwd_h2_kind;
        const bool use_h2 = h2_bwd && h2_kind != 0 && (h2_kind == 2 || ready[dst]);
        const ConvPlan &bp = use_h2 ? (lig_only ? st.bwd_lig_h2 : st.bwd_h2) : (lig_only ? st.bwd_lig : st.bwd);
        ConvArgs a = bp.a;
        if (use_h2) a.out_scale = st.bwd.a.out_scale, a.accumulate = st.bwd.a.accumulate;  // (Dense-block layers)
        a.in = g_ptr(dst) + st.conv.a.out_c0;  // Dense layers: the 16-channel slice this conv produced
        a.in_cs = m->buf_cp[dst];
        a.in_act = act_ptr(dst) + st.conv.a.out_c0;
        if (bf16 && !st.bwd.a.act_f32)  // bf16 activation tensor: the slice offset is in 2-byte elements
          a.in_act = reinterpret_cast<const float *>(reinterpret_cast<const unsigned short *>(act_ptr(dst)) + st.conv.a.out_c0);
        a.in_act_cs = m->buf_cp[dst];
        if (st.conv.a.pool == 1) {
          a.sparse = 1;  // un-pooled gradient: at most 1 of 8 voxels per cell is non-zero
          a.in_mode = 2;
          a.in_argmax = argm_buf(s, slot_of(dst), count_of(dst));
        } else if (avg_unpool_fused(i + 1)) {
          const Step &pl = gsteps[i + 1];
          a.in = g_ptr(pl.dst);
          a.in_cs = m->buf_cp[pl.dst];
          a.in_mode = 3;
          a.sparse = 0;
          MIG_CHECK(conv_lds_bytes(a) <= 160 * 1024, 2, "conv tile exceeds LDS");
        } else {
          a.in_mode = st.conv.a.relu ? 1 : 0;
          // a gradient masked by (activation > 0) is as sparse as the ReLU'd activation: per-MFMA zero test
          a.sparse = (a.in_mode == 1 && a.ksize == 3 && !bf16 && !option(OPT_MI_GNINA_NO_RELU_SKIP)) ? 2 : 0;
        }
        a.out = g_ptr(src);
        a.out_cs = m->buf_cp[src];
        // this launch produces the gradient of buffer `src`: prepare it for a split-fp16 consumer
        const bool n16_here = bp.cfg == CONV_CFG_N16_TM1 || bp.cfg == CONV_CFG_N16_TM2 || bp.cfg == CONV_CFG_N16_TM3 || bp.cfg == CONV_CFG_N16_TM4;
        const bool can_prepare = !n16_here && !(bp.has_lat && !use_h2 && bp.lat_cfg != CONV_CFG_4x1_1x1);  // (kernels whose epilogue masks)
        const bool mk = wants_mask(src) && !a.accumulate && !a.out_scale && can_prepare;
        ready[src] = mk;
        // Dense blocks: the next launch of this pass reads a channel slice of the gradient this launch writes (or adds to) --
        // its final value: masked and measured here, on those channels, instead of by launch_grad_mask_amax
        int slice_step = -1, slice_c0 = 0, slice_c1 = 0;
        if (h2_bwd && !mk && can_prepare && i > 0 && gsteps[i - 1].kind == OpKind::Conv && gsteps[i - 1].conv.dst == src &&
            gsteps[i - 1].bwd_h2_kind == 2 && !option(OPT_MI_GNINA_H2_BWD_PREPASS)) {
          const Step &nx = gsteps[i - 1];
          const bool nx_lig = nx.has_bwd_lig && (s.cur_flex == nullptr || s.flex_rows.empty());
          slice_c0 = nx.conv.a.out_c0, slice_c1 = slice_c0 + nx.conv.a.cout;
          if (!(nx_lig && !nx.has_bwd_lig_h2) && slice_c1 <= (lig_only ? 0 : st.conv.cin)) slice_step = i - 1;
        }
        if (slice_step >= 0) slice_ready[slice_step] = 1;
        const bool prep = mk || slice_step >= 0;
        const float *prep_mask = prep ? act_ptr(src) : nullptr;
        unsigned *prep_amax = mk ? amax_of(src) : slice_step >= 0 ? amax_of((int)nbufs + slice_step) : nullptr;
        const int prep_c0 = mk ? 0 : slice_c0, prep_c1 = mk ? (1 << 30) : slice_c1;
        char nm[96];
        const int cout_here = lig_only ? st.conv.cin - st.lig_c0 : st.conv.cin;  // grid-gradient channels this launch computes
        snprintf(nm, sizeof nm, "convT%d_s%d_%dto%d%s%s", a.ksize, a.S, st.conv.a.cout, cout_here, lig_only ? "_lig" : "", use_h2 ? "_h2" : "");
        const double S3 = (double)a.S * a.S * a.S;
        if (bf16) strncat(nm, "_bf16", sizeof nm - strlen(nm) - 1);
        if (use_h2 && h2_kind == 2 && !slice_ready[i]) {  // a slice of a concat buffer's gradient no launch has prepared: ReLU mask and per-pose maximum, in place
          ProfScope ps2(s, "grad_mask_amax", 0.0, 0.0, nb);
          launch_grad_mask_amax(g_ptr(dst) + st.conv.a.out_c0, act_ptr(dst) + st.conv.a.out_c0, st.conv.a.cout, m->buf_cp[dst],
                                m->buf_cp[dst], st.conv.a.pool ? (long)(a.S / 2) * (a.S / 2) * (a.S / 2) : (long)a.S * a.S * a.S, nb,
                                amax_of((int)nbufs + i), s.stream);
        }
        ProfScope ps(s, nm, 2.0 * nb * S3 * a.ksize * a.ksize * a.ksize * cout_here * st.conv.a.cout, 0.0, nb);
        if (use_h2) {
          ConvArgs h;
          int cfg;
          h2_launch_args(bp, a, nb, h, cfg);
          h.in_split = h.out_split = 0;
          h.h2_wlds = bp.h2_planar ? 1 : 0;  // (3x3x3: the gradient-pass variant is compiled for the weights-in-LDS shapes only)
          h.in_mode = st.conv.a.pool == 1 ? 2 : 0;  // (the ReLU mask is in the gradient already)
          h.in_argmax = a.in_argmax;
          h.in_amax = h2_kind == 2 ? amax_of((int)nbufs + i) : amax_of(dst);
          h.out_scale = a.out_scale, h.accumulate = a.accumulate;
          h.sparse = option(OPT_MI_GNINA_H2_BWD_SKIP) ? 1 : 0;
          h.out_mask = prep_mask;
          h.out_mask_cs = m->buf_cp[src];
          h.out_mask_c0 = prep_c0, h.out_mask_c1 = prep_c1;
          h.out_amax = prep_amax;
          s.d_ovf.ensure(1);
          h.h2_overflow = s.d_ovf.p;
          launch_conv_h2(h, cfg, nb, s.stream);
        } else if (bf16) {
          a.sparse = 0;
          launch_conv_bf16(a, st.bwd.cfg, nb, s.stream);
        } else {
          a.out_mask = prep_mask;
          a.out_mask_cs = m->buf_cp[src];
          a.out_mask_c0 = prep_c0, a.out_mask_c1 = prep_c1;
          a.out_amax = prep_amax;
          int cfg;
          pick_tile(bp, nb, a, cfg);
          launch_conv(a, cfg, nb, s.stream);
        }
        break;
      }
      case OpKind::GMax: {
        ProfScope ps(s, "gmax_backward", 0.0, 0.0, nb);
        if (st.src_bf16)
          launch_gmax_backward_bf16(act_ptr(st.src), g_ptr(st.dst), g_ptr(st.src), nb, st.C, m->buf_cp[st.src],
                                    m->buf_cp[st.dst], m->d.bufs[st.src].S, s.stream);
        else
          launch_gmax_backward(act_ptr(st.src), g_ptr(st.dst), g_ptr(st.src), nb, st.C, m->buf_cp[st.src],
                               m->buf_cp[st.dst], m->d.bufs[st.src].S, s.stream);
        break;
      }
      case OpKind::Pool: {
        MIG_CHECK(st.pool_mode == 2, 1, "gradient of a stand-alone max pool is not supported");
        if (avg_unpool_fused(i)) break;  // the transposed conv below un-pools while it stages
        ProfScope ps(s, "unpool_avg", 0.0, 0.0, nb);
        launch_unpool_avg(g_ptr(st.dst), g_ptr(st.src), nb, st.C, m->buf_cp[st.dst], m->buf_cp[st.src],
                          m->d.bufs[st.src].S, s.stream);
        break;
      }
      default:
        throw Error(1, "gradient not supported for this model family");
    }
  }
  MIG_HIP(hipGetLastError());
  return g_ptr(m->input_dst);
}

// CNNTorchScorer::score(m, compute_gradient = true, ...) (cnn_torch_scorer.cpp:105-198): forward,
// loss.backward() and GridMaker::backward (torch_model.cpp:197-221) for B poses; ligand-atom
// gradients are averaged over the ensemble like m.scale_minus_forces(1 / cnt).
// ---- the split-fp16 range flag (Scorer::d_ovf) ----
static void h2_flag_reset(Scorer &s) {
  s.d_ovf.ensure(1);
  if (!s.h_ovf) MIG_HIP(hipHostMalloc((void **)&s.h_ovf, sizeof(unsigned), hipHostMallocDefault));
  if (s.ovf_pending) return;  // device-output calls since the last synchronize: the flag stays sticky across them
  if (s.ovf_clean) {  // (the previous call's reduction kernel cleared it: score_batch_once, out_direct)
    s.ovf_clean = false;
    return;
  }
  MIG_HIP(hipMemsetAsync(s.d_ovf.p, 0, sizeof(unsigned), s.stream));
}
// after the call's work is enqueued; the caller synchronizes the stream before reading *s.h_ovf
static void h2_flag_fetch(Scorer &s) {
  MIG_HIP(hipMemcpyAsync(s.h_ovf, s.d_ovf.p, sizeof(unsigned), hipMemcpyDeviceToHost, s.stream));
}

static void lane_tune_after_call(Scorer &s, int B, double us);

// the scorer's pinned host block for a call's outputs, at least n floats
static void ensure_pinned_out(Scorer &s, size_t n) {
  if (s.h_out4_n >= n) return;
  if (s.h_out4) (void)hipHostFree(s.h_out4);
  s.h_out4 = nullptr;
  s.h_out4_n = 0;
  MIG_HIP(hipHostMalloc((void **)&s.h_out4, n * sizeof(float), hipHostMallocDefault));
  s.h_out4_n = n;
}

static void score_batch_grad_once(Scorer &s, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                                  const float *centers, float *pose, float *aff, float *loss, float *var,
                                  float *lig_grad, unsigned flags, const float *flex_xyz = nullptr,
                                  float *flex_grad = nullptr) {
  MIG_CHECK(s.have_receptor, 4, "mi_scorer_set_receptor must be called before scoring");
  MIG_CHECK(B >= 0 && L >= 0 && (B == 0 || (lig_xyz && lig_smt)), 1, "bad ligand arguments");
  MIG_CHECK(pose && aff && loss && (lig_grad || flex_grad), 1, "output arrays must not be NULL");
  MIG_CHECK(!(flags & (MI_LIG_ON_DEVICE | MI_OUT_ON_DEVICE)), 1, "score_grad takes host pointers");
  for (Model *m : s.models)
    MIG_CHECK(m->grad_supported, 1, "gradient not supported for model " + m->d.name + ": " + m->grad_unsupported_reason);
  if (B == 0) return;
  set_call_capacity(s, B, true);
  const int nm = (int)s.models.size();
  s.d_lig.upload(lig_xyz, (size_t)B * L * 3, s.stream);
  const float *d_lig = s.d_lig.p, *d_cen = nullptr;
  if (centers) {
    s.d_centers_in.upload(centers, (size_t)B * 3, s.stream);
    d_cen = s.d_centers_in.p;
  }
  s.d_centers.ensure((size_t)B * 3);
  s.d_pose_m.ensure((size_t)nm * B);
  s.d_aff_m.ensure((size_t)nm * B);
  s.d_loss_m.ensure((size_t)nm * B);
  s.d_lig_grad.ensure((size_t)B * L * 3);
  launch_zero_u32(reinterpret_cast<unsigned *>(s.d_lig_grad.p), (size_t)B * L * 3, s.stream);  // (one launch; a memset of this size takes two)
  const int n_flex = (int)s.flex_rows.size();
  MIG_CHECK(!flex_xyz || n_flex > 0, 1, "flex coordinates given but mi_scorer_set_flex declared no flexible rows");
  MIG_CHECK(!flex_grad || flex_xyz, 1, "flex gradient without flex coordinates");
  struct FlexGuard {  // cur_flex never outlives the call
    Scorer &s;
    ~FlexGuard() { s.cur_flex = nullptr; }
  } flex_guard{s};
  if (flex_xyz) {
    s.d_flex.upload(flex_xyz, (size_t)B * n_flex * 3, s.stream);
    s.cur_flex = s.d_flex.p;
    s.d_flex_grad.ensure((size_t)B * n_flex * 3);
    MIG_HIP(hipMemsetAsync(s.d_flex_grad.p, 0, (size_t)B * n_flex * 3 * sizeof(float), s.stream));
  }
  // Lanes (as in score_batch_once): a small gradient call of an ensemble runs every model -- forward program, backward pass,
  // voxel backward -- on a stream of its own, in its own buffer set, the voxel groups side by side; every model accumulates
  // its (scaled) atom gradients into a buffer of its own and the buffers are added in model order at the end, which are the
  // additions the one-stream call makes.  gnina's default ensemble at B = 1: 1,616 -> ~? us per call.
  int lanes_max_b = 64;  // (measured: lanes win up to 64 poses per call -- B = 9: 1,137 -> 652 us, B = 64: 2,699 -> 2,414; gradient calls alike)
  if (const char *ev = option(OPT_MI_GNINA_LANES_MAX_B)) lanes_max_b = atoi(ev);
  size_t max_bufs = 0;
  bool any_overlap = false;
  for (Model *m : s.models) max_bufs = std::max(max_bufs, m->d.bufs.size()), any_overlap = any_overlap || m->overlap;
  const bool lanes = nm > 1 && B <= lanes_max_b && B <= s.cap && !s.profile && !option(OPT_MI_GNINA_NO_LANES) &&
                     (!option(OPT_MI_GNINA_LANES) || atoi(option(OPT_MI_GNINA_LANES)) != 0) && !option(OPT_MI_GNINA_NO_GRAD_LANES) &&
                     s.groups.size() <= 2 && max_bufs <= kLaneSlots && ((size_t)nm + 1) * kLaneSlots <= kPooledSlot2 && !any_overlap &&
                     s.alone_on_device;
  s.last_call_lanes = lanes;
  struct GradLaneGuard {  // run_program / run_backward launch on s.stream into buffer set s.act_lane around s.grad_pooled_slot
    Scorer &s;
    hipStream_t main;
    ~GradLaneGuard() { s.stream = main, s.act_lane = 0, s.grad_pooled_slot = kPooledSlot, s.centers_stride = 0; }
  } lane_guard{s, s.stream};
  const size_t n_lg_all = (size_t)B * L * 3, n_fg_all = (size_t)B * n_flex * 3;
  if (lanes) {
    if (s.lane_streams_offset != s.lane_offset) s.lane_streams.clear();
    s.lane_streams_offset = s.lane_offset;
    while ((int)s.lane_streams.size() < nm) s.lane_streams.push_back(device_lane_stream(s.device, ((int)s.lane_streams.size() + s.lane_offset) % 8));
    while ((int)s.lane_done.size() < nm) {
      hipEvent_t e = nullptr;
      MIG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      s.lane_done.push_back(e);
    }
    while (s.lane_start.size() < s.groups.size() + 1) {
      hipEvent_t e = nullptr;
      MIG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      s.lane_start.push_back(e);
    }
    // (every shared buffer at its final size before the first lane starts: a grow-only buffer must not move under a lane)
    s.d_raw3.ensure((size_t)3 * s.cap * (nm + 1));
    s.d_gamax.ensure((size_t)(nm + 1) * kGamaxLaneSlots * s.cap);
    s.d_ovf.ensure(1);
    s.d_lig_grad_m.ensure((size_t)nm * n_lg_all);
    launch_zero_u32(reinterpret_cast<unsigned *>(s.d_lig_grad_m.p), (size_t)nm * n_lg_all, s.stream);
    if (flex_xyz) {
      s.d_flex_grad_m.ensure((size_t)nm * n_fg_all);
      MIG_HIP(hipMemsetAsync(s.d_flex_grad_m.p, 0, (size_t)nm * n_fg_all * sizeof(float), s.stream));
    }
    s.centers_stride = (size_t)B * 3;
    s.d_centers.ensure((size_t)2 * B * 3);
    MIG_HIP(hipEventRecord(s.lane_start[s.groups.size()], s.stream));  // the call's inputs are uploaded, its accumulators cleared
  }
  int gi = -1;
  for (const VoxGroup &g : s.groups) {
    gi++;
    Model *m0 = s.models[g.first_model];
    LigSetup ls = setup_ligand(s, g, lig_smt, L);
    const BufDecl &ib = m0->d.bufs[m0->input_dst];
    const size_t pooled_n = (size_t)s.cap * ib.S * ib.S * ib.S * pooled_stride(m0);  // (one size per slot: DevBuf::ensure re-allocates when asked for more)
    for (int b0 = 0; b0 < B; b0 += s.cap) {
      const int nb = std::min(s.cap, B - b0);
      const int set = lanes ? (gi & 1) : 0;
      const size_t pslot = set ? kPooledSlot2 : kPooledSlot;
      float *pooled = act_buf(s, pslot, pooled_n);
      unsigned char *am0 = m0->input_pool == 1 ? argm_buf(s, pslot, pooled_n) : nullptr;
      hipStream_t vs = s.stream;
      if (lanes && gi > 0) {
        vs = s.lane_streams[g.first_model];
        MIG_HIP(hipStreamWaitEvent(vs, s.lane_start[s.groups.size()], 0));
      }
      if (m0->input_pool == 0)  // full-resolution grid (Overlap model): the tile kernel only writes touched voxels
        MIG_HIP(hipMemsetAsync(pooled, 0, (size_t)nb * ib.S * ib.S * ib.S * m0->buf_cp[m0->input_dst] * sizeof(float), vs));
      voxelize_chunk(s, g, ls, d_lig, L, d_cen, flags, b0, nb, m0->input_pool, pooled, am0, vs, set);
      if (lanes) MIG_HIP(hipEventRecord(s.lane_start[gi], vs));
      for (int mi : g.models) {
        Model *m = s.models[mi];
        if (lanes) {
          MIG_HIP(hipStreamWaitEvent(s.lane_streams[mi], s.lane_start[gi], 0));
          if (gi == 0) MIG_HIP(hipStreamWaitEvent(s.lane_streams[mi], s.lane_start[s.groups.size()], 0));
          s.stream = s.lane_streams[mi];
          s.act_lane = mi + 1;
        }
        s.grad_pooled_slot = pslot;
        run_program(s, mi, nb, s.d_pose_m.p + (size_t)mi * B + b0, s.d_aff_m.p + (size_t)mi * B + b0,
                    s.d_loss_m.p + (size_t)mi * B + b0, true, pslot);
        float *g0 = run_backward(s, mi, nb);
        VoxBackArgs vb{};
        vb.lig_xyz = d_lig + (size_t)b0 * L * 3;
        vb.L = L;
        vb.lig_perm = ls.perm;
        vb.lig_consts = ls.consts;
        vb.lig_chan = ls.chan;
        vb.n_lig = ls.n_lig;
        vb.centers = s.d_centers.p + (size_t)set * s.centers_stride + (size_t)b0 * 3;
        vb.grad_pooled = g0;
        vb.argmax = am0;
        vb.N = m->N;
        vb.Cp = m->overlap ? m->C : m->Cp;
        vb.res = m->d.resolution;
        vb.half_dim = m->d.dimension / 2.0f;
        vb.qa = m->qa;
        vb.qb = m->qb;
        vb.lig_grad = (lanes ? s.d_lig_grad_m.p + (size_t)mi * n_lg_all : s.d_lig_grad.p) + (size_t)b0 * L * 3;
        vb.scale = 1.0f / (float)nm;
        vb.accumulate = 1;
        vb.rot = s.cur_rot ? s.cur_rot + (size_t)b0 * 4 : nullptr;
        {
          ProfScope ps(s, "voxel_backward", 0.0, 0.0, nb);
          launch_voxel_backward(vb, nb, m->input_pool, s.stream);
        }
        TypedReceptor &tr = *s.receptors[g.rec_idx];
        if (flex_grad && tr.n_flex_typed > 0) {  // getReceptorGradient: the same backward over the flexible atoms
          vb.lig_xyz = s.cur_flex + (size_t)b0 * n_flex * 3;
          vb.L = n_flex;
          vb.lig_perm = tr.flex_perm.p;
          vb.lig_consts = tr.flex_consts.p;
          vb.lig_chan = tr.flex_chan.p;
          vb.n_lig = tr.n_flex_typed;
          vb.lig_grad = (lanes ? s.d_flex_grad_m.p + (size_t)mi * n_fg_all : s.d_flex_grad.p) + (size_t)b0 * n_flex * 3;
          ProfScope ps(s, "voxel_backward_flex", 0.0, 0.0, nb);
          launch_voxel_backward(vb, nb, m->input_pool, s.stream);
        }
        MIG_HIP(hipGetLastError());
        if (lanes) {
          MIG_HIP(hipEventRecord(s.lane_done[mi], s.stream));
          s.stream = lane_guard.main, s.act_lane = 0;
        }
      }
    }
  }
  s.grad_pooled_slot = kPooledSlot;
  if (lanes) {  // the per-model gradients, added in model order
    for (int mi = 0; mi < nm; mi++) MIG_HIP(hipStreamWaitEvent(s.stream, s.lane_done[mi], 0));
    launch_sum_models(s.d_lig_grad_m.p, nm, n_lg_all, s.d_lig_grad.p, s.stream);
    if (flex_xyz) launch_sum_models(s.d_flex_grad_m.p, nm, n_fg_all, s.d_flex_grad.p, s.stream);
  }
  s.d_pose.ensure(B);
  s.d_aff.ensure(B);
  s.d_loss.ensure(B);
  s.d_var.ensure(B);
  // Outputs through ONE pinned host block [pose | aff | loss | var | lig_grad | flex_grad]: the reduction kernel writes the
  // first four (and the range flag) itself, the gradients take one copy each -- into pinned memory, so really asynchronous --
  // and the caller's arrays are filled by memcpy.  As six hipMemcpyAsync into the caller's (pageable) arrays the tail of a
  // B = 1 gradient call was ~100 us of its 370: each such copy is staged and waited for (kernel trace, r6_b1grad.sh).
  const size_t n_lg = lig_grad ? (size_t)B * L * 3 : 0, n_fg = flex_grad ? (size_t)B * n_flex * 3 : 0;
  ensure_pinned_out(s, (size_t)4 * B + n_lg + n_fg);
  const bool out_direct = B <= 4096 && !option(OPT_MI_GNINA_OUT_COPY) && s.h_ovf && s.d_ovf.p && !s.ovf_pending;
  float *h4 = s.h_out4, *h_lg = s.h_out4 + (size_t)4 * B, *h_fg = h_lg + n_lg;
  if (out_direct) {
    launch_ensemble_reduce(s.d_pose_m.p, s.d_aff_m.p, s.d_loss_m.p, nm, B, h4, h4 + B, h4 + 2 * (size_t)B, h4 + 3 * (size_t)B, s.stream,
                           s.d_ovf.p, s.h_ovf);
  } else {
    s.d_out4.ensure((size_t)4 * B);
    float *d4 = s.d_out4.p;
    launch_ensemble_reduce(s.d_pose_m.p, s.d_aff_m.p, s.d_loss_m.p, nm, B, d4, d4 + B, d4 + 2 * (size_t)B, d4 + 3 * (size_t)B, s.stream);
    MIG_HIP(hipMemcpyAsync(h4, d4, (size_t)4 * B * sizeof(float), hipMemcpyDeviceToHost, s.stream));
    h2_flag_fetch(s);
  }
  s.last_B = B;
  if (n_lg) MIG_HIP(hipMemcpyAsync(h_lg, s.d_lig_grad.p, n_lg * sizeof(float), hipMemcpyDeviceToHost, s.stream));
  if (n_fg) MIG_HIP(hipMemcpyAsync(h_fg, s.d_flex_grad.p, n_fg * sizeof(float), hipMemcpyDeviceToHost, s.stream));
  MIG_HIP(hipStreamSynchronize(s.stream));
  s.ovf_clean = out_direct;
  memcpy(pose, h4, B * sizeof(float));
  memcpy(aff, h4 + B, B * sizeof(float));
  memcpy(loss, h4 + 2 * (size_t)B, B * sizeof(float));
  if (var) memcpy(var, h4 + 3 * (size_t)B, B * sizeof(float));
  if (n_lg) memcpy(lig_grad, h_lg, n_lg * sizeof(float));
  if (n_fg) memcpy(flex_grad, h_fg, n_fg * sizeof(float));
}

// A call whose split-fp16 kernels met an activation outside the fp16 range (|a| > 65504: a user model with large
// activations, a pathological input; NaN too) is repeated on the fp32-MFMA kernels -- the reference runs any model in fp32
// (torch_model.cpp:185).  The shipped models never get here (their activations stay below ~1e3).
static void score_batch_grad(Scorer &s, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                             const float *centers, float *pose, float *aff, float *loss, float *var,
                             float *lig_grad, unsigned flags, const float *flex_xyz = nullptr,
                             float *flex_grad = nullptr) {
  if (B <= 0) return score_batch_grad_once(s, lig_xyz, lig_smt, B, L, centers, pose, aff, loss, var, lig_grad, flags, flex_xyz, flex_grad);
  std::lock_guard<std::recursive_mutex> one_call(device_call_lock(s.device));
  s.alone_on_device = scorer_alone_on_device(s.device, &s) || (option(OPT_MI_GNINA_LANES) && atoi(option(OPT_MI_GNINA_LANES)) > 1);
  RotScope rot_scope(s, B);
  h2_flag_reset(s);
  const auto t_call = std::chrono::steady_clock::now();
  score_batch_grad_once(s, lig_xyz, lig_smt, B, L, centers, pose, aff, loss, var, lig_grad, flags, flex_xyz, flex_grad);
  lane_tune_after_call(s, B, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count());
  // (with device-output calls pending the flag is sticky and may be theirs: the repeat is then unnecessary at worst --
  // never skipped, the header promises a flagged host-output call is repeated before it returns)
  if (s.conv_path == 0 || *s.h_ovf == 0u) return;
  s.h2_fallbacks++;
  struct PathGuard {
    Scorer &s;
    int saved;
    ~PathGuard() { s.conv_path = saved; }
  } guard{s, s.conv_path};
  s.conv_path = 0;
  score_batch_grad_once(s, lig_xyz, lig_smt, B, L, centers, pose, aff, loss, var, lig_grad, flags, flex_xyz, flex_grad);
}

static void score_batch_once(Scorer &s, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                             const float *centers, float *pose, float *aff, float *loss, float *var, unsigned flags,
                             bool ragged = false) {
  MIG_CHECK(s.have_receptor, 4, "mi_scorer_set_receptor must be called before scoring");
  MIG_CHECK(B >= 0 && L >= 0 && (B == 0 || (lig_xyz && lig_smt)), 1, "bad ligand arguments");
  MIG_CHECK(pose && aff && loss, 1, "output arrays must not be NULL");
  if (B == 0) return;
  set_call_capacity(s, B, false);
  const int nm = (int)s.models.size();
  const float *d_lig = lig_xyz;
  const float *d_cen = centers;
  if (!(flags & MI_LIG_ON_DEVICE)) {
    // A small synchronous call hands its poses over in pinned host memory, which the gather kernel reads directly (it stages
    // them in LDS once): a hipMemcpyAsync from the caller's pageable array is a staged copy the call's first kernel waits for.
    // (Not for device-output calls: they return before the gather ran, and the next call would overwrite the block.)
    const size_t n_lig_f = (size_t)B * L * 3;
    if (n_lig_f <= 1536 && !(flags & MI_OUT_ON_DEVICE) && !option(OPT_MI_GNINA_LIG_COPY)) {
      if (!s.h_lig_pin) MIG_HIP(hipHostMalloc((void **)&s.h_lig_pin, 1536 * sizeof(float), hipHostMallocDefault));
      memcpy(s.h_lig_pin, lig_xyz, n_lig_f * sizeof(float));
      d_lig = s.h_lig_pin;
    } else {
      s.d_lig.upload(lig_xyz, n_lig_f, s.stream);
      d_lig = s.d_lig.p;
    }
    if (centers) {
      s.d_centers_in.upload(centers, (size_t)B * 3, s.stream);
      d_cen = s.d_centers_in.p;
    }
  }
  s.d_centers.ensure((size_t)B * 3);
  s.d_pose_m.ensure((size_t)nm * B);
  s.d_aff_m.ensure((size_t)nm * B);
  s.d_loss_m.ensure((size_t)nm * B);
  if (s.timing) {
    for (auto &e : s.ev)
      if (!e) MIG_HIP(hipEventCreate(&e));
    s.last_ms[0] = s.last_ms[1] = 0.f;
    MIG_HIP(hipEventRecord(s.ev[0], s.stream));
  }
  // Lanes (Scorer::lane_streams): a small call of an ensemble runs every model's program on its own stream.
  int lanes_max_b = 64;  // (measured: lanes win up to 64 poses per call -- B = 9: 1,137 -> 652 us, B = 64: 2,699 -> 2,414; gradient calls alike)
  if (const char *ev = option(OPT_MI_GNINA_LANES_MAX_B)) lanes_max_b = atoi(ev);
  // On by default (MI_GNINA_LANES=0 or MI_GNINA_NO_LANES=1: one stream).  Every group's grid is voxelized first (main stream),
  // then every model's program starts on its own stream: 1,800 B = 1 calls of three ensembles, every one the serial call's
  // bits (tools/experiments/lanes_diag3.py), gnina's default ensemble 1,260 -> 717 us per B = 1 call.
  const bool lanes_on = !option(OPT_MI_GNINA_LANES) || atoi(option(OPT_MI_GNINA_LANES)) != 0;
  // (at most two voxel groups: one pooled slot each)
  // (a lane's activation set is slots [64 l, 64 l + 64): a model with more buffers than that -- the generic converter emits one
  // per conv / pool output -- would run into the next lane's set, and lane 64 into kPooledSlot2: such ensembles take one stream)
  size_t max_bufs = 0;
  for (Model *m : s.models) max_bufs = std::max(max_bufs, m->d.bufs.size());
  const bool lane_slots_ok = max_bufs <= kLaneSlots && ((size_t)nm + 1) * kLaneSlots <= kPooledSlot2;
  const bool lanes = lanes_on && nm > 1 && B <= lanes_max_b && B <= s.cap && !s.profile && !option(OPT_MI_GNINA_NO_LANES) &&
                     s.groups.size() <= 2 && lane_slots_ok && s.alone_on_device;
  struct LaneJob {
    int gi;
    size_t slot;
    bool split;
  };
  std::vector<LaneJob> lane_jobs;
  s.last_call_lanes = lanes;
  if (lanes) {
    if (option(OPT_MI_GNINA_LANE_OFFSET)) s.lane_offset = atoi(option(OPT_MI_GNINA_LANE_OFFSET)) & 7;  // (fixed by hand: no tuning)
    if (s.lane_streams_offset != s.lane_offset) s.lane_streams.clear();
    s.lane_streams_offset = s.lane_offset;
    while ((int)s.lane_streams.size() < nm) s.lane_streams.push_back(device_lane_stream(s.device, ((int)s.lane_streams.size() + s.lane_offset) % 8));
    while ((int)s.lane_done.size() < nm) {
      hipEvent_t e = nullptr;
      MIG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      s.lane_done.push_back(e);
    }
    while (s.lane_start.size() < s.groups.size() + 1) {  // one per voxel group + "the call's inputs are uploaded"
      hipEvent_t e = nullptr;
      MIG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      s.lane_start.push_back(e);
    }
  }
  // the ligand's description per voxel group first (cached per group; a miss uploads on the main stream and waits)
  // (a ragged batch's per-pose arrays are one set per scorer: its groups are set up and voxelized one after the other)
  std::vector<LigSetup> lsv;
  if (!ragged)
    for (const VoxGroup &g : s.groups) lsv.push_back(setup_ligand(s, g, lig_smt, L));
  // Lanes: the groups are voxelized SIDE BY SIDE -- group 0 on the main stream, group 1 on the lane of its first model, each
  // into its own pooled slot / candidate lists / occupancy bytes / centres -- and a model's lane starts behind ITS group's grid
  // (round 6: a voxelizer may run next to conv kernels; until then every grid was voxelized before the first lane started, and
  // a B = 1 call of gnina's default ensemble spent 230 us in two gathers and two voxelizations one after the other).
  s.centers_stride = lanes ? (size_t)B * 3 : 0;
  if (lanes) {
    s.d_centers.ensure((size_t)2 * B * 3);
    MIG_HIP(hipEventRecord(s.lane_start[s.groups.size()], s.stream));
  }
  // run_program launches on s.stream into buffer set s.act_lane: a lane's slices set both and this puts them back
  struct LaneGuard {
    Scorer &s;
    hipStream_t main;
    ~LaneGuard() { s.stream = main, s.act_lane = 0; }
  } guard{s, s.stream};
  // The programs are enqueued round robin, kLaneSlice steps of each model at a time: a launch costs the host ~4.4 us, a
  // Dense program has 18 of them, and enqueued one program after the other the second Dense lane started 80 us and the
  // third lane 310 us behind the first (kernel trace, tools/experiments/r5_calls.sh 57).
  int kLaneSlice = 2;  // (MI_GNINA_LANE_SLICE: an experiment switch; a large value enqueues one program after the other)
  if (const char *ev = option(OPT_MI_GNINA_LANE_SLICE)) kLaneSlice = std::max(1, atoi(ev));
  std::vector<int> lane_lo(nm, 0);  // steps of model mi's program enqueued so far
  auto lane_slice = [&](int mi, size_t slot, bool split, int lo, int hi) {
    s.stream = s.lane_streams[mi];
    s.act_lane = mi + 1;
    run_program(s, mi, B, s.d_pose_m.p + (size_t)mi * B, s.d_aff_m.p + (size_t)mi * B, s.d_loss_m.p + (size_t)mi * B, false, slot, split, lo, hi);
    s.stream = guard.main, s.act_lane = 0;
    lane_lo[mi] = hi;
  };
  int gi = -1;
  for (const VoxGroup &g : s.groups) {
    gi++;
    Model *m0 = s.models[g.first_model];
    const LigSetup ls = ragged ? setup_ligand_ragged(s, g, lig_smt, B, L) : lsv[gi];
    const BufDecl &ib = m0->d.bufs[m0->input_dst];
    const size_t pooled_n = (size_t)s.cap * ib.S * ib.S * ib.S * pooled_stride(m0);
    // the pooled grid goes out in split format when every model of the group reads it with a split-fp16 first conv
    bool split = s.conv_path != 0 && s.precision != 1 && !option(OPT_MI_GNINA_H2_NO_SPLIT_TENSORS);
    for (int mi : g.models) split = split && s.models[mi]->pooled_split_ok;
    for (int b0 = 0; b0 < B; b0 += s.cap) {
      const int nb = std::min(s.cap, B - b0);
      const int set = lanes ? (gi & 1) : 0;
      const size_t slot = set ? kPooledSlot2 : kPooledSlot;
      float *pooled = act_buf(s, slot, pooled_n);
      hipStream_t vs = s.stream;
      if (lanes && gi > 0 && !ragged && !option(OPT_MI_GNINA_VOX_SERIAL)) {  // (MI_GNINA_VOX_SERIAL=1: the groups one after the other on the main stream)
        vs = s.lane_streams[g.first_model];
        MIG_HIP(hipStreamWaitEvent(vs, s.lane_start[s.groups.size()], 0));
      }
      if (m0->input_pool == 0)
        MIG_HIP(hipMemsetAsync(pooled, 0, (size_t)nb * ib.S * ib.S * ib.S * m0->buf_cp[m0->input_dst] * sizeof(float), vs));
      voxelize_chunk(s, g, ls, d_lig, L, d_cen, flags, b0, nb, m0->input_pool, pooled, nullptr, vs, set, split);
      if (lanes) {
        MIG_HIP(hipEventRecord(s.lane_start[gi], vs));
        lane_jobs.push_back(LaneJob{gi, slot, split});  // (B <= cap: one chunk per group)
        // the first launches of this group's programs go out BEFORE the next group's gather and voxelizer: the host is what
        // the front of a per-pose call waits for (the first conv of a Dense lane started 20 us behind its grid)
        for (int mi : g.models) {
          MIG_HIP(hipStreamWaitEvent(s.lane_streams[mi], s.lane_start[gi], 0));
          if (!s.models[mi]->overlap && !option(OPT_MI_GNINA_LANE_SLICE)) lane_slice(mi, slot, split, 0, kLaneSlice);
        }
      } else {
        for (int mi : g.models)
          run_program(s, mi, nb, s.d_pose_m.p + (size_t)mi * B + b0, s.d_aff_m.p + (size_t)mi * B + b0,
                      s.d_loss_m.p + (size_t)mi * B + b0, false, slot, split);
      }
    }
  }
  if (lanes) {
    // now every model's program on its own stream, behind its group's grid
    // (every buffer a program touches is allocated before its first launch: a grow-only buffer must not move under a lane)
    int longest = 0;
    for (const LaneJob &job : lane_jobs)
      for (int mi : s.groups[job.gi].models) {
        Model &m = *s.models[mi];
        longest = std::max(longest, m.overlap ? 1 : (int)program_steps(s, m, false).size());
      }
    for (int lo = 0; lo < longest; lo += kLaneSlice)
      for (const LaneJob &job : lane_jobs)
        for (int mi : s.groups[job.gi].models)
          if (lane_lo[mi] <= lo) lane_slice(mi, job.slot, job.split, lo, lo + kLaneSlice);
    for (int mi = 0; mi < nm; mi++) MIG_HIP(hipEventRecord(s.lane_done[mi], s.lane_streams[mi]));
    s.stream = guard.main, s.act_lane = 0;
    // the ensemble reduction (main stream) reads what the lanes wrote
    for (int mi = 0; mi < nm; mi++) MIG_HIP(hipStreamWaitEvent(s.stream, s.lane_done[mi], 0));
  }
  const bool out_dev = (flags & MI_OUT_ON_DEVICE) != 0;
  float *o_pose = pose, *o_aff = aff, *o_loss = loss, *o_var = var;
  // A host-output call: one [4][B] block in pinned host memory -> the caller's four arrays.  Small calls have the reduction
  // kernel write that block (and the range flag) itself -- the kernel's stores cross the bus instead of two copies enqueued
  // behind it, ~10 us of a per-pose call; larger ones go through a device block and one copy (MI_GNINA_OUT_COPY=1: always).
  const bool out_direct = !out_dev && B <= 4096 && !option(OPT_MI_GNINA_OUT_COPY) && s.h_ovf && s.d_ovf.p && !s.ovf_pending;
  if (!out_dev) {
    ensure_pinned_out(s, (size_t)4 * B);
    float *blk = s.h_out4;
    if (!out_direct) {
      s.d_out4.ensure((size_t)4 * B);
      blk = s.d_out4.p;
    }
    o_pose = blk, o_aff = blk + B, o_loss = blk + 2 * (size_t)B, o_var = blk + 3 * (size_t)B;
  }
  launch_ensemble_reduce(s.d_pose_m.p, s.d_aff_m.p, s.d_loss_m.p, nm, B, o_pose, o_aff, o_loss, o_var, s.stream,
                         out_direct ? s.d_ovf.p : nullptr, out_direct ? s.h_ovf : nullptr);
  if (s.timing) MIG_HIP(hipEventRecord(s.ev[2], s.stream));
  s.last_B = B;
  if (!out_dev) {
    if (!out_direct) {
      MIG_HIP(hipMemcpyAsync(s.h_out4, s.d_out4.p, (size_t)4 * B * sizeof(float), hipMemcpyDeviceToHost, s.stream));
      h2_flag_fetch(s);
    }
    MIG_HIP(hipStreamSynchronize(s.stream));
    s.ovf_clean = out_direct;
    memcpy(pose, s.h_out4, B * sizeof(float));
    memcpy(aff, s.h_out4 + B, B * sizeof(float));
    memcpy(loss, s.h_out4 + 2 * (size_t)B, B * sizeof(float));
    if (var) memcpy(var, s.h_out4 + 3 * (size_t)B, B * sizeof(float));
    if (s.timing) MIG_HIP(hipEventElapsedTime(&s.last_ms[2], s.ev[0], s.ev[2]));
  }
}

// lane-offset tuning (Scorer::lane_offset) behind a synchronous call that ran on lanes: calls of one batch size, kLaneTunePer
// per candidate assignment of models to lane streams, the fastest kept (scoring and gradient calls share the choice: whichever
// kind comes first tunes)
static void lane_tune_after_call(Scorer &s, int B, double us) {
  constexpr int kLaneTunePer = 4, kLaneCands = 8;
  if (!s.last_call_lanes || s.lane_tune_calls >= kLaneCands * kLaneTunePer || option(OPT_MI_GNINA_LANE_OFFSET) ||
      (int)s.models.size() > kLaneCands || (s.lane_tune_calls != 0 && B != s.lane_tune_B))
    return;
  s.lane_tune_B = B;
  const int k = s.lane_tune_calls % kLaneTunePer;
  if (k == 0) s.lane_tune_cur_us = 1e30;             // (the first call on a new set of streams is not counted)
  else s.lane_tune_cur_us = std::min(s.lane_tune_cur_us, us);
  if (k == kLaneTunePer - 1 && s.lane_tune_cur_us < s.lane_tune_best_us) s.lane_tune_best_us = s.lane_tune_cur_us, s.lane_tune_best = s.lane_offset;
  s.lane_tune_calls++;
  s.lane_offset = s.lane_tune_calls < kLaneCands * kLaneTunePer ? s.lane_tune_calls / kLaneTunePer : s.lane_tune_best;
}

// (see score_batch_grad.)  A device-output call (MI_OUT_ON_DEVICE) returns before its kernels ran: the flag then stays
// sticky until mi_scorer_synchronize, which reports it as MI_ERR_RANGE -- the caller repeats those calls under
// MI_PRECISION_FP32_MFMA (mi_pool_score_batch does).
static void score_batch(Scorer &s, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                        const float *centers, float *pose, float *aff, float *loss, float *var, unsigned flags,
                        bool ragged = false) {
  if (B <= 0) return score_batch_once(s, lig_xyz, lig_smt, B, L, centers, pose, aff, loss, var, flags, ragged);
  std::lock_guard<std::recursive_mutex> one_call(device_call_lock(s.device));
  s.alone_on_device = scorer_alone_on_device(s.device, &s) || (option(OPT_MI_GNINA_LANES) && atoi(option(OPT_MI_GNINA_LANES)) > 1);  // (MI_GNINA_LANES=2: lanes whatever else runs)
  RotScope rot_scope(s, B);
  h2_flag_reset(s);
  const auto t_call = std::chrono::steady_clock::now();
  score_batch_once(s, lig_xyz, lig_smt, B, L, centers, pose, aff, loss, var, flags, ragged);
  if (!(flags & MI_OUT_ON_DEVICE)) lane_tune_after_call(s, B, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count());
  if (s.conv_path == 0) return;
  if (flags & MI_OUT_ON_DEVICE) {
    s.ovf_pending = true;
    return;
  }
  if (*s.h_ovf == 0u) return;  // (see score_batch_grad: repeated also when the sticky flag may belong to a pending call)
  s.h2_fallbacks++;
  struct PathGuard {
    Scorer &s;
    int saved;
    ~PathGuard() { s.conv_path = saved; }
  } guard{s, s.conv_path};
  s.conv_path = 0;
  score_batch_once(s, lig_xyz, lig_smt, B, L, centers, pose, aff, loss, var, flags, ragged);
}

}  // namespace mig

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
using namespace mig;

#define MI_TRY try {
#define MI_CATCH_STATUS                                  \
  }                                                      \
  catch (const mig::Error &e) {                          \
    mig::set_last_error(e.what());                       \
    return e.code;                                       \
  }                                                      \
  catch (const std::exception &e) {                      \
    mig::set_last_error(e.what());                       \
    return MI_ERR_INVALID;                               \
  }
#define MI_CATCH_NULL                                    \
  }                                                      \
  catch (const std::exception &e) {                      \
    mig::set_last_error(e.what());                       \
    return nullptr;                                      \
  }

extern "C" {

int mi_gnina_abi_version(void) { return MI_GNINA_ABI_VERSION; }

const char *mi_last_error(void) { return g_last_error.c_str(); }

int mi_gnina_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

mi_status mi_gnina_init(int device) {
  MI_TRY
  // Scorers and mi_vina handles each own a HIP stream so that independent ligands overlap on the device.  The
  // HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and a queue runs its
  // kernels in order: ask for 16 unless the user chose (only effective before the first HIP call of the process).
  process_env_once();
  int n = 0;
  MIG_HIP(hipGetDeviceCount(&n));
  MIG_CHECK(n > 0, 3, "no HIP device visible");
  MIG_CHECK(device >= 0 && device < n, 1, "device index out of range");
  MIG_HIP(hipSetDevice(device));
  // The device's first three lane streams are created NOW, ahead of the streams of whatever else the process sets up (mi_vina
  // handles, pools, the host's own): which hardware queue a stream lands on is decided by the runtime from what exists at
  // that moment, and lanes created behind a few dozen short-lived streams shared queues with each other -- gnina's default
  // ensemble at B = 1 took 0.88-0.96 ms instead of 0.60 after the bench's C3 / C5 configurations had run in the process.
  for (int k = 0; k < 8; k++) (void)device_lane_stream(device, k);
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_gnina_set_option(const char *name, const char *value) {
  MI_TRY
  MIG_CHECK(name && set_option(name, value) == 0, 1, std::string("mi_gnina_set_option: no such switch: ") + (name ? name : "(null)"));
  return MI_OK;
  MI_CATCH_STATUS
}

const char *mi_gnina_options(void) { return options_summary(); }

mi_model *mi_model_load(const void *blob, size_t nbytes, const char *name) {
  MI_TRY
  ModelDesc d = parse_blob(blob, nbytes, name);
  Model *m = build_model(std::move(d));
  return reinterpret_cast<mi_model *>(m);
  MI_CATCH_NULL
}

mi_model *mi_model_load_file(const char *path) {
  MI_TRY
  MIG_CHECK(path, 1, "NULL path");
  std::ifstream f(path, std::ios::binary);
  MIG_CHECK((bool)f, 2, std::string("could not open model file ") + path);  // usage_error, cnn_torch_scorer.cpp:87
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  ModelDesc d = parse_blob(raw.data(), raw.size(), nullptr);
  return reinterpret_cast<mi_model *>(build_model(std::move(d)));
  MI_CATCH_NULL
}

mi_model *mi_model_load_file_ex(const char *path, float resolution, float dimension) {
  MI_TRY
  MIG_CHECK(path, 1, "NULL path");
  std::ifstream f(path, std::ios::binary);
  MIG_CHECK((bool)f, 2, std::string("could not open model file ") + path);
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  ModelDesc d = parse_blob(raw.data(), raw.size(), nullptr);
  regrid(d, resolution > 0 ? resolution : d.resolution, dimension > 0 ? dimension : d.dimension);
  return reinterpret_cast<mi_model *>(build_model(std::move(d)));
  MI_CATCH_NULL
}

void mi_model_retain(mi_model *m) {
  if (m) reinterpret_cast<Model *>(m)->refs++;
}

void mi_model_release(mi_model *m) {
  if (!m) return;
  Model *p = reinterpret_cast<Model *>(m);
  if (--p->refs == 0) delete p;
}

mi_status mi_model_info(const mi_model *m, float *resolution, float *dimension, int *n_rec, int *n_lig, int *N) {
  MI_TRY
  MIG_CHECK(m, 1, "NULL model");
  const Model *p = reinterpret_cast<const Model *>(m);
  if (resolution) *resolution = p->d.resolution;
  if (dimension) *dimension = p->d.dimension;
  if (n_rec) *n_rec = p->d.recmap.n_channels;
  if (n_lig) *n_lig = p->d.ligmap.n_channels;
  if (N) *N = p->N;
  return MI_OK;
  MI_CATCH_STATUS
}

const char *mi_model_name(const mi_model *m) { return m ? reinterpret_cast<const Model *>(m)->d.name.c_str() : ""; }

int mi_model_type_channel(const mi_model *m, int is_ligand, int smt, float *radius) {
  if (!m || smt < 0 || smt >= kNumSminaTypes) {
    if (radius) *radius = 0.f;
    return -1;
  }
  const Model *p = reinterpret_cast<const Model *>(m);
  if (radius) *radius = smina_xs_radius(smt);
  return (is_ligand ? p->d.ligmap : p->d.recmap).chan_of_smt[smt];
}

mi_scorer *mi_scorer_create(mi_model *const *models, int n_models) {
  MI_TRY
  MIG_CHECK(models && n_models > 0, 1, "scorer needs at least one model");
  std::unique_ptr<Scorer> s(new Scorer());
  for (int i = 0; i < n_models; i++) {
    MIG_CHECK(models[i], 1, "NULL model in ensemble");
    Model *m = reinterpret_cast<Model *>(models[i]);
    m->refs++;
    s->models.push_back(m);
  }
  MIG_HIP(hipGetDevice(&s->device));
  MIG_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  build_groups(*s);
  return reinterpret_cast<mi_scorer *>(s.release());
  MI_CATCH_NULL
}

void mi_scorer_destroy(mi_scorer *s) { delete reinterpret_cast<Scorer *>(s); }

int mi_scorer_num_models(const mi_scorer *s) { return s ? (int)reinterpret_cast<const Scorer *>(s)->models.size() : 0; }

mi_status mi_scorer_set_receptor(mi_scorer *sc, const float *xyz, const int32_t *smt, int n) {
  MI_TRY
  MIG_CHECK(sc && n >= 0 && (n == 0 || (xyz && smt)), 1, "bad receptor arguments");
  set_receptor(*reinterpret_cast<Scorer *>(sc), xyz, smt, n);
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_scorer_score_batch_ex(mi_scorer *sc, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                                   const float *centers, float *pose, float *affinity, float *loss, float *aff_var,
                                   unsigned flags) {
  MI_TRY
  MIG_CHECK(sc, 1, "NULL scorer");
  score_batch(*reinterpret_cast<Scorer *>(sc), lig_xyz, lig_smt, B, L, centers, pose, affinity, loss, aff_var, flags);
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_scorer_score_batch(mi_scorer *sc, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                                const float *centers, float *pose, float *affinity, float *loss, float *aff_var) {
  return mi_scorer_score_batch_ex(sc, lig_xyz, lig_smt, B, L, centers, pose, affinity, loss, aff_var, MI_MEM_HOST);
}

mi_status mi_scorer_score_ragged(mi_scorer *sc, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                                 const float *centers, float *pose, float *affinity, float *loss, float *aff_var) {
  MI_TRY
  MIG_CHECK(sc, 1, "NULL scorer");
  score_batch(*reinterpret_cast<Scorer *>(sc), lig_xyz, lig_smt, B, L, centers, pose, affinity, loss, aff_var,
              MI_MEM_HOST, true);
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_scorer_score_grad(mi_scorer *sc, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                               const float *centers, float *pose, float *affinity, float *loss, float *aff_var,
                               float *lig_grad) {
  MI_TRY
  MIG_CHECK(sc, 1, "NULL scorer");
  score_batch_grad(*reinterpret_cast<Scorer *>(sc), lig_xyz, lig_smt, B, L, centers, pose, affinity, loss, aff_var,
                   lig_grad, MI_MEM_HOST);
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_debug_h2_layout(const int32_t *tile_cells3, int n_mtiles, int geometry_mask, int32_t *mt_pad_y_pad_x, float *cycles) {
  MI_TRY
  MIG_CHECK(tile_cells3 && mt_pad_y_pad_x && n_mtiles > 0, 1, "bad arguments");
  const int tc[3] = {tile_cells3[0], tile_cells3[1], tile_cells3[2]};
  int mt, py, px;
  double c = 0;
  h2_choose_layout(tc, n_mtiles, geometry_mask, mt, py, px, &c);
  mt_pad_y_pad_x[0] = mt, mt_pad_y_pad_x[1] = py, mt_pad_y_pad_x[2] = px;
  if (cycles) *cycles = (float)c;
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_debug_split_f16(const float *x, int n, float scale, uint16_t *hi, uint16_t *lo, float *scale_out) {
  MI_TRY
  MIG_CHECK(n >= 0 && (n == 0 || (x && hi && lo)), 1, "bad arguments");
  const float sw = scale != 0.f ? scale : h2_weight_scale(x, (size_t)n);
  if (scale_out) *scale_out = sw;
  for (int i = 0; i < n; i++) {
    const float v = x[i] * sw;
    hi[i] = host_f2h(v);
    lo[i] = host_f2h(v - host_h2f(hi[i]));
  }
  return MI_OK;
  MI_CATCH_STATUS
}

// Diagnostic: activation buffer `buf` of model `mi` as the LAST forward call of this scorer left it, converted to fp32
// channels-last [B][S][S][S][C] on the host (a split-format buffer is decoded: value = h + l).  info = {S, C, split}.
mi_status mi_debug_read_activation(mi_scorer *sc, int mi, int buf, int B, int32_t *info, float *out, size_t out_floats) {
  MI_TRY
  MIG_CHECK(sc && info, 1, "NULL argument");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  MIG_CHECK(mi >= 0 && mi < (int)s.models.size(), 1, "model index out of range");
  Model *m = s.models[mi];
  MIG_CHECK(!m->overlap && buf >= 0 && buf < (int)m->d.bufs.size() && m->buf_cp[buf] > 0, 1, "no such activation buffer");
  const BufDecl &bd = m->d.bufs[buf];
  // (the pooled voxel grid -- buffer input_dst, slot kPooledSlot -- is readable when the call wrote it as fp32: the
  // fp32-MFMA program, or MI_GNINA_H2_NO_SPLIT_TENSORS)
  const bool pooled = buf == m->input_dst;
  MIG_CHECK(!pooled || s.conv_path == 0 || option(OPT_MI_GNINA_H2_NO_SPLIT_TENSORS), 1, "the pooled grid of this call is in the split format");
  const int cs = pooled ? pooled_stride(m) : m->buf_cp[buf];
  const bool split = !pooled && s.precision == 0 && s.conv_path != 0 && m->buf_split[buf];
  info[0] = bd.S, info[1] = bd.C, info[2] = split ? 1 : 0;
  if (!out) return MI_OK;
  const size_t S3 = (size_t)bd.S * bd.S * bd.S;
  MIG_CHECK(B >= 1 && B <= s.cap && out_floats >= (size_t)B * S3 * bd.C, 1, "bad batch / output size");
  // (a call that ran on lanes wrote this model's buffers into its lane's set)
  const size_t slot = pooled ? kPooledSlot : (size_t)buf + (size_t)(mi < (int)s.last_lane.size() ? s.last_lane[mi] : 0) * kLaneSlots;
  MIG_CHECK(slot < s.act.size() && s.act[slot] && s.act[slot]->n >= (size_t)B * S3 * cs, 2, "buffer not allocated by a forward call");
  MIG_HIP(hipStreamSynchronize(s.stream));
  std::vector<float> raw((size_t)B * S3 * cs);
  MIG_HIP(hipMemcpy(raw.data(), s.act[slot]->p, raw.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (int b = 0; b < B; b++)
    for (size_t v = 0; v < S3; v++)
      for (int c = 0; c < bd.C; c++) {
        float val;
        if (split) {  // [octet][voxel][h0..h7 | l0..l7]
          const unsigned short *rec = reinterpret_cast<const unsigned short *>(raw.data() + (size_t)b * S3 * cs) + (((size_t)(c >> 3) * S3 + v) * 16);
          val = host_h2f(rec[c & 7]) + host_h2f(rec[8 + (c & 7)]);
        } else {
          val = raw[((size_t)b * S3 + v) * cs + c];
        }
        out[((size_t)b * S3 + v) * bd.C + c] = val;
      }
  return MI_OK;
  MI_CATCH_STATUS
}

// Diagnostic: the candidate lists gather_pose_atoms left for pose 0 of the last call: info = {n_slab, cap}; counts [n_slab],
// chan [n_slab][cap], rec [n_slab][cap][8] (AtomRec as floats); pass NULL arrays to query info.
mi_status mi_debug_read_candidates(mi_scorer *sc, int32_t *info, int32_t *counts, int32_t *chan, float *rec) {
  MI_TRY
  MIG_CHECK(sc && info, 1, "NULL argument");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  info[0] = s.dbg_nslab, info[1] = s.dbg_cap;
  if (!counts) return MI_OK;
  MIG_HIP(hipStreamSynchronize(s.stream));
  const size_t n = (size_t)s.dbg_nslab * s.dbg_cap;
  MIG_HIP(hipMemcpy(counts, s.d_cand_n.p, s.dbg_nslab * sizeof(int), hipMemcpyDeviceToHost));
  if (chan) MIG_HIP(hipMemcpy(chan, s.d_cand_chan.p, n * sizeof(int), hipMemcpyDeviceToHost));
  if (rec) MIG_HIP(hipMemcpy(rec, s.d_cand.p, n * sizeof(AtomRec), hipMemcpyDeviceToHost));
  return MI_OK;
  MI_CATCH_STATUS
}

// Diagnostic (tools/experiments/vox_stress.py): gather + voxelize ONE pose `iters` times on the scorer's stream -- no
// network behind it, no device_call_lock -- and compare the pooled grid of every iteration, on the device, with the one
// this scorer produced when called with MI_STRESS_MAKE_REF (a quiet moment).  flags: 1 = make the reference and return,
// 2 = gather only before the first iteration (the candidate lists are then never rewritten while the tiles run), 4 = fill
// the grid with 0xFF bytes before every iteration (a lost store shows), 8 = hand the tile kernel a trap ring (a
// -DMI_VOX_TRAP build reports into it; trap_out [1024][16] receives it).  log [log_cap][4]: row 0 = {differing dwords,
// 0, workgroups of the compare kernel that saw one, 0}, then {iteration, dword index, got, want} per differing dword.
mi_status mi_debug_vox_stress(mi_scorer *sc, const float *lig_xyz, const int32_t *lig_smt, int L, int iters, int flags,
                              int32_t *log, int log_cap, uint32_t *trap_out) {
  MI_TRY
  MIG_CHECK(sc && lig_xyz && lig_smt && L > 0, 1, "NULL argument");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  MIG_CHECK(s.have_receptor, 4, "mi_scorer_set_receptor must be called before scoring");
  const VoxGroup &g = s.groups[0];
  Model *m0 = s.models[g.first_model];
  MIG_CHECK(m0->input_pool != 0, 1, "pooled-input models only");
  set_call_capacity(s, 1, false);
  s.d_lig.upload(lig_xyz, (size_t)L * 3, s.stream);
  s.d_centers.ensure(3);
  LigSetup ls = setup_ligand(s, g, lig_smt, L);
  const BufDecl &ib = m0->d.bufs[m0->input_dst];
  bool split = s.conv_path != 0 && s.precision != 1 && !option(OPT_MI_GNINA_H2_NO_SPLIT_TENSORS);
  for (int mi : g.models) split = split && s.models[mi]->pooled_split_ok;
  const size_t pooled_n = (size_t)s.cap * ib.S * ib.S * ib.S * pooled_stride(m0);
  float *pooled = act_buf(s, kPooledSlot, pooled_n);
  const size_t n_dw = (size_t)ib.S * ib.S * ib.S * (split ? m0->Cp8 : m0->buf_cp[m0->input_dst]);
  s.d_dbg_ref.ensure(n_dw);
  h2_flag_reset(s);
  if (flags & 1) {
    voxelize_chunk(s, g, ls, s.d_lig.p, L, nullptr, 0, 0, 1, m0->input_pool, pooled, nullptr, nullptr, 0, split);
    MIG_HIP(hipMemcpyAsync(s.d_dbg_ref.p, pooled, n_dw * 4, hipMemcpyDeviceToDevice, s.stream));
    MIG_HIP(hipStreamSynchronize(s.stream));
    return MI_OK;
  }
  MIG_CHECK(log && log_cap >= 2 && iters >= 1, 1, "bad log / iteration arguments");
  s.d_dbg_log.ensure((size_t)log_cap * 4);
  MIG_HIP(hipMemsetAsync(s.d_dbg_log.p, 0, (size_t)log_cap * 16, s.stream));
  struct TrapGuard {
    Scorer &s;
    ~TrapGuard() { s.dbg_trap = nullptr; }
  } trap_guard{s};
  if (flags & 8) {
    s.d_dbg_trap.ensure(1024 * 16);
    MIG_HIP(hipMemsetAsync(s.d_dbg_trap.p, 0, 1024 * 16 * 4, s.stream));
    s.dbg_trap = s.d_dbg_trap.p;
  }
  for (int it = 0; it < iters; it++) {
    if (flags & 4) MIG_HIP(hipMemsetAsync(pooled, 0xFF, n_dw * 4, s.stream));
    voxelize_chunk(s, g, ls, s.d_lig.p, L, nullptr, 0, 0, 1, m0->input_pool, pooled, nullptr, nullptr, 0, split, (flags & 2) && it > 0);
    launch_dword_compare(reinterpret_cast<const unsigned *>(pooled), s.d_dbg_ref.p, n_dw, it, s.d_dbg_log.p, log_cap, s.stream);
    if ((it & 63) == 63) MIG_HIP(hipStreamSynchronize(s.stream));  // (a shallow queue: the launches stay B = 1-call shaped)
  }
  MIG_HIP(hipMemcpyAsync(log, s.d_dbg_log.p, (size_t)log_cap * 16, hipMemcpyDeviceToHost, s.stream));
  if ((flags & 8) && trap_out) MIG_HIP(hipMemcpyAsync(trap_out, s.d_dbg_trap.p, 1024 * 16 * 4, hipMemcpyDeviceToHost, s.stream));
  MIG_HIP(hipStreamSynchronize(s.stream));
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_scorer_set_precision(mi_scorer *sc, int precision) {
  MI_TRY
  MIG_CHECK(sc, 1, "NULL scorer");
  MIG_CHECK(precision == MI_PRECISION_FP32 || precision == MI_PRECISION_BF16 || precision == MI_PRECISION_FP32_MFMA || precision == MI_PRECISION_FP16, 1,
            "unknown precision");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  s.precision = precision == MI_PRECISION_BF16 ? 1 : 0;
  if (precision != MI_PRECISION_BF16) s.conv_path = precision == MI_PRECISION_FP32_MFMA ? 0 : 1;
  s.h2_honly = precision == MI_PRECISION_FP16 ? 1 : 0;
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_scorer_set_rotations(mi_scorer *sc, const float *quats, int B) {
  MI_TRY
  MIG_CHECK(sc && B >= 0 && (B == 0 || quats), 1, "bad rotation arguments");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  s.next_rot.assign(quats, quats + (size_t)4 * B);
  for (int b = 0; b < B; b++) {
    const float *q = quats + 4 * b;
    const float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (!(std::fabs(n2 - 1.0f) < 1e-3f)) {
      s.next_rot.clear();
      MIG_CHECK(false, 1, "rotation quaternions must have unit length");
    }
  }
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_scorer_set_flex(mi_scorer *sc, const int32_t *rec_rows, int n_flex) {
  MI_TRY
  MIG_CHECK(sc, 1, "NULL scorer");
  set_flex(*reinterpret_cast<Scorer *>(sc), rec_rows, n_flex);
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_scorer_score_flex(mi_scorer *sc, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                               const float *centers, const float *flex_xyz, float *pose, float *affinity, float *loss,
                               float *aff_var, float *lig_grad, float *flex_grad) {
  MI_TRY
  MIG_CHECK(sc, 1, "NULL scorer");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  if (lig_grad || flex_grad) {
    score_batch_grad(s, lig_xyz, lig_smt, B, L, centers, pose, affinity, loss, aff_var, lig_grad, MI_MEM_HOST, flex_xyz,
                     flex_grad);
  } else {
    const int n_flex = (int)s.flex_rows.size();
    MIG_CHECK(!flex_xyz || n_flex > 0, 1, "flex coordinates given but mi_scorer_set_flex declared no flexible rows");
    struct FlexGuard {
      Scorer &s;
      ~FlexGuard() { s.cur_flex = nullptr; }
    } guard{s};
    if (flex_xyz && B > 0) {
      s.d_flex.upload(flex_xyz, (size_t)B * n_flex * 3, s.stream);
      s.cur_flex = s.d_flex.p;
    }
    score_batch(s, lig_xyz, lig_smt, B, L, centers, pose, affinity, loss, aff_var, MI_MEM_HOST);
  }
  return MI_OK;
  MI_CATCH_STATUS
}

int mi_scorer_flex_count(const mi_scorer *sc) {
  return sc ? (int)reinterpret_cast<const Scorer *>(sc)->flex_rows.size() : 0;
}

int mi_model_supports_gradient(const mi_model *m) {
  return m && reinterpret_cast<const Model *>(m)->grad_supported ? 1 : 0;
}

mi_status mi_scorer_last_model_outputs(mi_scorer *sc, int m, float *pose, float *affinity, float *loss, int B) {
  MI_TRY
  MIG_CHECK(sc, 1, "NULL scorer");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  MIG_CHECK(m >= 0 && m < (int)s.models.size() && B == s.last_B && B > 0, 1, "bad model index or batch size");
  MIG_HIP(hipStreamSynchronize(s.stream));
  if (pose) MIG_HIP(hipMemcpy(pose, s.d_pose_m.p + (size_t)m * B, B * sizeof(float), hipMemcpyDeviceToHost));
  if (affinity) MIG_HIP(hipMemcpy(affinity, s.d_aff_m.p + (size_t)m * B, B * sizeof(float), hipMemcpyDeviceToHost));
  if (loss) MIG_HIP(hipMemcpy(loss, s.d_loss_m.p + (size_t)m * B, B * sizeof(float), hipMemcpyDeviceToHost));
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_voxelize_batch(mi_scorer *sc, int mi, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                            const float *centers, float *grid_out, float *centers_out, unsigned flags) {
  MI_TRY
  MIG_CHECK(sc, 1, "NULL scorer");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  MIG_CHECK(s.have_receptor, 4, "mi_scorer_set_receptor must be called before voxelizing");
  MIG_CHECK(mi >= 0 && mi < (int)s.models.size(), 1, "model index out of range");
  MIG_CHECK(B >= 0 && L >= 0 && grid_out && (B == 0 || (lig_xyz && lig_smt)), 1, "bad arguments");
  if (B == 0) return MI_OK;
  std::lock_guard<std::recursive_mutex> one_call(device_call_lock(s.device));
  RotScope rot_scope(s, B);  // mi_scorer_set_rotations applies to this call too (and is consumed by it)
  const VoxGroup *grp = nullptr;
  for (auto &g : s.groups)
    for (int m : g.models)
      if (m == mi) grp = &g;
  Model *m = s.models[mi];
  const float *d_lig = lig_xyz, *d_cen = centers;
  if (!(flags & MI_LIG_ON_DEVICE)) {
    s.d_lig.upload(lig_xyz, (size_t)B * L * 3, s.stream);
    d_lig = s.d_lig.p;
    if (centers) {
      s.d_centers_in.upload(centers, (size_t)B * 3, s.stream);
      d_cen = s.d_centers_in.p;
    }
  }
  s.d_centers.ensure((size_t)B * 3);
  LigSetup ls = setup_ligand(s, *grp, lig_smt, L);
  const size_t per_pose = (size_t)m->C * m->N * m->N * m->N;
  set_call_capacity(s, B, false);
  s.cap = std::max(1, std::min(s.cap, (int)(((size_t)16 << 30) / (per_pose * 4))));  // <= 16 GB of staging
  const bool out_dev = (flags & MI_OUT_ON_DEVICE) != 0;
  DevBuf<float> tmp;
  for (int b0 = 0; b0 < B; b0 += s.cap) {
    const int nb = std::min(s.cap, B - b0);
    float *dst = out_dev ? grid_out + (size_t)b0 * per_pose : (tmp.ensure((size_t)s.cap * per_pose), tmp.p);
    MIG_HIP(hipMemsetAsync(dst, 0, (size_t)nb * per_pose * sizeof(float), s.stream));  // torch::zeros, torch_model.cpp:179
    voxelize_chunk(s, *grp, ls, d_lig, L, d_cen, flags, b0, nb, 0, dst);
    if (!out_dev)
      MIG_HIP(hipMemcpyAsync(grid_out + (size_t)b0 * per_pose, dst, (size_t)nb * per_pose * sizeof(float),
                             hipMemcpyDeviceToHost, s.stream));
  }
  if (centers_out)
    MIG_HIP(hipMemcpyAsync(centers_out, s.d_centers.p, (size_t)B * 3 * sizeof(float), hipMemcpyDeviceToHost, s.stream));
  MIG_HIP(hipStreamSynchronize(s.stream));
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_model_forward_grids(mi_scorer *sc, int mi, const float *grids, int B, float *pose, float *affinity,
                                 float *loss) {
  MI_TRY
  MIG_CHECK(sc, 1, "NULL scorer");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  std::lock_guard<std::recursive_mutex> one_call(device_call_lock(s.device));
  MIG_CHECK(mi >= 0 && mi < (int)s.models.size() && B >= 0 && grids && pose && affinity && loss, 1, "bad arguments");
  if (B == 0) return MI_OK;
  Model *m = s.models[mi];
  const size_t per_pose = (size_t)m->C * m->N * m->N * m->N;
  DevBuf<float> d_grid, d_out;
  set_call_capacity(s, B, false);
  s.cap = std::max(1, std::min(s.cap, (int)(((size_t)16 << 30) / (per_pose * 4))));
  d_grid.ensure((size_t)std::min(B, s.cap) * per_pose);
  d_out.ensure((size_t)3 * B);
  const BufDecl &ib = m->d.bufs[m->input_dst];
  // The grids are the caller's: the entry point most likely to meet values beyond the fp16 range of the split-fp16 kernels
  // (the reference's module.forward is fp32, torch_model.cpp:185) -- flag, fetch and repeat on the fp32-MFMA kernels like
  // score_batch does.
  struct PathGuard {
    Scorer &s;
    int saved;
    ~PathGuard() { s.conv_path = saved; }
  } guard{s, s.conv_path};
  for (int attempt = 0; attempt < 2; attempt++) {
    h2_flag_reset(s);
    for (int b0 = 0; b0 < B; b0 += s.cap) {
      const int nb = std::min(s.cap, B - b0);
      MIG_HIP(hipMemcpyAsync(d_grid.p, grids + (size_t)b0 * per_pose, (size_t)nb * per_pose * sizeof(float),
                             hipMemcpyHostToDevice, s.stream));
      float *pooled = act_buf(s, kPooledSlot, (size_t)s.cap * ib.S * ib.S * ib.S * pooled_stride(m));
      if (m->overlap)  // the full grid is the network input
        MIG_HIP(hipMemcpyAsync(pooled, d_grid.p, (size_t)nb * per_pose * sizeof(float), hipMemcpyDeviceToDevice, s.stream));
      else
        launch_pool_input(d_grid.p, pooled, nb, m->C, m->Cp, m->N, m->input_pool, s.stream);
      run_program(s, mi, nb, d_out.p + b0, d_out.p + B + b0, d_out.p + 2 * (size_t)B + b0);
    }
    MIG_HIP(hipMemcpyAsync(pose, d_out.p, B * sizeof(float), hipMemcpyDeviceToHost, s.stream));
    MIG_HIP(hipMemcpyAsync(affinity, d_out.p + B, B * sizeof(float), hipMemcpyDeviceToHost, s.stream));
    MIG_HIP(hipMemcpyAsync(loss, d_out.p + 2 * (size_t)B, B * sizeof(float), hipMemcpyDeviceToHost, s.stream));
    h2_flag_fetch(s);
    MIG_HIP(hipStreamSynchronize(s.stream));
    if (s.conv_path == 0 || s.precision != 0 || *s.h_ovf == 0u) break;
    s.h2_fallbacks++;
    s.conv_path = 0;  // (restored by the guard)
  }
  return MI_OK;
  MI_CATCH_STATUS
}

void *mi_scorer_stream(mi_scorer *sc) { return sc ? (void *)reinterpret_cast<Scorer *>(sc)->stream : nullptr; }

mi_status mi_scorer_synchronize(mi_scorer *sc) {
  MI_TRY
  MIG_CHECK(sc, 1, "NULL scorer");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  if (s.ovf_pending) h2_flag_fetch(s);
  MIG_HIP(hipStreamSynchronize(s.stream));
  if (s.timing && s.ev[2]) (void)hipEventElapsedTime(&s.last_ms[2], s.ev[0], s.ev[2]);
  if (s.ovf_pending) {
    s.ovf_pending = false;
    if (*s.h_ovf != 0u) {
      s.h2_fallbacks++;
      throw Error(MI_ERR_RANGE, "an activation left the fp16 range of the split-fp16 kernels in a device-output call since the last "
                                "synchronize: repeat those calls under MI_PRECISION_FP32_MFMA");
    }
  }
  return MI_OK;
  MI_CATCH_STATUS
}

int mi_scorer_h2_fallbacks(const mi_scorer *sc) { return sc ? reinterpret_cast<const Scorer *>(sc)->h2_fallbacks : 0; }

mi_status mi_scorer_set_chunk(mi_scorer *sc, int poses_per_chunk) {
  MI_TRY
  MIG_CHECK(sc && poses_per_chunk > 0 && poses_per_chunk <= 65535, 1, "chunk must be in [1, 65535]");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  MIG_HIP(hipStreamSynchronize(s.stream));
  s.chunk = poses_per_chunk;
  s.act.clear();
  s.gact.clear();
  s.argm.clear();
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_scorer_enable_profile(mi_scorer *sc, int on) {
  MI_TRY
  MIG_CHECK(sc, 1, "NULL scorer");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  MIG_HIP(hipStreamSynchronize(s.stream));
  s.profile = on != 0;
  for (auto &r : s.prof) {
    s.ev_pool.push_back(r.e0);
    s.ev_pool.push_back(r.e1);
  }
  s.prof.clear();
  return MI_OK;
  MI_CATCH_STATUS
}

const char *mi_scorer_profile_json(mi_scorer *sc) {
  MI_TRY
  MIG_CHECK(sc, 1, "NULL scorer");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  MIG_HIP(hipStreamSynchronize(s.stream));
  struct Agg {
    std::string name;
    double ms = 0, flops = 0, bytes = 0, mfma = 0;
    long launches = 0, poses = 0, counted = 0;
  };
  std::vector<unsigned long long> cnt((size_t)s.cnt_used * kMfmaCountSlots);
  if (s.cnt_used) MIG_HIP(hipMemcpy(cnt.data(), s.d_mfma_cnt.p, cnt.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  s.cnt_used = 0;
  std::vector<Agg> aggs;
  for (auto &r : s.prof) {
    float ms = 0.f;
    MIG_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
    Agg *a = nullptr;
    for (auto &x : aggs)
      if (x.name == r.name) a = &x;
    if (!a) {
      aggs.emplace_back();
      a = &aggs.back();
      a->name = r.name;
    }
    a->ms += ms;
    a->flops += r.flops;
    a->bytes += r.bytes;
    a->launches++;
    a->poses += r.poses;
    if (r.cnt_slot >= 0) {
      for (int i = 0; i < kMfmaCountSlots; i++) a->mfma += (double)cnt[(size_t)r.cnt_slot * kMfmaCountSlots + i];
      a->counted++;
    }
    s.ev_pool.push_back(r.e0);
    s.ev_pool.push_back(r.e1);
  }
  s.prof.clear();
  std::string j = "[";
  for (size_t i = 0; i < aggs.size(); i++) {
    char buf[512];
    snprintf(buf, sizeof buf,
             "%s{\"kernel\": \"%s\", \"launches\": %ld, \"poses\": %ld, \"ms_total\": %.6f, \"flops\": %.6e, "
             "\"bytes\": %.6e, \"mfma_counted_launches\": %ld, \"mfma_executed\": %.6e}",
             i ? ", " : "", aggs[i].name.c_str(), aggs[i].launches, aggs[i].poses, aggs[i].ms, aggs[i].flops,
             aggs[i].bytes, aggs[i].counted, aggs[i].mfma);
    j += buf;
  }
  j += "]";
  s.prof_json = j;
  return s.prof_json.c_str();
  MI_CATCH_NULL
}

mi_status mi_scorer_enable_timing(mi_scorer *sc, int on) {
  MI_TRY
  MIG_CHECK(sc, 1, "NULL scorer");
  reinterpret_cast<Scorer *>(sc)->timing = on != 0;
  return MI_OK;
  MI_CATCH_STATUS
}

mi_status mi_scorer_last_timing(mi_scorer *sc, float *ms3) {
  MI_TRY
  MIG_CHECK(sc && ms3, 1, "bad arguments");
  Scorer &s = *reinterpret_cast<Scorer *>(sc);
  ms3[0] = s.last_ms[0];
  ms3[1] = s.last_ms[1];
  ms3[2] = s.last_ms[2];
  return MI_OK;
  MI_CATCH_STATUS
}

}  // extern "C"
