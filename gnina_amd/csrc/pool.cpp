// pool.cpp -- one node, many GPUs, ONE process: the C++ multi-device host path.
//
// What it replaces: gnina fans work out to worker threads, each with its own fresh_copy() of the scorer
// (main.cpp:1418-1442, parallel_mc.cpp:183-214).  Here the fan-out is over devices: mi_pool owns one worker thread, one
// HIP stream and one mi_scorer per GPU (weights and receptor replicated -- a few MB), a batch of poses is split into
// contiguous shards [g*B/G, (g+1)*B/G) and the shards run concurrently; nothing is reduced across poses, so the data
// path has no collective.
//
// Where the bytes travel:
//  * host buffers (the DLScorer seam's case): every worker copies its own shard H2D and its scores D2H -- each GPU
//    has its own PCIe link, so this is the fastest path and needs no GPU-to-GPU traffic at all;
//  * device-resident poses (MI_LIG_ON_DEVICE: the docking kernels produced them on devices[0]): the shards are
//    scattered from devices[0] and the scores gathered back to it over xGMI with RCCL point-to-point calls
//    (ncclSend / ncclRecv inside one group: 12 B per pose back -- launch latency, not bandwidth), so a caller that
//    keeps everything in HBM never touches the host.
// librccl.so (570 MB) is opened on demand, only when a pool of more than one device takes the device-resident path;
// single-GPU users and the host-buffer path never load it.
#include <dlfcn.h>
#include <stdlib.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>
#include <vector>

#include "../../include/mi_gnina.h"
#include "common.h"
#include "options.h"
#include "pool_protocol.h"

namespace mig {

namespace {

// the handful of RCCL entry points the pool uses, bound at run time
struct Rccl {
  void *h = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string &err) {
    if (h) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) {
      err = std::string("cannot open librccl.so: ") + dlerror();
      return false;
    }
    auto sym = [&](const char *n) { return dlsym(h, n); };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    CommAbort = reinterpret_cast<decltype(CommAbort)>(sym("ncclCommAbort"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !Broadcast || !GetErrorString) {
      err = "librccl.so lacks an expected symbol";
      return false;
    }
    return true;
  }
};

struct Worker : TaskThread {
  int device = 0, rank = 0;
  mi_scorer *scorer = nullptr;
  std::vector<mi_model *> models;
  hipStream_t stream = nullptr;  // for the RCCL scatter / gather and staging copies (the scorer has its own)
  ncclComm_t comm = nullptr;
  DevBuf<float> d_lig, d_cen, d_out;  // staging on this device for the device-resident path
  // split-fp16 range fallback of device-output calls (score_dev_out): after two flagged calls in a row this worker's scorer
  // stays on the fp32-MFMA kernels -- a model that is persistently out of range does not pay the double run on every shard
  int range_streak = 0;
  bool range_sticky = false;
};

struct Pool {
  std::vector<std::unique_ptr<Worker>> w;
  Rccl rccl;
  bool comms_ready = false;
  // transport of the device-resident path: RCCL point-to-point (default), or plain device-to-device copies
  // (hipMemcpyAsync between the devices' buffers: xGMI DMA with peer access) when MI_POOL_NO_RCCL is set, when librccl
  // cannot be initialised, or when a device is listed twice (MI_POOL_ALLOW_DUPLICATE_DEVICES=1: a test hook that lets a
  // one-GPU box run several workers and so exercise every shard / offset of both paths)
  bool use_rccl = true, duplicates = false;
  std::string rccl_note;
  long calls_host = 0, calls_device = 0;
  std::string last_phases;  // phases the last RCCL device-path call went through (mi_pool_info_json)
  std::string info;
};

// run f(worker) on every worker's thread; the first error message wins ("" = ok) -- pool_protocol.h run_on_all
std::string on_all(Pool &p, const std::function<std::string(Worker &)> &f) {
  std::vector<TaskThread *> th;
  for (auto &wk : p.w) th.push_back(wk.get());
  return run_on_all(th, "", 0.0, nullptr, [&](int r) { return f(*p.w[r]); });
}

std::string last(const char *what) { return std::string(what) + ": " + mi_last_error(); }

// One device-output call on its worker's scorer, complete on return (host or device input: `flags`).  A device-output call
// cannot repeat itself when an activation leaves the split-fp16 kernels' range (mi_scorer_synchronize reports MI_ERR_RANGE,
// include/mi_gnina.h): the call is made again on the fp32-MFMA kernels, and the scorer goes back to the precision the pool
// keeps it at (workers' scorers are never handed out: MI_PRECISION_FP32, or fp32-MFMA once the fallback turned sticky).
std::string score_dev_out(Worker &x, const float *lig, const int32_t *smt, int nb, int L, const float *cen, float *o_pose,
                          float *o_aff, float *o_loss, float *o_var, unsigned flags) {
  for (int attempt = 0; attempt < 2; attempt++) {
    if (mi_scorer_score_batch_ex(x.scorer, lig, smt, nb, L, cen, o_pose, o_aff, o_loss, o_var, flags | MI_OUT_ON_DEVICE) != MI_OK)
      return last("mi_scorer_score_batch_ex");
    const mi_status st = mi_scorer_synchronize(x.scorer);
    if (attempt == 1 && !x.range_sticky) (void)mi_scorer_set_precision(x.scorer, MI_PRECISION_FP32);
    if (st == MI_OK) {
      if (attempt == 0) x.range_streak = 0;
      return "";
    }
    if (st != MI_ERR_RANGE || attempt == 1) return last("mi_scorer_synchronize");
    if (++x.range_streak >= 2) x.range_sticky = true;
    if (mi_scorer_set_precision(x.scorer, MI_PRECISION_FP32_MFMA) != MI_OK) return last("mi_scorer_set_precision");
  }
  return "";
}
std::string score_resident(Worker &x, const float *lig, const int32_t *smt, int nb, int L, const float *cen, float *o_pose,
                           float *o_aff, float *o_loss, float *o_var) {
  return score_dev_out(x, lig, smt, nb, L, cen, o_pose, o_aff, o_loss, o_var, MI_LIG_ON_DEVICE);
}

void shard(int B, int G, int g, int &b0, int &nb) { pool_shard(B, G, g, b0, nb); }

}  // namespace
}  // namespace mig

using namespace mig;

#define PTRY try {
#define PCATCH_STATUS                 \
  }                                   \
  catch (const mig::Error &e) {       \
    mig::set_last_error(e.what());    \
    return (mi_status)e.code;         \
  }                                   \
  catch (const std::exception &e) {   \
    mig::set_last_error(e.what());    \
    return MI_ERR_DEVICE;             \
  }

extern "C" {

mi_pool *mi_pool_create(const int *devices, int n_devices, const char *const *model_paths, int n_models) {
  try {
    MIG_CHECK(devices && n_devices > 0 && model_paths && n_models > 0, 1, "bad arguments");
    const int have = mi_gnina_device_count();
    const bool allow_dup = option(OPT_MI_POOL_ALLOW_DUPLICATE_DEVICES) != nullptr;
    bool dup = false;
    for (int g = 0; g < n_devices; g++) {
      MIG_CHECK(devices[g] >= 0 && devices[g] < have, 1, "device index out of range");
      for (int k = 0; k < g; k++) {
        MIG_CHECK(devices[k] != devices[g] || allow_dup, 1, "a device is listed twice");
        dup = dup || devices[k] == devices[g];
      }
    }
    process_env_once();  // (before the worker threads exist: see common.h)
    auto p = std::make_unique<Pool>();
    p->duplicates = dup;
    p->use_rccl = !dup && option(OPT_MI_POOL_NO_RCCL) == nullptr;
    for (int g = 0; g < n_devices; g++) {
      auto wk = std::make_unique<Worker>();
      wk->device = devices[g];
      wk->rank = g;
      Worker *x = wk.get();
      x->start();
      p->w.push_back(std::move(wk));
    }
    std::vector<std::string> paths(model_paths, model_paths + n_models);
    const std::string err = on_all(*p, [&](Worker &x) -> std::string {
      if (mi_gnina_init(x.device) != MI_OK) return last("mi_gnina_init");  // hipSetDevice for this worker thread
      if (hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking) != hipSuccess) return "hipStreamCreate failed";
      for (const std::string &path : paths) {
        mi_model *m = mi_model_load_file(path.c_str());  // weights are replicated: one device-resident copy per GPU
        if (!m) return last("mi_model_load_file");
        x.models.push_back(m);
      }
      x.scorer = mi_scorer_create(x.models.data(), (int)x.models.size());
      if (!x.scorer) return last("mi_scorer_create");
      return "";
    });
    if (!err.empty()) {
      mi_pool_destroy(reinterpret_cast<mi_pool *>(p.release()));
      throw mig::Error(3, err);
    }
    return reinterpret_cast<mi_pool *>(p.release());
  } catch (const mig::Error &e) {
    mig::set_last_error(e.what());
  } catch (const std::exception &e) {
    mig::set_last_error(e.what());
  }
  return nullptr;
}

void mi_pool_destroy(mi_pool *pp) {
  if (!pp) return;
  Pool *p = reinterpret_cast<Pool *>(pp);
  (void)on_all(*p, [&](Worker &x) -> std::string {
    if (x.comm && p->rccl.CommDestroy) p->rccl.CommDestroy(x.comm);
    if (x.scorer) mi_scorer_destroy(x.scorer);
    for (mi_model *m : x.models) mi_model_release(m);
    x.d_lig.release();
    x.d_cen.release();
    x.d_out.release();
    if (x.stream) (void)hipStreamDestroy(x.stream);
    return "";
  });
  for (auto &wk : p->w) {
    wk->join();
  }
  delete p;
}

int mi_pool_size(const mi_pool *pp) { return pp ? (int)reinterpret_cast<const Pool *>(pp)->w.size() : 0; }

mi_status mi_pool_set_receptor(mi_pool *pp, const float *xyz, const int32_t *smt, int n_atoms) {
  PTRY
  MIG_CHECK(pp && xyz && smt && n_atoms > 0, 1, "bad arguments");
  Pool &p = *reinterpret_cast<Pool *>(pp);
  // replicated: every GPU types and uploads the same 16 B per atom over its own PCIe link (tens of KB -- a broadcast
  // over xGMI would only add a hop)
  const std::string err = on_all(p, [&](Worker &x) -> std::string {
    return mi_scorer_set_receptor(x.scorer, xyz, smt, n_atoms) == MI_OK ? "" : last("mi_scorer_set_receptor");
  });
  MIG_CHECK(err.empty(), 3, err);
  return MI_OK;
  PCATCH_STATUS
}

static void ensure_comms(Pool &p) {
  if (p.comms_ready) return;
  std::string err;
  MIG_CHECK(p.rccl.load(err), 3, err);
  std::vector<int> devs;
  for (auto &wk : p.w) devs.push_back(wk->device);
  std::vector<ncclComm_t> comms(devs.size());
  const ncclResult_t r = p.rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data());
  MIG_CHECK(r == ncclSuccess, 3, std::string("ncclCommInitAll: ") + p.rccl.GetErrorString(r));
  for (size_t g = 0; g < devs.size(); g++) p.w[g]->comm = comms[g];
  p.comms_ready = true;
}

mi_status mi_pool_score_batch(mi_pool *pp, const float *lig_xyz, const int32_t *lig_smt, int B, int L,
                              const float *centers, float *pose, float *affinity, float *loss, float *aff_var,
                              unsigned flags) {
  PTRY
  MIG_CHECK(pp && B >= 0 && L >= 0 && (B == 0 || (lig_xyz && lig_smt)) && pose && affinity && loss, 1, "bad arguments");
  Pool &p = *reinterpret_cast<Pool *>(pp);
  const int G = (int)p.w.size();
  const bool in_dev = (flags & MI_LIG_ON_DEVICE) != 0, out_dev = (flags & MI_OUT_ON_DEVICE) != 0;
  MIG_CHECK(in_dev == out_dev || G == 1, 1,
            "a multi-device pool takes host inputs with host outputs, or device inputs with device outputs (on devices[0])");
  if (B == 0) return MI_OK;
  if (G == 1) {  // nothing to shard: the single scorer, same bits as mi_scorer_score_batch
    const std::string err = on_all(p, [&](Worker &x) -> std::string {
      if (in_dev && out_dev) return score_resident(x, lig_xyz, lig_smt, B, L, centers, pose, affinity, loss, aff_var);
      if (out_dev) return score_dev_out(x, lig_xyz, lig_smt, B, L, centers, pose, affinity, loss, aff_var, flags);
      if (mi_scorer_score_batch_ex(x.scorer, lig_xyz, lig_smt, B, L, centers, pose, affinity, loss, aff_var, flags) != MI_OK)
        return last("mi_scorer_score_batch_ex");
      return "";
    });
    MIG_CHECK(err.empty(), 3, err);
    return MI_OK;
  }
  if (!in_dev) {  // host buffers: every worker moves its own shard over its own PCIe link
    p.calls_host++;
    const std::string err = on_all(p, [&](Worker &x) -> std::string {
      int b0, nb;
      shard(B, G, x.rank, b0, nb);
      if (nb == 0) return "";
      return mi_scorer_score_batch(x.scorer, lig_xyz + (size_t)b0 * L * 3, lig_smt, nb, L,
                                   centers ? centers + (size_t)b0 * 3 : nullptr, pose + b0, affinity + b0, loss + b0,
                                   aff_var ? aff_var + b0 : nullptr) == MI_OK
                 ? ""
                 : last("mi_scorer_score_batch");
    });
    MIG_CHECK(err.empty(), 3, err);
    return MI_OK;
  }
  // device-resident: poses / centres / outputs live on devices[0]; scatter and gather over xGMI
  p.calls_device++;
  if (p.use_rccl) {
    try {
      ensure_comms(p);
    } catch (const std::exception &e) {  // no usable librccl: the copy transport does the same job
      p.use_rccl = false;
      p.rccl_note = e.what();
    }
  }
  if (!p.use_rccl) {
    // every worker pulls its shard from devices[0] and pushes its scores back with device-to-device copies; no
    // rendezvous between workers at all
    const std::string err = on_all(p, [&](Worker &x) -> std::string {
      try {
        int b0, nb;
        shard(B, G, x.rank, b0, nb);
        if (nb == 0) return "";
        const float *my_lig = lig_xyz, *my_cen = centers;
        float *o_pose = pose, *o_aff = affinity, *o_loss = loss, *o_var = aff_var;
        if (x.rank != 0) {
          x.d_lig.ensure((size_t)nb * L * 3);
          MIG_HIP(hipMemcpyAsync(x.d_lig.p, lig_xyz + (size_t)b0 * L * 3, (size_t)nb * L * 3 * sizeof(float), hipMemcpyDefault, x.stream));
          my_lig = x.d_lig.p;
          if (centers) {
            x.d_cen.ensure((size_t)nb * 3);
            MIG_HIP(hipMemcpyAsync(x.d_cen.p, centers + (size_t)b0 * 3, (size_t)nb * 3 * sizeof(float), hipMemcpyDefault, x.stream));
            my_cen = x.d_cen.p;
          }
          MIG_HIP(hipStreamSynchronize(x.stream));
          x.d_out.ensure((size_t)4 * nb);
          o_pose = x.d_out.p, o_aff = o_pose + nb, o_loss = o_aff + nb, o_var = aff_var ? o_loss + nb : nullptr;
        }
        const std::string se = score_resident(x, my_lig, lig_smt, nb, L, my_cen, o_pose, o_aff, o_loss, o_var);
        if (!se.empty()) return se;
        if (x.rank != 0) {
          float *dst[4] = {pose, affinity, loss, aff_var};
          for (int a = 0; a < (aff_var ? 4 : 3); a++)
            MIG_HIP(hipMemcpyAsync(dst[a] + b0, x.d_out.p + (size_t)a * nb, (size_t)nb * sizeof(float), hipMemcpyDefault, x.stream));
          MIG_HIP(hipStreamSynchronize(x.stream));
        }
        return "";
      } catch (const std::exception &ex) {
        return ex.what();
      }
    });
    MIG_CHECK(err.empty(), 3, err);
    return MI_OK;
  }
  // RCCL transport: the four phases of pool_protocol.h (allocations | scatter group | scoring | gather group, host-side
  // rendezvous between them, every rank closes every group it opened, a watchdog on the two transport phases)
  Rccl &R = p.rccl;
  struct RcclTransport : PoolTransport {
    Pool &p;
    Rccl &R;
    const float *lig_xyz, *centers;
    float *out[4];
    int L;
    RcclTransport(Pool &p_, Rccl &R_) : p(p_), R(R_) {}
    std::string nc(ncclResult_t r, const char *what) { return r == ncclSuccess ? "" : std::string(what) + ": " + R.GetErrorString(r); }
    // array -> (pointer of pose `first` on this rank, floats per pose): rank 0 addresses the caller's arrays, rank g its staging
    void *at(int rank, int array, int first, int b0, size_t &per) {
      Worker &x = *p.w[rank];
      if (array == 0) return per = (size_t)L * 3, rank == 0 ? (void *)(lig_xyz + (size_t)first * L * 3) : (void *)x.d_lig.p;
      if (array == 1) return per = 3, rank == 0 ? (void *)(centers + (size_t)first * 3) : (void *)x.d_cen.p;
      per = 1;
      return rank == 0 ? (void *)(out[array - 2] + first) : (void *)(x.d_out.p + (size_t)(array - 2) * nb_of[rank]);
    }
    std::vector<int> nb_of;
    std::string group_start(int) override { return nc(R.GroupStart(), "ncclGroupStart"); }
    std::string group_end(int) override { return nc(R.GroupEnd(), "ncclGroupEnd"); }
    std::string send(int rank, int peer, int array, int first, int count) override {
      size_t per;
      void *ptr = at(rank, array, first, 0, per);
      Worker &x = *p.w[rank];
      return nc(R.Send(ptr, (size_t)count * per, ncclFloat, peer, x.comm, x.stream), "ncclSend");
    }
    std::string recv(int rank, int peer, int array, int first, int count) override {
      size_t per;
      void *ptr = at(rank, array, first, 0, per);
      Worker &x = *p.w[rank];
      return nc(R.Recv(ptr, (size_t)count * per, ncclFloat, peer, x.comm, x.stream), "ncclRecv");
    }
    std::string sync(int rank) override {
      return hipStreamSynchronize(p.w[rank]->stream) == hipSuccess ? "" : "hipStreamSynchronize failed after the transfers";
    }
    void abort_all() override {  // (any thread, once: ncclCommAbort is what unblocks a rank stuck inside a group; the handles
                                 // stay in place -- their own threads may be using them -- and are dropped after the call)
      for (auto &wk : p.w)
        if (wk->comm && R.CommAbort) (void)R.CommAbort(wk->comm);
    }
  } T(p, R);
  T.lig_xyz = lig_xyz, T.centers = centers, T.L = L;
  T.out[0] = pose, T.out[1] = affinity, T.out[2] = loss, T.out[3] = aff_var;
  T.nb_of.resize(G);
  for (int g = 0; g < G; g++) {
    int b0;
    shard(B, G, g, b0, T.nb_of[g]);
  }
  struct Work : PoolWork {
    Pool &p;
    const float *lig_xyz, *centers;
    const int32_t *lig_smt;
    float *pose, *affinity, *loss, *aff_var;
    int L;
    Work(Pool &p_) : p(p_) {}
    std::string alloc(int rank, int, int nb) override {
      Worker &x = *p.w[rank];
      if (rank != 0 && nb > 0) {
        x.d_lig.ensure((size_t)nb * L * 3);
        if (centers) x.d_cen.ensure((size_t)nb * 3);
        x.d_out.ensure((size_t)4 * nb);
      }
      return "";
    }
    std::string score(int rank, int, int nb) override {
      Worker &x = *p.w[rank];
      if (rank == 0) return score_resident(x, lig_xyz, lig_smt, nb, L, centers, pose, affinity, loss, aff_var);
      float *o = x.d_out.p;
      return score_resident(x, x.d_lig.p, lig_smt, nb, L, centers ? x.d_cen.p : nullptr, o, o + nb, o + 2 * (size_t)nb, aff_var ? o + 3 * (size_t)nb : nullptr);
    }
  } W(p);
  W.lig_xyz = lig_xyz, W.centers = centers, W.lig_smt = lig_smt, W.pose = pose, W.affinity = affinity, W.loss = loss, W.aff_var = aff_var, W.L = L;
  std::vector<TaskThread *> threads;
  for (auto &wk : p.w) threads.push_back(wk.get());
  // (MI_POOL_WATCHDOG_S: seconds a transport phase may take before its communicators are aborted; 0 = never)
  double watchdog_s = 120.0;
  if (const char *ev = option(OPT_MI_POOL_WATCHDOG_S)) watchdog_s = atof(ev);
  std::string err, phases;
  const PoolOutcome oc = pool_device_path(B, threads, T, W, aff_var ? 4 : 3, centers != nullptr, watchdog_s, err, &phases);
  p.last_phases = phases;
  if (oc == PoolOutcome::transport_abandoned) {  // later calls take the copy transport
    for (auto &wk : p.w) wk->comm = nullptr;  // (aborted by the protocol)
    p.comms_ready = false;
    p.use_rccl = false;
    p.rccl_note = err;
  }
  MIG_CHECK(oc == PoolOutcome::ok, 3, err);
  return MI_OK;
  PCATCH_STATUS
}

// The virtual-screen seam (config C4): B poses of DIFFERENT ligands, rows padded to Lmax with smt = -1
// (mi_scorer_score_ragged's layout), host buffers; contiguous pose shards, results in input order.
mi_status mi_pool_score_ragged(mi_pool *pp, const float *lig_xyz, const int32_t *lig_smt, int B, int Lmax,
                               const float *centers, float *pose, float *affinity, float *loss, float *aff_var) {
  PTRY
  MIG_CHECK(pp && B >= 0 && Lmax >= 0 && (B == 0 || (lig_xyz && lig_smt)) && pose && affinity && loss, 1, "bad arguments");
  Pool &p = *reinterpret_cast<Pool *>(pp);
  const int G = (int)p.w.size();
  if (B == 0) return MI_OK;
  p.calls_host++;
  const std::string err = on_all(p, [&](Worker &x) -> std::string {
    int b0, nb;
    shard(B, G, x.rank, b0, nb);
    if (nb == 0) return "";
    return mi_scorer_score_ragged(x.scorer, lig_xyz + (size_t)b0 * Lmax * 3, lig_smt + (size_t)b0 * Lmax, nb, Lmax,
                                  centers ? centers + (size_t)b0 * 3 : nullptr, pose + b0, affinity + b0, loss + b0,
                                  aff_var ? aff_var + b0 : nullptr) == MI_OK
               ? ""
               : last("mi_scorer_score_ragged");
  });
  MIG_CHECK(err.empty(), 3, err);
  return MI_OK;
  PCATCH_STATUS
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// mi_vina_pool: the Vina / Monte-Carlo half of the path over the node's GPUs.  gnina runs the `exhaustiveness` chains of a
// dock as tasks of a thread pool, every task with its own copy of the model and one seed (parallel_mc.cpp:183-214), and
// merges the per-chain containers afterwards; chains never exchange anything.  Here the tasks go to devices: every GPU
// holds its own mi_vina handle -- pair tables, receptor, cache grids, ligand(s): all of it set up by the SAME calls on every
// device (mi_vina_pool_configure runs a caller-supplied function once per handle, on that device's worker thread) -- and
// a launch of B chains is split by chain id into contiguous shards (a screen: by ligand, round robin, so that a ligand's
// chains share a device).  A chain is a function of its seed and the handle's state only: the pool's outputs equal a
// single handle's bit for bit, whatever the number of devices.  No collective: per-chain containers come back to the
// host arrays of the caller, who merges them (mi_merge_mc_outputs) as parallel_mc.cpp:165-181 does.
// ---------------------------------------------------------------------------------------------------------------------
namespace mig {
namespace {
struct VinaPool {
  std::vector<std::unique_ptr<Worker>> w;  // (scorer / models / comm unused)
  std::vector<mi_vina *> h;
  std::string info;
};
std::string on_all_vina(VinaPool &p, const std::function<std::string(Worker &, mi_vina *)> &f) {
  std::vector<std::future<std::string>> futs;
  for (size_t g = 0; g < p.w.size(); g++) {
    Worker *x = p.w[g].get();
    mi_vina *hv = p.h[g];
    futs.push_back(x->post([x, hv, &f]() -> std::string {
      try {
        return f(*x, hv);
      } catch (const std::exception &e) {
        return std::string("worker ") + std::to_string(x->rank) + ": " + e.what();
      } catch (...) {
        return std::string("worker ") + std::to_string(x->rank) + ": unknown exception";
      }
    }));
  }
  std::string err;
  for (auto &fu : futs) {
    std::string e = fu.get();
    if (err.empty() && !e.empty()) err = e;
  }
  return err;
}
}  // namespace
}  // namespace mig

extern "C" {

mi_vina_pool *mi_vina_pool_create(const int *devices, int n_devices, const float *weights5, float cutoff, float factor) {
  try {
    MIG_CHECK(devices && n_devices > 0, 1, "bad arguments");
    const int have = mi_gnina_device_count();
    const bool allow_dup = option(OPT_MI_POOL_ALLOW_DUPLICATE_DEVICES) != nullptr;
    for (int g = 0; g < n_devices; g++) {
      MIG_CHECK(devices[g] >= 0 && devices[g] < have, 1, "device index out of range");
      for (int k = 0; k < g; k++) MIG_CHECK(devices[k] != devices[g] || allow_dup, 1, "a device is listed twice");
    }
    process_env_once();
    auto p = std::make_unique<VinaPool>();
    for (int g = 0; g < n_devices; g++) {
      auto wk = std::make_unique<Worker>();
      wk->device = devices[g];
      wk->rank = g;
      Worker *x = wk.get();
      x->start();
      p->w.push_back(std::move(wk));
      p->h.push_back(nullptr);
    }
    std::vector<float> w5;
    if (weights5) w5.assign(weights5, weights5 + 5);
    std::vector<std::future<std::string>> futs;
    for (int g = 0; g < n_devices; g++) {
      Worker *x = p->w[g].get();
      mi_vina **slot = &p->h[g];
      futs.push_back(x->post([x, slot, &w5, cutoff, factor]() -> std::string {
        if (mi_gnina_init(x->device) != MI_OK) return last("mi_gnina_init");
        *slot = mi_vina_create(w5.empty() ? nullptr : w5.data(), cutoff, factor);
        return *slot ? "" : last("mi_vina_create");
      }));
    }
    std::string err;
    for (auto &fu : futs) {
      const std::string e = fu.get();
      if (err.empty() && !e.empty()) err = e;
    }
    if (!err.empty()) {
      mi_vina_pool_destroy(reinterpret_cast<mi_vina_pool *>(p.release()));
      throw mig::Error(3, err);
    }
    return reinterpret_cast<mi_vina_pool *>(p.release());
  } catch (const std::exception &e) {
    mig::set_last_error(e.what());
  }
  return nullptr;
}

void mi_vina_pool_destroy(mi_vina_pool *pp) {
  if (!pp) return;
  VinaPool *p = reinterpret_cast<VinaPool *>(pp);
  (void)on_all_vina(*p, [](Worker &, mi_vina *hv) -> std::string {
    if (hv) mi_vina_destroy(hv);
    return "";
  });
  for (auto &wk : p->w) {
    wk->join();
  }
  delete p;
}

int mi_vina_pool_size(const mi_vina_pool *pp) { return pp ? (int)reinterpret_cast<const VinaPool *>(pp)->w.size() : 0; }

mi_status mi_vina_pool_configure(mi_vina_pool *pp, mi_vina_pool_fn fn, void *user) {
  PTRY
  MIG_CHECK(pp && fn, 1, "bad arguments");
  VinaPool &p = *reinterpret_cast<VinaPool *>(pp);
  const std::string err = on_all_vina(p, [&](Worker &x, mi_vina *hv) -> std::string {
    return fn(hv, x.rank, user) == MI_OK ? "" : last("mi_vina_pool_configure callback");
  });
  MIG_CHECK(err.empty(), 3, err);
  return MI_OK;
  PCATCH_STATUS
}

mi_status mi_vina_pool_mc_batch(mi_vina_pool *pp, int B, const uint64_t *seeds, const float *corner1, const float *corner2,
                                const mi_mc_params *params, int conf_size, int n_heavy, int32_t *out_n, float *out_e,
                                float *out_conf, float *out_coords, int32_t *evals) {
  PTRY
  MIG_CHECK(pp && B >= 0 && (B == 0 || (seeds && corner1 && corner2 && params && out_n && out_e && out_conf && out_coords)) &&
                conf_size > 0 && n_heavy >= 0, 1, "bad arguments");
  VinaPool &p = *reinterpret_cast<VinaPool *>(pp);
  const int G = (int)p.w.size();
  if (B == 0) return MI_OK;
  const size_t ns = (size_t)params->num_saved;
  const std::string err = on_all_vina(p, [&](Worker &x, mi_vina *hv) -> std::string {
    int b0, nb;
    shard(B, G, x.rank, b0, nb);  // chains [b0, b0 + nb): split by chain id
    if (nb == 0) return "";
    return mi_vina_mc_batch(hv, nb, seeds + b0, corner1, corner2, params, out_n + b0, out_e + (size_t)b0 * ns,
                            out_conf + (size_t)b0 * ns * conf_size, out_coords + (size_t)b0 * ns * n_heavy * 3,
                            evals ? evals + b0 : nullptr) == MI_OK
               ? ""
               : last("mi_vina_mc_batch");
  });
  MIG_CHECK(err.empty(), 3, err);
  return MI_OK;
  PCATCH_STATUS
}

mi_status mi_vina_pool_mc_screen(mi_vina_pool *pp, int B, const int32_t *chain_ligand, const uint64_t *seeds,
                                 const float *corner1, const float *corner2, const mi_mc_params *params, int max_conf,
                                 int max_heavy, int32_t *out_n, float *out_e, float *out_conf, float *out_coords,
                                 int32_t *evals) {
  PTRY
  MIG_CHECK(pp && B >= 0 && (B == 0 || (chain_ligand && seeds && corner1 && corner2 && params && out_n && out_e && out_conf && out_coords)) &&
                max_conf > 0 && max_heavy >= 0, 1, "bad arguments");
  VinaPool &p = *reinterpret_cast<VinaPool *>(pp);
  const int G = (int)p.w.size();
  if (B == 0) return MI_OK;
  {  // the single-handle call's argument check, ahead of the fan-out: a negative id would match no rank (its chain silently
     // skipped), an id past the screen would fail on one worker after the others did their work
    const int n_screen = mi_vina_screen_size(p.h.empty() ? nullptr : p.h[0]);
    for (int b = 0; b < B; b++)
      MIG_CHECK(chain_ligand[b] >= 0 && chain_ligand[b] < n_screen, 1, "mi_vina_pool_mc_screen: chain_ligand[" + std::to_string(b) + "] = " +
                                                                         std::to_string(chain_ligand[b]) + " is not a ligand of the screen (" + std::to_string(n_screen) + ")");
  }
  const size_t ns = (size_t)params[0].num_saved, cs = ns * max_conf, xs = ns * (size_t)max_heavy * 3;
  const std::string err = on_all_vina(p, [&](Worker &x, mi_vina *hv) -> std::string {
    // ligand l goes to device l % G: this device's chains, in launch order
    std::vector<int> idx;
    for (int b = 0; b < B; b++)
      if (chain_ligand[b] % G == x.rank) idx.push_back(b);
    const int nb = (int)idx.size();
    if (nb == 0) return "";
    std::vector<int32_t> cl(nb), on(nb), ev(nb);
    std::vector<uint64_t> sd(nb);
    std::vector<float> oe((size_t)nb * ns), oc((size_t)nb * cs), ox((size_t)nb * xs);
    for (int i = 0; i < nb; i++) cl[i] = chain_ligand[idx[i]], sd[i] = seeds[idx[i]];
    if (mi_vina_mc_screen(hv, nb, cl.data(), sd.data(), corner1, corner2, params, on.data(), oe.data(), oc.data(), ox.data(),
                          ev.data()) != MI_OK)
      return last("mi_vina_mc_screen");
    for (int i = 0; i < nb; i++) {  // (disjoint rows of the caller's arrays: no two workers write the same chain)
      const size_t b = (size_t)idx[i];
      out_n[b] = on[i];
      if (evals) evals[b] = ev[i];
      std::memcpy(out_e + b * ns, oe.data() + (size_t)i * ns, ns * sizeof(float));
      std::memcpy(out_conf + b * cs, oc.data() + (size_t)i * cs, cs * sizeof(float));
      std::memcpy(out_coords + b * xs, ox.data() + (size_t)i * xs, xs * sizeof(float));
    }
    return "";
  });
  MIG_CHECK(err.empty(), 3, err);
  return MI_OK;
  PCATCH_STATUS
}

const char *mi_vina_pool_info_json(mi_vina_pool *pp) {
  if (!pp) return "{}";
  VinaPool &p = *reinterpret_cast<VinaPool *>(pp);
  std::ostringstream o;
  o << "{\"devices\": [";
  for (size_t g = 0; g < p.w.size(); g++) o << (g ? ", " : "") << p.w[g]->device;
  o << "], \"ranks\": " << p.w.size() << ", \"sharding\": \"chains by chain id (contiguous); screens by ligand, round robin\"}";
  p.info = o.str();
  return p.info.c_str();
}

const char *mi_pool_info_json(mi_pool *pp) {
  if (!pp) return "{}";
  Pool &p = *reinterpret_cast<Pool *>(pp);
  std::ostringstream o;
  o << "{\"devices\": [";
  for (size_t g = 0; g < p.w.size(); g++) o << (g ? ", " : "") << p.w[g]->device;
  o << "], \"ranks\": " << p.w.size() << ", \"device_path_transport\": \"" << (p.use_rccl ? "rccl" : "copies") << "\""
    << ", \"rccl_loaded\": " << (p.rccl.h ? "true" : "false")
    << ", \"rccl_comms\": " << (p.comms_ready ? "true" : "false") << ", \"calls_host_path\": " << p.calls_host
    << ", \"calls_device_path\": " << p.calls_device << ", \"last_device_path_phases\": \"" << p.last_phases << "\""
    << ", \"rccl_note\": \"" << [&] {
         std::string n = p.rccl_note;
         for (char &c : n)
           if (c == '"' || c == '\\' || c == '\n') c = ' ';
         return n;
       }() << "\"}";
  p.info = o.str();
  return p.info.c_str();
}

}  // extern "C"
