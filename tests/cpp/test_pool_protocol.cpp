// test_pool_protocol.cpp -- mi_pool's device-path protocol (gnina_amd/csrc/pool_protocol.h) on a MOCK transport: no GPU, no
// RCCL.  The mock has the semantics that make the real thing dangerous: point-to-point operations posted inside a group
// only start at group_end, and group_end BLOCKS until every posted operation has been matched by its peer (a send without
// a receive never completes).  What the reference does with host threads (gninasrc/lib/parallel_mc.cpp:183-214) mi_pool
// does over GPUs; a rank that fails between two collectives must not leave its peers inside a group.
//
// Scenarios (each prints one line; exit code 0 = all passed):
//   ok            data arrives: every shard scattered, scored, gathered in pose order
//   alloc_fails   a rank fails in phase 0: no group is ever opened, pool usable
//   score_fails   a rank fails in phase 2: the gather is never posted, nobody is left inside a group, transport NOT abandoned
//   send_fails    a send returns an error inside the scatter group: every rank still closes its group, transport abandoned
//   hang          a rank never posts its receive: group_end blocks on rank 0; the watchdog names the phase, aborts the
//                 transport (which releases the blocked rank) and the call fails with the phase in its message
#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>

#include "../../gnina_amd/csrc/pool_protocol.h"

using namespace mig;

namespace {

struct Mock : PoolTransport, PoolWork {
  int G, B, L = 4;
  std::vector<float> lig, out[3];                  // rank 0's arrays (the caller's)
  std::vector<std::vector<float>> s_lig, s_out;    // staging of rank g
  std::vector<int> b0, nb;
  // mailbox: (src, dst, array) -> payload, delivered at group_end
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::tuple<int, int, int>, std::vector<float>> box;
  struct Op { bool send; int peer, array, first, count; };
  std::vector<std::vector<Op>> pending;            // per rank, inside its open group
  std::vector<int> depth;                          // group depth per rank (must be 0 between phases)
  std::atomic<bool> aborted{false};
  std::atomic<int> aborts{0}, allocs{0}, scores{0};
  int fail_alloc = -1, fail_score = -1, fail_send_to = -1, skip_recv_on = -1;
  Mock(int G_, int B_) : G(G_), B(B_) {
    lig.resize((size_t)B * L);
    for (size_t i = 0; i < lig.size(); i++) lig[i] = (float)i;
    for (auto &o : out) o.assign(B, -1.f);
    s_lig.resize(G), s_out.resize(G), b0.resize(G), nb.resize(G), pending.resize(G), depth.assign(G, 0);
    for (int g = 0; g < G; g++) pool_shard(B, G, g, b0[g], nb[g]);
  }
  float *at(int rank, int array, int first, size_t &per) {
    if (array == 0) return per = L, rank == 0 ? lig.data() + (size_t)first * L : s_lig[rank].data();
    per = 1;
    return rank == 0 ? out[array - 2].data() + first : s_out[rank].data() + (size_t)(array - 2) * nb[rank];
  }
  std::string group_start(int r) override { depth[r]++; return ""; }
  std::string send(int r, int peer, int array, int first, int count) override {
    if (peer == fail_send_to) return "mock: send failed";
    pending[r].push_back({true, peer, array, first, count});
    return "";
  }
  std::string recv(int r, int peer, int array, int first, int count) override {
    if (r == skip_recv_on) return "";  // (a rank that "forgets" to post: its sender blocks in group_end)
    pending[r].push_back({false, peer, array, first, count});
    return "";
  }
  std::string group_end(int r) override {
    depth[r]--;
    std::vector<Op> ops;
    ops.swap(pending[r]);
    std::unique_lock<std::mutex> l(mu);
    for (const Op &o : ops)
      if (o.send) {
        size_t per;
        float *p = at(r, o.array, o.first, per);
        box[{r, o.peer, o.array}] = std::vector<float>(p, p + (size_t)o.count * per);
      }
    cv.notify_all();
    for (const Op &o : ops) {
      if (o.send) {  // complete when the peer has taken it
        cv.wait(l, [&] { return aborted.load() || !box.count({r, o.peer, o.array}); });
      } else {
        cv.wait(l, [&] { return aborted.load() || box.count({o.peer, r, o.array}); });
        if (aborted.load()) break;
        auto it = box.find({o.peer, r, o.array});
        size_t per;
        float *p = at(r, o.array, o.first, per);
        memcpy(p, it->second.data(), it->second.size() * sizeof(float));
        box.erase(it);
        cv.notify_all();
      }
      if (aborted.load()) break;
    }
    return aborted.load() ? "mock: communicator aborted" : "";
  }
  std::string sync(int) override { return ""; }
  void abort_all() override {
    aborts++;
    aborted = true;
    std::lock_guard<std::mutex> l(mu);
    cv.notify_all();
  }
  std::string alloc(int r, int, int n) override {
    allocs++;
    if (r == fail_alloc) return "mock: allocation failed";
    if (r != 0) s_lig[r].assign((size_t)n * L, 0.f), s_out[r].assign((size_t)3 * n, 0.f);
    return "";
  }
  std::string score(int r, int first, int n) override {
    scores++;
    if (r == fail_score) return "mock: scoring failed";
    for (int i = 0; i < n; i++) {
      size_t per;
      const float *x = at(r, 0, first, per) + (size_t)(r == 0 ? i : i) * L;
      if (r == 0) x = lig.data() + (size_t)(first + i) * L;
      const float v = x[0] + x[1] + x[2] + x[3];   // "score" of pose first + i
      for (int a = 0; a < 3; a++) (r == 0 ? out[a][first + i] : s_out[r][(size_t)a * n + i]) = v * (a + 1);
    }
    return "";
  }
};

int run(const char *name, int G, int B, const std::function<void(Mock &)> &setup, const std::function<bool(Mock &, PoolOutcome, const std::string &, const std::string &)> &check,
        double watchdog = 0.0) {
  std::vector<std::unique_ptr<TaskThread>> th;
  std::vector<TaskThread *> tp;
  for (int g = 0; g < G; g++) th.emplace_back(new TaskThread()), th.back()->start(), tp.push_back(th.back().get());
  Mock m(G, B);
  setup(m);
  std::string err, phases;
  const PoolOutcome oc = pool_device_path(B, tp, m, m, 3, false, watchdog, err, &phases);
  bool in_group = false;
  for (int d : m.depth) in_group = in_group || d != 0;
  const bool ok = !in_group && check(m, oc, err, phases);
  printf("%-12s %s  outcome %d  phases [%s]  aborts %d  %s\n", name, ok ? "PASS" : "FAIL", (int)oc, phases.c_str(), m.aborts.load(), err.c_str());
  for (auto &t : th) t->join();
  return ok ? 0 : 1;
}

}  // namespace

int main() {
  int bad = 0;
  bad += run("ok", 3, 19, [](Mock &) {}, [](Mock &m, PoolOutcome oc, const std::string &err, const std::string &) {
    if (oc != PoolOutcome::ok || !err.empty() || m.aborts != 0) return false;
    for (int b = 0; b < m.B; b++) {
      const float v = m.lig[(size_t)b * 4] + m.lig[(size_t)b * 4 + 1] + m.lig[(size_t)b * 4 + 2] + m.lig[(size_t)b * 4 + 3];
      for (int a = 0; a < 3; a++)
        if (m.out[a][b] != v * (a + 1)) return false;
    }
    return true;
  });
  bad += run("ok_empty", 4, 2, [](Mock &) {}, [](Mock &m, PoolOutcome oc, const std::string &, const std::string &) {   // ranks without poses
    return oc == PoolOutcome::ok && m.out[0][0] >= 0 && m.out[0][1] >= 0;
  });
  bad += run("alloc_fails", 3, 19, [](Mock &m) { m.fail_alloc = 2; }, [](Mock &m, PoolOutcome oc, const std::string &err, const std::string &ph) {
    return oc == PoolOutcome::failed_pool_usable && ph == "alloc" && m.aborts == 0 && m.scores == 0 && err.find("allocation") != std::string::npos;
  });
  bad += run("score_fails", 3, 19, [](Mock &m) { m.fail_score = 1; }, [](Mock &m, PoolOutcome oc, const std::string &err, const std::string &ph) {
    return oc == PoolOutcome::failed_pool_usable && ph == "alloc > scatter > score" && m.aborts == 0 && m.box.empty() && err.find("scoring") != std::string::npos;
  });
  bad += run("send_fails", 3, 19, [](Mock &m) { m.fail_send_to = 2; }, [](Mock &m, PoolOutcome oc, const std::string &err, const std::string &ph) {
    // rank 2 posted a receive that is never matched: only an abort releases it -- issued at once by the rank whose send
    // failed (no watchdog configured here: the call must return by itself)
    return oc == PoolOutcome::transport_abandoned && ph == "alloc > scatter" && m.aborts == 1 && m.scores == 0 && err.find("send failed") != std::string::npos;
  }, 0.0);
  bad += run("hang", 3, 19, [](Mock &m) { m.skip_recv_on = 1; }, [](Mock &m, PoolOutcome oc, const std::string &err, const std::string &ph) {
    return oc == PoolOutcome::transport_abandoned && ph == "alloc > scatter" && m.aborts == 1 && err.find("1: scatter") != std::string::npos &&
           err.find("watchdog") != std::string::npos;
  }, 1.0);
  printf("%s\n", bad ? "FAILED" : "all scenarios passed");
  return bad ? 1 : 0;
}
