// pool_protocol.h -- the host-side protocol of mi_pool's device-resident path, free of HIP and RCCL types.
//
// gnina fans its work out over host threads (gninasrc/lib/parallel_mc.cpp:183-214: one task per Monte-Carlo chain;
// gninasrc/main/main.cpp:1418-1442: one worker per ligand); mi_pool fans pose shards out over GPUs, one host thread per
// device, and -- when poses and outputs live on devices[0] -- moves the shards with RCCL point-to-point calls.  What can
// go wrong there is host logic: a rank that fails between two collectives leaves its peers inside an ncclGroupEnd waiting
// for a transfer that is never posted.  The rules that prevent it are stated ONCE, here, against two small interfaces,
// so that they can be driven without a GPU (tests/cpp/test_pool_protocol.cpp runs them on a mock transport with failing
// and hanging ranks) and with the real thing (pool.cpp):
//
//   phase 0  allocations                  (may fail: no RCCL call has been made)
//   phase 1  scatter   group { sends | receives }   every rank opens AND closes its group, whatever failed in between
//   phase 2  scoring                      (may fail: nothing is pending in the transport; the pool stays usable)
//   phase 3  gather    group { receives | sends }   posted only once ALL ranks have scored
//
// A phase runs on every rank's own thread and the caller waits for all of them (host-side rendezvous between phases).
// A failed group makes the communicators unusable: they are aborted -- at once, by the rank whose post failed: a peer may
// already be blocked on the operation that will now never be matched -- and the pool falls back to its copy transport.  A
// phase that does not return within the watchdog period is a hang inside the transport: the watchdog names the phase
// and the ranks still inside it, aborts the communicators (ncclCommAbort is the documented way out of a blocked RCCL call)
// and keeps waiting for the threads to come back with their errors.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mig {

// one worker thread with a task queue (mi_pool / mi_vina_pool: one per device; tasks report through their string)
struct TaskThread {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::packaged_task<std::string()>> q;
  bool stop = false;
  void loop() {
    for (;;) {
      std::packaged_task<std::string()> t;
      {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return stop || !q.empty(); });
        if (q.empty()) return;
        t = std::move(q.front());
        q.pop_front();
      }
      t();
    }
  }
  void start() {
    th = std::thread([this] { loop(); });
  }
  void join() {
    {
      std::lock_guard<std::mutex> l(mu);
      stop = true;
    }
    cv.notify_one();
    if (th.joinable()) th.join();
  }
  std::future<std::string> post(std::function<std::string()> f) {
    std::packaged_task<std::string()> t(std::move(f));
    auto fut = t.get_future();
    {
      std::lock_guard<std::mutex> l(mu);
      q.push_back(std::move(t));
    }
    cv.notify_one();
    return fut;
  }
};

// Run f(rank) on every rank's thread and wait for all of them; the first error text wins ("" = ok).  watchdog_s > 0: when
// a rank has not returned after that many seconds, report the phase and the missing ranks on stderr, call on_hang (once)
// and go on waiting -- the threads cannot be abandoned, they run on references to the caller's frame.
inline std::string run_on_all(const std::vector<TaskThread *> &threads, const char *phase, double watchdog_s,
                              const std::function<void()> &on_hang, const std::function<std::string(int)> &f, bool *hung = nullptr) {
  std::vector<std::future<std::string>> futs;
  for (size_t r = 0; r < threads.size(); r++) {
    const int rank = (int)r;
    // (a task that throws would leave the other workers running on references to a frame that is being unwound)
    futs.push_back(threads[r]->post([rank, &f]() -> std::string {
      try {
        return f(rank);
      } catch (const std::exception &e) {
        return std::string("worker ") + std::to_string(rank) + ": " + e.what();
      } catch (...) {
        return std::string("worker ") + std::to_string(rank) + ": unknown exception";
      }
    }));
  }
  bool fired = false;
  if (watchdog_s > 0) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(watchdog_s);
    std::string missing;
    for (size_t r = 0; r < futs.size(); r++)
      if (futs[r].wait_until(deadline) != std::future_status::ready) missing += (missing.empty() ? "" : ", ") + std::to_string(r);
    if (!missing.empty()) {
      fired = true;
      fprintf(stderr, "mi_pool watchdog: phase '%s' has not returned on rank(s) %s after %.0f s -- aborting the transport\n", phase,
              missing.c_str(), watchdog_s);
      fflush(stderr);
      if (on_hang) on_hang();
    }
  }
  std::string err;
  for (auto &fu : futs) {
    std::string e = fu.get();
    if (err.empty() && !e.empty()) err = e;
  }
  if (hung) *hung = fired;
  if (fired) err = std::string("phase '") + phase + "' hung (watchdog)" + (err.empty() ? "" : ": " + err);
  return err;
}

// what the device-resident path needs from its transport; every call is made on `rank`'s own thread.  Arrays: 0 = ligand
// coordinates, 1 = grid centres (inputs, scattered from rank 0), 2 + a = output array a (gathered on rank 0); `first`,
// `count` are in POSES.  "" = ok.
struct PoolTransport {
  virtual ~PoolTransport() {}
  virtual std::string group_start(int rank) = 0;
  virtual std::string group_end(int rank) = 0;
  virtual std::string send(int rank, int peer, int array, int first, int count) = 0;
  virtual std::string recv(int rank, int peer, int array, int first, int count) = 0;
  virtual std::string sync(int rank) = 0;  // the transfers this rank posted are complete
  virtual void abort_all() = 0;            // the communicators are unusable (also the watchdog's way out of a hang)
};

struct PoolWork {
  virtual ~PoolWork() {}
  virtual std::string alloc(int rank, int first, int count) = 0;  // staging buffers for this rank's shard
  virtual std::string score(int rank, int first, int count) = 0;  // device in, device out; no transport call inside
};

inline void pool_shard(int B, int G, int g, int &b0, int &nb) {  // contiguous [g*B/G, (g+1)*B/G), SURVEY 8e
  b0 = (int)((long)B * g / G);
  nb = (int)((long)B * (g + 1) / G) - b0;
}

enum class PoolOutcome { ok, failed_pool_usable, transport_abandoned };

// The four phases.  n_out = output arrays to gather (3 or 4), centres = array 1 exists.
inline PoolOutcome pool_device_path(int B, const std::vector<TaskThread *> &threads, PoolTransport &T, PoolWork &W, int n_out, bool centres,
                                    double watchdog_s, std::string &err, std::string *phase_log = nullptr) {
  const int G = (int)threads.size();
  auto note = [&](const char *ph) {
    if (phase_log) *phase_log += std::string(phase_log->empty() ? "" : " > ") + ph;
  };
  std::atomic<bool> aborted{false};
  auto abort_once = [&] {  // (any thread: a rank whose post failed, the watchdog, the caller)
    if (!aborted.exchange(true)) T.abort_all();
  };
  bool hung = false;
  note("alloc");
  err = run_on_all(threads, "0: allocations", 0.0, nullptr, [&](int r) {
    int b0, nb;
    pool_shard(B, G, r, b0, nb);
    return W.alloc(r, b0, nb);
  }, &hung);
  if (!err.empty()) return PoolOutcome::failed_pool_usable;
  // one group per rank = one fused transfer set; nothing but the point-to-point calls between start and end, and the end
  // is reached whatever they returned
  auto grouped = [&](const char *phase, bool scatter) {
    return run_on_all(threads, phase, watchdog_s, abort_once, [&](int r) -> std::string {
      int b0, nb;
      pool_shard(B, G, r, b0, nb);
      std::string e = T.group_start(r);
      if (!e.empty()) return e;
      if (r == 0) {
        for (int g = 1; g < G; g++) {
          int c0, cn;
          pool_shard(B, G, g, c0, cn);
          if (cn == 0) continue;
          if (scatter) {
            if (e.empty()) e = T.send(0, g, 0, c0, cn);
            if (e.empty() && centres) e = T.send(0, g, 1, c0, cn);
          } else {
            for (int a = 0; a < n_out; a++)
              if (e.empty()) e = T.recv(0, g, 2 + a, c0, cn);
          }
        }
      } else if (nb > 0) {
        if (scatter) {
          e = T.recv(r, 0, 0, b0, nb);
          if (e.empty() && centres) e = T.recv(r, 0, 1, b0, nb);
        } else {
          for (int a = 0; a < n_out; a++)
            if (e.empty()) e = T.send(r, 0, 2 + a, b0, nb);
        }
      }
      // a post that failed leaves a peer with an operation nobody will match: it would sit in its group_end until the
      // watchdog -- abort right away (ncclCommAbort releases the blocked ranks), then close this rank's group
      if (!e.empty()) abort_once();
      const std::string e1 = T.group_end(r);  // (always: the group must be closed on this thread)
      if (e.empty()) e = e1;
      if (e.empty()) e = T.sync(r);
      return e;
    }, &hung);
  };
  note("scatter");
  err = grouped("1: scatter (rank 0 sends shard g to rank g)", true);
  if (!err.empty()) {
    abort_once();  // a failed group: the communicators are not reusable
    return PoolOutcome::transport_abandoned;
  }
  note("score");
  err = run_on_all(threads, "2: scoring", 0.0, nullptr, [&](int r) -> std::string {  // (no watchdog: scoring time is the caller's batch)
    int b0, nb;
    pool_shard(B, G, r, b0, nb);
    return nb > 0 ? W.score(r, b0, nb) : std::string();
  }, &hung);
  if (!err.empty()) return PoolOutcome::failed_pool_usable;  // (nothing is pending in the transport)
  note("gather");
  err = grouped("3: gather (rank g sends its scores to rank 0)", false);
  if (!err.empty()) {
    abort_once();
    return PoolOutcome::transport_abandoned;
  }
  return PoolOutcome::ok;
}

}  // namespace mig
