// common.h -- error plumbing shared by the engine sources (no exceptions cross the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <stdexcept>
#include <string>

namespace mig {

void set_last_error(const std::string &msg);

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define MIG_HIP(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      throw ::mig::Error(3, std::string(#expr) + " failed: " + hipGetErrorString(e_) + " at " +     \
                                __FILE__ + ":" + std::to_string(__LINE__));                         \
  } while (0)

#define MIG_CHECK(cond, code, msg)                                    \
  do {                                                                \
    if (!(cond)) throw ::mig::Error((code), std::string(msg));        \
  } while (0)

template <typename T> struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  // grow-only allocation
  void ensure(size_t count) {
    if (count <= n) return;
    release();
    MIG_HIP(hipMalloc((void **)&p, count * sizeof(T)));
    n = count;
  }
  void upload(const T *src, size_t count, hipStream_t s) {
    ensure(count);
    if (count) MIG_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Kernels whose dynamic LDS request can exceed the 64 KB default need their limit raised -- once per DEVICE (the
// attribute lives in the device's copy of the code object; a pool drives several devices from one process).
void ensure_max_lds(const void *kernel, int bytes);

// setenv("GPU_MAX_HW_QUEUES") exactly once per process (engine.cpp); called by mi_gnina_init and, ahead of its worker
// threads, by the pools
void process_env_once();

}  // namespace mig
