// options.h -- the library's behaviour switches, parsed ONCE.
//
// Every MI_* switch below used to be read with getenv() where it was consulted -- some per conv layer per call, on pool
// worker threads, while tests and bench.py mutated os.environ: a setenv / getenv race and avoidable work on the B = 1
// latency path.  Now the environment is read once per process (the first option() call, normally inside
// mi_gnina_init), the values live in an atomic table, and a switch is changed at run time through
// mi_gnina_set_option(name, value) -- which tests/ and bench.py use instead of os.environ.  Load-time switches (read while
// a model is built: kernel selection, LDS budgets, fusion) affect models loaded AFTER the change; run-time switches
// (consulted per call) take effect on the next call.  All of them are experiment / A-B knobs: the defaults are the product.
#pragma once

namespace mig {

#define MI_OPTION_LIST(X)            \
  X(MI_GNINA_NO_H2)                  \
  X(MI_GNINA_SPARSE_LDS_KB)          \
  X(MI_GNINA_NO_RELU_SKIP)           \
  X(MI_GNINA_NO_H2_BWD)              \
  X(MI_GNINA_LDS_KB)                 \
  X(MI_VOX_LDS_PAD)                  \
  X(MI_POOL_ALLOW_DUPLICATE_DEVICES) \
  X(MI_GNINA_NO_MT_X)                \
  X(MI_GNINA_NO_H2_BWD_DENSE)        \
  X(MI_GNINA_H2_PF)                  \
  X(MI_GNINA_H2_OCC)                 \
  X(MI_GNINA_H2_NO_SPLIT_TENSORS)    \
  X(MI_GNINA_CONV_PATH)              \
  X(MI_VINA_MC_WAVES)                \
  X(MI_VINA_MC_PROFILE)              \
  X(MI_POOL_NO_RCCL)                 \
  X(MI_GNINA_NO_UNPOOL_FUSE)         \
  X(MI_GNINA_NO_SPARSE)              \
  X(MI_GNINA_NO_LIG_BWD)             \
  X(MI_GNINA_NO_LAT)                 \
  X(MI_GNINA_NO_H2_BWD_K1)           \
  X(MI_GNINA_NO_FUSE1X1)             \
  X(MI_GNINA_NO_DENSE_SPLIT)         \
  X(MI_GNINA_H2_WN1)                 \
  X(MI_GNINA_H2_WLDS)                \
  X(MI_GNINA_H2_NO_SKIP)             \
  X(MI_GNINA_H2_NO_FUSE1X1)          \
  X(MI_GNINA_H2_LDS_KB)              \
  X(MI_GNINA_H2_DBG)                 \
  X(MI_GNINA_H2_BWD_SKIP)            \
  X(MI_GNINA_H2_BWD_PREPASS)         \
  X(MI_GNINA_D16_NP)                 \
  X(MI_GNINA_BF16_LDS_KB)            \
  X(MI_GNINA_ACT_GB)                 \
  X(MI_POOL_WATCHDOG_S)              \
  X(MI_GNINA_K1_TILE)                \
  X(MI_GNINA_D16_DBG)                \
  X(MI_GNINA_D16_TILE)               \
  X(MI_GNINA_D16_GROUP_MAX)          \
  X(MI_GNINA_H16_WLDS)               \
  X(MI_GNINA_OUT_COPY)               \
  X(MI_GNINA_LIG_COPY)               \
  X(MI_GNINA_NO_GMAX_FUSE)           \
  X(MI_GNINA_NO_GRAD_LANES)          \
  X(MI_GNINA_K1S_DBG)                \
  X(MI_GNINA_NO_LANES)               \
  X(MI_GNINA_LANES)                  \
  X(MI_GNINA_LANES_MAX_B)            \
  X(MI_GNINA_LANE_SLICE)             \
  X(MI_GNINA_D16_PERSIST)            \
  X(MI_GNINA_K1S_PERSIST)            \
  X(MI_VOX_DBG)                      \
  X(MI_GNINA_NO_CALL_LOCK)           \
  X(MI_GNINA_CALL_LOCK)              \
  X(MI_GNINA_H2_WS)                  \
  X(MI_GNINA_VOX_SERIAL)             \
  X(MI_GNINA_LANES_OTHERS)           \
  X(MI_GNINA_LANE_OFFSET)

enum OptionId {
#define X(n) OPT_##n,
  MI_OPTION_LIST(X)
#undef X
      OPT_COUNT
};

// the switch's value as the environment (or mi_gnina_set_option) gave it; nullptr = not set.  The returned string is never
// freed (an override leaks the few bytes of the value it replaces, so that a reader on another thread never sees freed memory).
const char *option(OptionId id);
// 0 = ok, -1 = no such switch.  value nullptr = unset.
int set_option(const char *name, const char *value);
// "NAME=value NAME2=value2" of the switches that are set (diagnostics: bench.py records it)
const char *options_summary();

}  // namespace mig
