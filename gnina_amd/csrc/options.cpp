// options.cpp -- see options.h
#include "options.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

namespace mig {

namespace {
const char *const kNames[OPT_COUNT] = {
#define X(n) #n,
    MI_OPTION_LIST(X)
#undef X
};
std::atomic<const char *> g_val[OPT_COUNT];
std::once_flag g_once;
void parse_env() {
  std::call_once(g_once, [] {
    for (int i = 0; i < OPT_COUNT; i++) {
      const char *v = getenv(kNames[i]);
      g_val[i].store(v ? strdup(v) : nullptr, std::memory_order_release);
    }
  });
}
}  // namespace

const char *option(OptionId id) {
  parse_env();
  return g_val[id].load(std::memory_order_acquire);
}

int set_option(const char *name, const char *value) {
  if (!name) return -1;
  parse_env();
  for (int i = 0; i < OPT_COUNT; i++)
    if (!strcmp(name, kNames[i])) {
      g_val[i].store(value ? strdup(value) : nullptr, std::memory_order_release);
      return 0;
    }
  return -1;
}

const char *options_summary() {
  static thread_local std::string out;
  out.clear();
  for (int i = 0; i < OPT_COUNT; i++)
    if (const char *v = option((OptionId)i)) {
      if (!out.empty()) out += ' ';
      out += kNames[i];
      out += '=';
      out += v;
    }
  return out.c_str();
}

}  // namespace mig
