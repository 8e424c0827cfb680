#pragma once
#include <mutex>
namespace boost {
typedef std::mutex mutex;
typedef std::recursive_mutex recursive_mutex;
template <class M> using lock_guard = std::lock_guard<M>;
template <class M> using unique_lock = std::unique_lock<M>;
}
