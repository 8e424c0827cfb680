// Stand-in for boost::hash / hash_combine (oracle/_ref only): hash values only steer unordered containers.
#pragma once
#include <cstddef>
#include <functional>
namespace boost {
template <class T> struct hash;
template <class T> inline void hash_combine(std::size_t &seed, const T &v);
namespace hash_detail {
template <class T> auto call(const T &v, int) -> decltype(hash_value(v)) { return hash_value(v); }
template <class T> std::size_t call(const T &v, long) { return std::hash<T>()(v); }
}
template <class T> struct hash {
  std::size_t operator()(const T &v) const { return hash_detail::call(v, 0); }
};
template <class T> inline void hash_combine(std::size_t &seed, const T &v) {
  seed ^= boost::hash<T>()(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
}
}
