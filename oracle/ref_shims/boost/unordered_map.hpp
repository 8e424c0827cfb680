#pragma once
#include <boost/functional/hash.hpp>
#include <unordered_map>
#include <unordered_set>
namespace boost {
template <class K, class V, class H = boost::hash<K>, class E = std::equal_to<K>> using unordered_map = std::unordered_map<K, V, H, E>;
template <class K, class H = boost::hash<K>, class E = std::equal_to<K>> using unordered_set = std::unordered_set<K, H, E>;
}
