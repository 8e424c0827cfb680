#pragma once
#include <array>
#include <boost/serialization/access.hpp>
namespace boost {
template <class T, std::size_t N> struct array : std::array<T, N> {
  void assign(const T &t) { this->fill(t); }
};
}
