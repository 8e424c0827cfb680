#pragma once
#include <boost/serialization/access.hpp>
namespace boost { namespace serialization { namespace detail {
template <class Archive, class T> struct stack_construct {
  T t;
  stack_construct(Archive &, unsigned) : t() {}
  T &reference() { return t; }
};
}}}
