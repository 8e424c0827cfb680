// Inert archives (oracle/_ref only): cache::save / load and the model (de)serialisers compile but do nothing.
#pragma once
#include <boost/serialization/access.hpp>
#include <iosfwd>
namespace boost { namespace archive {
struct inert_archive {
  template <class S> explicit inert_archive(S &, unsigned = 0) {}
  template <class T> inert_archive &operator&(T &) { return *this; }
  template <class T> inert_archive &operator<<(const T &) { return *this; }
  template <class T> inert_archive &operator>>(T &) { return *this; }
  library_version_type get_library_version() const { return library_version_type(0); }
};
typedef inert_archive binary_iarchive, binary_oarchive, text_iarchive, text_oarchive;
}}
