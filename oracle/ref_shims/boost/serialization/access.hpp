// Stand-in for Boost.Serialization (oracle/_ref only): serialize() member templates are declared by the reference
// but never instantiated here; archives are inert.
#pragma once
#include <cstddef>
#include <type_traits>
#define BOOST_DEDUCED_TYPENAME typename
#define BOOST_STATIC_CONSTANT(type, assignment) static const type assignment
namespace boost {
using std::is_array;
using std::is_class;
using std::is_enum;
using std::is_fundamental;
template <class B, class D> struct is_base_and_derived : std::is_base_of<B, D> {};
namespace mpl {
struct integral_c_tag {};
template <int N> struct int_ { enum { value = N }; typedef int_ type; };
template <class C, class A, class B> struct eval_if { typedef typename std::conditional<C::value, A, B>::type::type type; };
template <class A, class B> struct or_ { enum { value = A::value || B::value }; };
}
namespace serialization {
class access {};
template <class Base, class Derived> Base &base_object(Derived &d) { return static_cast<Base &>(d); }
template <class T> struct nvp_t { T &v; };
template <class T> T &make_nvp(const char *, T &t) { return t; }
template <class T> const T &make_nvp(const char *, const T &t) { return t; }
struct item_version_type { unsigned v; item_version_type(unsigned x = 0) : v(x) {} operator unsigned() const { return v; } };
struct collection_size_type { std::size_t v; collection_size_type(std::size_t x = 0) : v(x) {} operator std::size_t() const { return v; } };
template <class Archive, class T> void split_free(Archive &, T &, const unsigned) {}
template <class Archive, class T> void split_member(Archive &, T &, const unsigned) {}
struct basic_traits {};
enum level_type { not_serializable = 0, primitive_type = 1, object_serializable = 2, object_class_info = 3 };
template <class T> struct implementation_level_impl { enum { value = object_serializable }; };
template <class T> struct version { enum { value = 0 }; };
}  // namespace serialization
namespace archive {
struct library_version_type { unsigned v; library_version_type(unsigned x = 0) : v(x) {} operator unsigned() const { return v; } };
}
}  // namespace boost
#define BOOST_SERIALIZATION_SPLIT_MEMBER() \
  template <class Archive> void serialize(Archive &, const unsigned int) {}
#define BOOST_SERIALIZATION_SPLIT_FREE(T)
#define BOOST_CLASS_VERSION(T, N)
#define BOOST_SERIALIZATION_NVP(x) x
