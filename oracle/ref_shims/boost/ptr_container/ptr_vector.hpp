// Stand-in for boost::ptr_vector (oracle/_ref only): an owning vector of heap objects whose iterators and
// operator[] yield references, sorted through the pointees' operator<.
#pragma once
#include <algorithm>
#include <boost/serialization/access.hpp>
#include <cstddef>
#include <memory>
#include <vector>
namespace boost {
template <class T> class ptr_vector {
  std::vector<T *> v;
  template <class P, class R> struct iter {
    P p;
    typedef std::random_access_iterator_tag iterator_category;
    typedef T value_type;
    typedef std::ptrdiff_t difference_type;
    typedef R *pointer;
    typedef R &reference;
    R &operator*() const { return **p; }
    R *operator->() const { return *p; }
    R &operator[](difference_type n) const { return **(p + n); }
    iter &operator++() { ++p; return *this; }
    iter operator++(int) { iter t = *this; ++p; return t; }
    iter &operator--() { --p; return *this; }
    iter operator--(int) { iter t = *this; --p; return t; }
    iter &operator+=(difference_type n) { p += n; return *this; }
    iter &operator-=(difference_type n) { p -= n; return *this; }
    iter operator+(difference_type n) const { return iter{p + n}; }
    iter operator-(difference_type n) const { return iter{p - n}; }
    difference_type operator-(const iter &o) const { return p - o.p; }
    bool operator==(const iter &o) const { return p == o.p; }
    bool operator!=(const iter &o) const { return p != o.p; }
    bool operator<(const iter &o) const { return p < o.p; }
    P base() const { return p; }
  };
 public:
  typedef iter<typename std::vector<T *>::iterator, T> iterator;
  typedef iter<typename std::vector<T *>::const_iterator, const T> const_iterator;
  typedef T value_type;
  typedef std::size_t size_type;
  ptr_vector() {}
  ptr_vector(const ptr_vector &o) { for (T *p : o.v) v.push_back(new T(*p)); }
  ptr_vector &operator=(const ptr_vector &o) {
    if (this != &o) { clear(); for (T *p : o.v) v.push_back(new T(*p)); }
    return *this;
  }
  ~ptr_vector() { clear(); }
  void clear() { for (T *p : v) delete p; v.clear(); }
  void push_back(T *p) { v.push_back(p); }
  void pop_back() { delete v.back(); v.pop_back(); }
  std::size_t size() const { return v.size(); }
  bool empty() const { return v.empty(); }
  void reserve(std::size_t n) { v.reserve(n); }
  void resize(std::size_t n) { while (v.size() > n) pop_back(); while (v.size() < n) v.push_back(new T()); }
  T &operator[](std::size_t i) { return *v[i]; }
  const T &operator[](std::size_t i) const { return *v[i]; }
  T &front() { return *v.front(); }
  const T &front() const { return *v.front(); }
  T &back() { return *v.back(); }
  const T &back() const { return *v.back(); }
  iterator begin() { return iterator{v.begin()}; }
  iterator end() { return iterator{v.end()}; }
  const_iterator begin() const { return const_iterator{v.begin()}; }
  const_iterator end() const { return const_iterator{v.end()}; }
  iterator erase(iterator it) { delete *it.base(); return iterator{v.erase(it.base())}; }
  iterator erase(iterator a, iterator b) { for (auto p = a.base(); p != b.base(); ++p) delete *p; return iterator{v.erase(a.base(), b.base())}; }
  void sort() { std::sort(v.begin(), v.end(), [](const T *a, const T *b) { return *a < *b; }); }
  template <class C> void sort(C c) { std::sort(v.begin(), v.end(), [&](const T *a, const T *b) { return c(*a, *b); }); }
  template <class Archive> void serialize(Archive &, const unsigned) {}
};
}  // namespace boost
