// Stand-in for boost::optional (oracle/_ref only).
#pragma once
#include <optional>
namespace boost {
struct none_t {};
static const none_t none = {};
template <class T> class optional : public std::optional<T> {
 public:
  using std::optional<T>::optional;
  optional() {}
  optional(none_t) {}
  optional(const std::optional<T> &o) : std::optional<T>(o) {}
  optional &operator=(none_t) { this->reset(); return *this; }
  optional &operator=(const T &t) { std::optional<T>::operator=(t); return *this; }
  T &get() { return **this; }
  const T &get() const { return **this; }
  T get_value_or(const T &d) const { return this->has_value() ? **this : d; }
  bool operator!() const { return !this->has_value(); }
  bool is_initialized() const { return this->has_value(); }
};
}
