// Stand-in for boost::math::quaternion (oracle/_ref only): gnina keeps its own `qt` and uses the Boost type only as
// a value carrier (quaternion.h:47-57); the Hamilton product is written in Boost's operand order.
#pragma once
#include <cmath>
namespace boost { namespace math {
template <class T> class quaternion {
  T a, b, c, d;
 public:
  quaternion(T a_ = T(), T b_ = T(), T c_ = T(), T d_ = T()) : a(a_), b(b_), c(c_), d(d_) {}
  T R_component_1() const { return a; }
  T R_component_2() const { return b; }
  T R_component_3() const { return c; }
  T R_component_4() const { return d; }
  T real() const { return a; }
  quaternion &operator*=(const quaternion &r) {
    T at = +a * r.a - b * r.b - c * r.c - d * r.d;
    T bt = +a * r.b + b * r.a + c * r.d - d * r.c;
    T ct = +a * r.c - b * r.d + c * r.a + d * r.b;
    T dt = +a * r.d + b * r.c - c * r.b + d * r.a;
    a = at, b = bt, c = ct, d = dt;
    return *this;
  }
  quaternion &operator*=(T s) { a *= s, b *= s, c *= s, d *= s; return *this; }
  quaternion &operator/=(T s) { a /= s, b /= s, c /= s, d /= s; return *this; }
  friend quaternion operator*(quaternion l, const quaternion &r) { return l *= r; }
  friend bool operator==(const quaternion &l, const quaternion &r) { return l.a == r.a && l.b == r.b && l.c == r.c && l.d == r.d; }
};
template <class T> T norm(const quaternion<T> &q) {
  return q.R_component_1() * q.R_component_1() + q.R_component_2() * q.R_component_2() +
         q.R_component_3() * q.R_component_3() + q.R_component_4() * q.R_component_4();
}
template <class T> T abs(const quaternion<T> &q) { return std::sqrt(norm(q)); }
template <class T> quaternion<T> conj(const quaternion<T> &q) {
  return quaternion<T>(q.R_component_1(), -q.R_component_2(), -q.R_component_3(), -q.R_component_4());
}
}}
