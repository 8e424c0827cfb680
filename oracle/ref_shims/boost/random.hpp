// Stand-in for Boost.Random (oracle/_ref only).  mt19937 is the standard generator (std::mt19937 is the same
// algorithm and seeding).  uniform_real / uniform_int restate Boost's published algorithms (>= 1.47:
// generate_uniform_real: numerator / divisor * (max - min) + min with divisor = 2^32, redrawn if the result
// reaches max; generate_uniform_int: equal buckets with rejection).  normal_distribution is Box-Muller (Boost
// < 1.56; later versions use a ziggurat): gnina pins no Boost version, so the stream of normals is "unpinned".
#pragma once
#include <cmath>
#include <cstdint>
#include <random>
namespace boost {
typedef std::mt19937 mt19937;
namespace random { typedef std::mt19937 mt19937; }
template <class T = double> class uniform_real {
  T lo, hi;
 public:
  typedef T result_type;
  typedef T input_type;
  explicit uniform_real(T a = T(0), T b = T(1)) : lo(a), hi(b) {}
  template <class E> T operator()(E &eng) const {
    for (;;) {
      T numerator = static_cast<T>(eng() - (E::min)());
      T divisor = static_cast<T>((E::max)() - (E::min)()) + 1;
      T result = numerator / divisor * (hi - lo) + lo;
      if (result < hi) return result;
    }
  }
};
template <class I = int> class uniform_int {
  I lo, hi;
 public:
  typedef I result_type;
  typedef I input_type;
  explicit uniform_int(I a = 0, I b = 9) : lo(a), hi(b) {}
  template <class E> I operator()(E &eng) const {
    typedef std::uint32_t U;
    const U range = U(hi) - U(lo), brange = U((E::max)() - (E::min)());
    if (range == 0) return lo;
    if (range == brange) return I(U(eng() - (E::min)()) + U(lo));
    U bucket = brange / (range + 1);
    if (brange % (range + 1) == range) ++bucket;
    for (;;) {
      U r = U(eng() - (E::min)()) / bucket;
      if (r <= range) return I(r + U(lo));
    }
  }
};
template <class T = double> class normal_distribution {
  T mean_, sigma_;
  mutable T r1, r2, cached_rho;
  mutable bool valid;
 public:
  typedef T result_type;
  typedef T input_type;
  explicit normal_distribution(T m = T(0), T s = T(1)) : mean_(m), sigma_(s), r1(0), r2(0), cached_rho(0), valid(false) {}
  template <class E> T operator()(E &eng) const {
    uniform_real<T> u01(T(0), T(1));
    if (!valid) {
      r1 = u01(eng);
      r2 = u01(eng);
      cached_rho = std::sqrt(-T(2) * std::log(T(1) - r2));
      valid = true;
    } else {
      valid = false;
    }
    const T pi = T(3.14159265358979323846);
    return cached_rho * (valid ? std::cos(T(2) * pi * r1) : std::sin(T(2) * pi * r1)) * sigma_ + mean_;
  }
};
template <class Engine, class Dist> class variate_generator {
  Engine eng;  // Engine is a reference type in gnina (rng&)
  Dist dist;
 public:
  typedef typename Dist::result_type result_type;
  variate_generator(Engine e, Dist d) : eng(e), dist(d) {}
  result_type operator()() { return dist(eng); }
};
}  // namespace boost
