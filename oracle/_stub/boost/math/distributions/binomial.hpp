// oracle/_stub/boost/math/distributions/binomial.hpp — TEST INFRASTRUCTURE (Boost is absent).  pdf(binomial(n, p), k) = C(n, k) p^k (1 - p)^(n - k), by the
// product formula: exact for the kernel the reference uses (n = 4, p = 0.5: 1/16, 4/16, 6/16, 4/16, 1/16).
#pragma once
#include <cmath>
namespace boost { namespace math {
template <class T = double> class binomial_distribution { T n_, p_; public: binomial_distribution(T n, T p) : n_(n), p_(p) {} T trials() const { return n_; } T success_fraction() const { return p_; } };
template <class T> inline T pdf(const binomial_distribution<T>& d, T k) {
  T c = 1; const int n = (int)d.trials(), kk = (int)k; for (int i = 1; i <= kk; ++i) c = c * (T)(n - kk + i) / (T)i;
  return c * std::pow(d.success_fraction(), (T)kk) * std::pow((T)1 - d.success_fraction(), (T)(n - kk));
}
template <class T> inline T pdf(const binomial_distribution<T>& d, unsigned long k) { return pdf(d, (T)k); }
} }
