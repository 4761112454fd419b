// oracle/_stub/boost/math/distributions/normal.hpp — TEST INFRASTRUCTURE.  Boost is absent; the fragment-length prior of the reference is
// cdf(normal(mu, sd), i + .5) - cdf(., i - .5).  This stand-in computes the same function with libm's erfc, so the pin built on it
// (tests/test_fld_pin.py) holds the STRUCTURE of FragmentLengthDistribution.cpp (kernel placement, bins, pmf / cmf / cacheCMF, min) to the reference's
// own code — not Boost's last bits of the prior table, which stay unpinned.
#pragma once
#include <cmath>
namespace boost { namespace math {
class normal { double mu_, sd_; public: normal(double mu = 0.0, double sd = 1.0) : mu_(mu), sd_(sd) {} double mean() const { return mu_; } double standard_deviation() const { return sd_; } };
inline double cdf(const normal& n, double x) { return 0.5 * std::erfc(-((x - n.mean()) / n.standard_deviation()) * 0.70710678118654752440); }
} }
