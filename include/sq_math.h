/* sq_math.h — deterministic fp64 elementary functions shared by host and gfx950 device code.
 *
 * Why this exists: the hot path's fp64 quantities (auxiliary log-probabilities, range-factorization
 * bins, VBEM exp(digamma) weights) feed integer decisions (bin ids, convergence iteration count).
 * libm (glibc) and the ROCm device library differ by an ulp in exp/log, which would flip those
 * decisions between the CPU checker and the GPU path.  Every function here is a fixed sequence of
 * IEEE-754 binary64 add/mul/fma/div operations, so a host build (-ffp-contract=off) and a device
 * build (-ffp-contract=off) produce bit-identical results.
 *
 * Reference arithmetic these stand in for: std::log / std::exp in
 * include/salmon/internal/util/SalmonMath.hpp:40-68 and boost::math::digamma in
 * src/inference/CollapsedEMOptimizer.cpp:119,127,256,269 (reference file:line).
 * Accuracy (tests/test_math.py): exp, log <= 2 ulp vs libm; digamma rel. err < 5e-15 vs scipy.
 */
#ifndef SQ_MATH_H
#define SQ_MATH_H

#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define SQ_HD __host__ __device__ inline
#else
#define SQ_HD static inline
#endif

/* salmon's log-space constants (SalmonMath.hpp:40-46). LOG_0 is +inf by design. */
#define SQ_LOG_0 ((double)HUGE_VAL)
#define SQ_LOG_1 (0.0)
#define SQ_LOG_EPSILON (-24.006680182952184) /* log(0.375e-10), fixed literal so host == device */

SQ_HD double sq_bits2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
SQ_HD uint64_t sq_d2bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

SQ_HD double sq_fma(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fma(a, b, c);
#else
  return fma(a, b, c);
#endif
}

/* exp(x): k = rint(x/ln2); r = x - k ln2 (two-part); Taylor to degree 13 by Horner+fma; scale 2^k. */
SQ_HD double sq_exp(double x) {
  if (x != x) return x;
  if (x > 709.782712893384) return (double)HUGE_VAL;
  if (x < -745.1332191019412) return 0.0;
  const double INV_LN2 = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  double kd = x * INV_LN2;
  /* round to nearest integer, ties away irrelevant (any consistent choice is fine) */
  kd = (kd >= 0.0) ? (double)(int64_t)(kd + 0.5) : (double)(int64_t)(kd - 0.5);
  double r = sq_fma(-kd, LN2_HI, x);
  r = sq_fma(-kd, LN2_LO, r);
  double p = 1.0 / 6227020800.0;            /* 1/13! */
  p = sq_fma(p, r, 1.0 / 479001600.0);      /* 1/12! */
  p = sq_fma(p, r, 1.0 / 39916800.0);
  p = sq_fma(p, r, 1.0 / 3628800.0);
  p = sq_fma(p, r, 1.0 / 362880.0);
  p = sq_fma(p, r, 1.0 / 40320.0);
  p = sq_fma(p, r, 1.0 / 5040.0);
  p = sq_fma(p, r, 1.0 / 720.0);
  p = sq_fma(p, r, 1.0 / 120.0);
  p = sq_fma(p, r, 1.0 / 24.0);
  p = sq_fma(p, r, 1.0 / 6.0);
  p = sq_fma(p, r, 0.5);
  p = sq_fma(p, r, 1.0);
  p = sq_fma(p, r, 1.0);
  int64_t k = (int64_t)kd;
  /* scale by 2^k in two steps to stay exact through the subnormal range */
  int64_t k1 = k / 2, k2 = k - k1;
  double s1 = sq_bits2d((uint64_t)(k1 + 1023) << 52);
  double s2 = sq_bits2d((uint64_t)(k2 + 1023) << 52);
  return (p * s1) * s2;
}

/* log(x) for x > 0 (x <= 0 is handled by callers, mirroring salmon::math::log). */
SQ_HD double sq_log(double x) {
  if (x != x) return x;
  if (x < 0.0) return sq_bits2d(0x7ff8000000000000ULL);
  if (x == 0.0) return -(double)HUGE_VAL;
  if (x == (double)HUGE_VAL) return x;
  uint64_t u = sq_d2bits(x);
  int64_t e = 0;
  if ((u >> 52) == 0) { /* subnormal: normalise */
    x = x * 18014398509481984.0; /* 2^54 */
    u = sq_d2bits(x);
    e = -54;
  }
  e += (int64_t)((u >> 52) & 0x7ff) - 1023;
  u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
  double m = sq_bits2d(u); /* [1,2) */
  if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  /* 2*atanh(s) = 2s (1 + z/3 + z^2/5 + ... ) ; |s| <= 0.1716 -> 12 terms */
  double p = 1.0 / 25.0;
  p = sq_fma(p, z, 1.0 / 23.0);
  p = sq_fma(p, z, 1.0 / 21.0);
  p = sq_fma(p, z, 1.0 / 19.0);
  p = sq_fma(p, z, 1.0 / 17.0);
  p = sq_fma(p, z, 1.0 / 15.0);
  p = sq_fma(p, z, 1.0 / 13.0);
  p = sq_fma(p, z, 1.0 / 11.0);
  p = sq_fma(p, z, 1.0 / 9.0);
  p = sq_fma(p, z, 1.0 / 7.0);
  p = sq_fma(p, z, 1.0 / 5.0);
  p = sq_fma(p, z, 1.0 / 3.0);
  /* log(m) = 2s + 2s*z*p */
  double t = (2.0 * s) * (z * p);
  double lm = sq_fma(2.0, s, t);
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  double ed = (double)e;
  return sq_fma(ed, LN2_HI, sq_fma(ed, LN2_LO, lm));
}

/* digamma(x), x > 0: shift up with psi(x) = psi(x+1) - 1/x until x >= 10, then asymptotic series. */
SQ_HD double sq_digamma(double x) {
  double acc = 0.0;
  while (x < 10.0) { acc -= 1.0 / x; x += 1.0; }
  double inv = 1.0 / x;
  double inv2 = inv * inv;
  /* sum_{n} B_{2n} / (2n x^{2n}) : 1/12, -1/120, 1/252, -1/240, 1/132, -691/32760, 1/12 */
  double p = 1.0 / 12.0;
  p = sq_fma(p, inv2, -691.0 / 32760.0);
  p = sq_fma(p, inv2, 1.0 / 132.0);
  p = sq_fma(p, inv2, -1.0 / 240.0);
  p = sq_fma(p, inv2, 1.0 / 252.0);
  p = sq_fma(p, inv2, -1.0 / 120.0);
  p = sq_fma(p, inv2, 1.0 / 12.0);
  double r = sq_log(x) - 0.5 * inv - inv2 * p;
  return r + acc;
}

/* salmon::math::logAdd (SalmonMath.hpp:55-67): LOG_0 (= +inf) is the additive identity. */
SQ_HD double sq_log_add(double x, double y) {
  if (fabs(x) == SQ_LOG_0) return y;
  if (fabs(y) == SQ_LOG_0) return x;
  if (y > x) { double t = x; x = y; y = t; }
  return x + sq_log(1.0 + sq_exp(y - x));
}

/* fixed-point accumulation helpers: order-independent sums on the GPU (integer atomics) that the
 * CPU checker reproduces exactly.  Used for eq-class weight sums (36 fractional bits, values in
 * [0,1], up to 2^28 fragments per class) and online transcript-mass increments (40 bits). */
#define SQ_WFRAC_BITS 36
#define SQ_MFRAC_BITS 40
SQ_HD uint64_t sq_to_fixed(double v, int bits) {
  double s = v * (double)(1ULL << bits);
  return (uint64_t)(s + 0.5);
}
SQ_HD double sq_from_fixed(uint64_t q, int bits) {
  return (double)q / (double)(1ULL << bits);
}

/* 64-bit mixing (murmur3 fmix64, bijective) and a label hash built on it. */
SQ_HD uint64_t sq_mix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return k;
}

#endif /* SQ_MATH_H */
