/* sq_rng.h — counter-based random numbers for the sampling rows (bootstrap a16, Gibbs a17).
 * The reference seeds std::mt19937 / pcg32 from std::random_device (CollapsedEMOptimizer.cpp:427-436,
 * CollapsedGibbsSampler.cpp:109-119), so its replicates are only statistically defined.  Here every
 * draw is a pure function of (seed, stream, index): host and device produce the same replicates,
 * whatever the thread order, and integer atomics make the accumulated counts order-free. */
#ifndef SQ_RNG_H
#define SQ_RNG_H
#include "sq_math.h"

SQ_HD uint64_t sq_r64(uint64_t seed, uint64_t a, uint64_t b) {
  return sq_mix64(seed ^ sq_mix64(a * 0x9E3779B97F4A7C15ULL + 0x1234567ULL) ^ sq_mix64(b * 0xD1B54A32D192ED03ULL + 0x89ABCDEFULL));
}
SQ_HD double sq_u01(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }
SQ_HD uint64_t sq_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}
SQ_HD double sq_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_sqrt(x);
#else
  return sqrt(x);
#endif
}
/* Gamma(shape a, scale) — Marsaglia & Tsang (2000) with polar normals; stream (k1, k2) gives up to 4096 uniforms */
SQ_HD double sq_gamma_draw(double a, double scale, uint64_t seed, uint64_t k1, uint64_t k2) {
  uint64_t ctr = 0;
  double boost = 1.0;
  if (a < 1.0) {
    double u = sq_u01(sq_r64(seed, k1, k2 * 4096 + ctr++)); if (u <= 0.0) u = 1.0 / 9007199254740992.0;
    boost = sq_exp(sq_log(u) / a); a += 1.0;
  }
  const double d = a - 1.0 / 3.0, c = 1.0 / sq_sqrt(9.0 * d);
  for (int tries = 0; tries < 200 && ctr < 4000; ++tries) {
    double u1 = 2.0 * sq_u01(sq_r64(seed, k1, k2 * 4096 + ctr++)) - 1.0, u2 = 2.0 * sq_u01(sq_r64(seed, k1, k2 * 4096 + ctr++)) - 1.0;
    double s = u1 * u1 + u2 * u2;
    if (s >= 1.0 || s == 0.0) continue;
    double z = u1 * sq_sqrt(-2.0 * sq_log(s) / s);
    double v = 1.0 + c * z;
    if (v <= 0.0) continue;
    v = v * v * v;
    double u = sq_u01(sq_r64(seed, k1, k2 * 4096 + ctr++)); if (u <= 0.0) u = 1.0 / 9007199254740992.0;
    if (sq_log(u) < 0.5 * z * z + d - d * v + d * sq_log(v)) return d * v * boost * scale;
  }
  return d * boost * scale;
}
#endif
