// host/crc32_fast.h — CRC-32 (the gzip / zlib polynomial, reflected 0xEDB88320) by carry-less multiplication: 64 bytes per trip folded into four
// 128-bit accumulators, then reduced (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009; the
// constants are x^n mod P for the fold distances, bit-reflected).  zlib 1.2.11's table-driven crc32 does ~1 GB/s per thread here; the parallel gzip
// reader (pgzip.cpp) checks every byte it inflates, so the checksum was a third of the inflating itself.  Falls back to zlib's where the CPU has no
// PCLMULQDQ, and for short buffers.  Same value as zlib's crc32(crc, buf, len) for every input (tests/test_pgzip.py).
#pragma once
#include <cstddef>
#include <cstdint>
#include <immintrin.h>
#include <zlib.h>

namespace sqcrc {
__attribute__((target("pclmul,sse4.1")))
static inline uint32_t crc32_clmul(uint32_t crc, const uint8_t* p, size_t n) {   // n >= 64, multiple of 16 handled by the caller's tail
  // fold constants for the reflected polynomial: k1 = x^(4*128+32) mod P, k2 = x^(4*128-32) mod P (fold by 64 bytes); k3 = x^(128+32), k4 = x^(128-32) (fold by 16)
  const __m128i k1k2 = _mm_set_epi64x(0x00000001c6e41596LL, 0x0000000154442bd4LL);
  const __m128i k3k4 = _mm_set_epi64x(0x00000000ccaa009eLL, 0x00000001751997d0LL);
  const __m128i k5k0 = _mm_set_epi64x(0x0000000000000000LL, 0x0000000163cd6124LL);
  const __m128i poly = _mm_set_epi64x(0x00000001f7011641LL, 0x00000001db710641LL);
  __m128i x0 = _mm_loadu_si128((const __m128i*)(p + 0)), x1 = _mm_loadu_si128((const __m128i*)(p + 16)), x2 = _mm_loadu_si128((const __m128i*)(p + 32)), x3 = _mm_loadu_si128((const __m128i*)(p + 48));
  x0 = _mm_xor_si128(x0, _mm_cvtsi32_si128((int)crc));
  p += 64; n -= 64;
  while (n >= 64) {
    __m128i t0 = _mm_clmulepi64_si128(x0, k1k2, 0x00), t1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), t2 = _mm_clmulepi64_si128(x2, k1k2, 0x00), t3 = _mm_clmulepi64_si128(x3, k1k2, 0x00);
    x0 = _mm_clmulepi64_si128(x0, k1k2, 0x11); x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11); x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
    x0 = _mm_xor_si128(_mm_xor_si128(x0, t0), _mm_loadu_si128((const __m128i*)(p + 0)));
    x1 = _mm_xor_si128(_mm_xor_si128(x1, t1), _mm_loadu_si128((const __m128i*)(p + 16)));
    x2 = _mm_xor_si128(_mm_xor_si128(x2, t2), _mm_loadu_si128((const __m128i*)(p + 32)));
    x3 = _mm_xor_si128(_mm_xor_si128(x3, t3), _mm_loadu_si128((const __m128i*)(p + 48)));
    p += 64; n -= 64;
  }
  // four accumulators -> one
  __m128i t;
  t = _mm_clmulepi64_si128(x0, k3k4, 0x00); x0 = _mm_clmulepi64_si128(x0, k3k4, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, t), x0);
  t = _mm_clmulepi64_si128(x1, k3k4, 0x00); x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11); x2 = _mm_xor_si128(_mm_xor_si128(x2, t), x1);
  t = _mm_clmulepi64_si128(x2, k3k4, 0x00); x2 = _mm_clmulepi64_si128(x2, k3k4, 0x11); x3 = _mm_xor_si128(_mm_xor_si128(x3, t), x2);
  while (n >= 16) {   // single 16-byte folds
    t = _mm_clmulepi64_si128(x3, k3k4, 0x00); x3 = _mm_clmulepi64_si128(x3, k3k4, 0x11);
    x3 = _mm_xor_si128(_mm_xor_si128(x3, t), _mm_loadu_si128((const __m128i*)p)); p += 16; n -= 16;
  }
  // 128 -> 64 bits
  const __m128i mask32 = _mm_set_epi32(0, 0, 0, -1);   // low 32 bits
  t = _mm_clmulepi64_si128(x3, k3k4, 0x10);            // low 64 bits of x3 times k4
  x3 = _mm_xor_si128(_mm_srli_si128(x3, 8), t);
  t = _mm_and_si128(x3, mask32);                       // 96 -> 64: low 32 bits times k5
  x3 = _mm_srli_si128(x3, 4);
  t = _mm_clmulepi64_si128(t, k5k0, 0x00);
  x3 = _mm_xor_si128(x3, t);
  // Barrett reduction 64 -> 32
  t = _mm_and_si128(x3, mask32); t = _mm_clmulepi64_si128(t, poly, 0x10);   // times mu
  t = _mm_and_si128(t, mask32);  t = _mm_clmulepi64_si128(t, poly, 0x00);   // times P
  x3 = _mm_xor_si128(x3, t);
  return (uint32_t)_mm_extract_epi32(x3, 1);
}
// same contract as zlib's crc32(): crc of the bytes so far in, crc including [p, p + n) out
static inline uint32_t crc32(uint32_t crc, const void* buf, size_t n) {
  static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
  const uint8_t* p = (const uint8_t*)buf;
  if (have && n >= 256) {
    const size_t body = n & ~(size_t)15;
    crc = ~crc32_clmul(~crc, p, body); p += body; n -= body;
  }
  while (n) { const size_t step = n < (1u << 30) ? n : (size_t)1 << 30; crc = (uint32_t)::crc32(crc, (const Bytef*)p, (uInt)step); p += step; n -= step; }
  return crc;
}
}  // namespace sqcrc
