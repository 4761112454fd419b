// host/sha2.h — SHA-256 / SHA-512 (FIPS 180-4), streaming, for the index's sequence / name digests.
//
// The reference index carries the digests of the reference sequences and names (pufferfish's info.json keys SeqHash, NameHash, SeqHash512,
// NameHash512, DecoySeqHash, DecoyNameHash), `salmon quant` copies them into aux_info/meta_info.json (include/salmon/internal/index/
// SalmonIndex.hpp:94-98,138-147; src/output/GZipWriter.cpp:573-578) and downstream tools (tximeta) identify the transcriptome by them.
// Written from the standard: initial values = fractional parts of the square roots of the first 8 primes, round constants = fractional
// parts of the cube roots of the first 64 / 80 primes (generated below, not tabulated), big-endian message schedule.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <cmath>

namespace sqsha {

inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

// first `bits` fractional bits of the `root`-th root (2 = square, 3 = cube) of prime p, by integer bisection on p << (root * bits)
// (exact: no floating point).  x = floor(p^(1/root) * 2^bits); the constant is x mod 2^bits.
inline uint64_t frac_root(uint32_t p, int root, int bits) {
  // find the largest x with x^root <= p * 2^(root*bits); x < 2^(bits + 4) since p < 2^(4 * root) for p <= 409 and root >= 2
  typedef unsigned __int128 u128;
  auto le = [&](u128 x) -> bool {   // x^root <= p << (root * bits), evaluated without overflow in 256-bit pieces for root 3 at 64 bits
    if (root == 2) {   // x < 2^68, x^2 < 2^136: split x = hi * 2^34 + lo
      // compare x*x with p << (2*bits) using 128-bit halves
      const u128 lo = x & (((u128)1 << 34) - 1), hi = x >> 34;
      // x^2 = hi^2 << 68 + 2 hi lo << 34 + lo^2 ; accumulate into (H, L) with L holding the low 68 bits
      u128 L = lo * lo, H = hi * hi;
      u128 mid = 2 * hi * lo;                         // < 2^70
      L += (mid & (((u128)1 << 34) - 1)) << 34; H += mid >> 34;
      H += L >> 68; L &= (((u128)1 << 68) - 1);
      // target = p << (2*bits): in the same split
      const int sh = 2 * bits; u128 TL, TH;
      if (sh >= 68) { TL = 0; TH = (u128)p << (sh - 68); } else { TL = ((u128)p << sh) & (((u128)1 << 68) - 1); TH = ((u128)p << sh) >> 68; }
      return H < TH || (H == TH && L <= TL);
    }
    // root 3: x < 2^(bits + 3) <= 2^67; use long multiplication in base 2^32 limbs
    uint32_t a[3] = {(uint32_t)x, (uint32_t)(x >> 32), (uint32_t)(x >> 64)};
    uint64_t sq[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) { uint64_t c = 0; for (int j = 0; j < 3; ++j) { const uint64_t t = (uint64_t)a[i] * a[j] + sq[i + j] + c; sq[i + j] = (uint32_t)t; c = t >> 32; } sq[i + 3] += c; }
    uint64_t cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; ++i) { uint64_t c = 0; for (int j = 0; j < 3; ++j) { const uint64_t t = sq[i] * a[j] + cu[i + j] + c; cu[i + j] = (uint32_t)t; c = t >> 32; } cu[i + 3] += c; }
    for (int i = 0; i < 8; ++i) { cu[i + 1] += cu[i] >> 32; cu[i] &= 0xFFFFFFFFu; }
    uint64_t tg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; const int sh = 3 * bits;
    { const int limb = sh / 32, off = sh % 32; const uint64_t v = (uint64_t)p << off; tg[limb] = (uint32_t)v; tg[limb + 1] = v >> 32; }
    for (int i = 8; i >= 0; --i) { if (cu[i] != tg[i]) return cu[i] < tg[i]; }
    return true;
  };
  u128 lo = 0, hi = (u128)1 << (bits + 5);
  while (hi - lo > 1) { const u128 mid = lo + (hi - lo) / 2; if (le(mid)) lo = mid; else hi = mid; }
  return bits == 64 ? (uint64_t)lo : (uint64_t)(lo & (((u128)1 << bits) - 1));
}

struct Consts {
  uint32_t k256[64], h256[8]; uint64_t k512[80], h512[8];
  Consts() {
    int n = 0;
    for (uint32_t p = 2; n < 80; ++p) {
      bool prime = true; for (uint32_t d = 2; d * d <= p; ++d) if (p % d == 0) { prime = false; break; }
      if (!prime) continue;
      if (n < 64) k256[n] = (uint32_t)frac_root(p, 3, 32);
      k512[n] = frac_root(p, 3, 64);
      if (n < 8) { h256[n] = (uint32_t)frac_root(p, 2, 32); h512[n] = frac_root(p, 2, 64); }
      ++n;
    }
  }
};
inline const Consts& consts() { static const Consts c; return c; }

struct Sha256 {
  uint32_t h[8]; uint8_t buf[64]; uint64_t len = 0; size_t fill = 0;
  Sha256() { memcpy(h, consts().h256, sizeof(h)); }
  void block(const uint8_t* p) {
    const uint32_t* K = consts().k256; uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      const uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      const uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
      const uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data; len += n;
    if (fill) { const size_t t = n < 64 - fill ? n : 64 - fill; memcpy(buf + fill, p, t); fill += t; p += t; n -= t; if (fill == 64) { block(buf); fill = 0; } }
    for (; n >= 64; p += 64, n -= 64) block(p);
    if (n) { memcpy(buf, p, n); fill = n; }
  }
  std::string hex() {   // finishes the digest
    const uint64_t bits = len * 8; uint8_t pad[72] = {0x80}; const size_t pn = (fill < 56 ? 56 : 120) - fill;
    update(pad, pn); uint8_t lb[8]; for (int i = 0; i < 8; ++i) lb[i] = (uint8_t)(bits >> (56 - 8 * i)); update(lb, 8);
    static const char* hx = "0123456789abcdef"; std::string s;
    for (int i = 0; i < 8; ++i) for (int j = 28; j >= 0; j -= 4) s.push_back(hx[(h[i] >> j) & 15]);
    return s;
  }
};

struct Sha512 {
  uint64_t h[8]; uint8_t buf[128]; uint64_t len = 0; size_t fill = 0;
  Sha512() { memcpy(h, consts().h512, sizeof(h)); }
  void block(const uint8_t* p) {
    const uint64_t* K = consts().k512; uint64_t w[80];
    for (int i = 0; i < 16; ++i) { uint64_t v = 0; for (int j = 0; j < 8; ++j) v = (v << 8) | p[8 * i + j]; w[i] = v; }
    for (int i = 16; i < 80; ++i) {
      const uint64_t s0 = rotr64(w[i - 15], 1) ^ rotr64(w[i - 15], 8) ^ (w[i - 15] >> 7), s1 = rotr64(w[i - 2], 19) ^ rotr64(w[i - 2], 61) ^ (w[i - 2] >> 6);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 80; ++i) {
      const uint64_t S1 = rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
      const uint64_t S0 = rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data; len += n;
    if (fill) { const size_t t = n < 128 - fill ? n : 128 - fill; memcpy(buf + fill, p, t); fill += t; p += t; n -= t; if (fill == 128) { block(buf); fill = 0; } }
    for (; n >= 128; p += 128, n -= 128) block(p);
    if (n) { memcpy(buf, p, n); fill = n; }
  }
  std::string hex() {
    const uint64_t bits = len * 8; uint8_t pad[136] = {0x80}; const size_t pn = (fill < 112 ? 112 : 240) - fill;
    update(pad, pn); uint8_t lb[16] = {0}; for (int i = 0; i < 8; ++i) lb[8 + i] = (uint8_t)(bits >> (56 - 8 * i)); update(lb, 16);
    static const char* hx = "0123456789abcdef"; std::string s;
    for (int i = 0; i < 8; ++i) for (int j = 60; j >= 0; j -= 4) s.push_back(hx[(h[i] >> j) & 15]);
    return s;
  }
};

}  // namespace sqsha
