// host/pgzip.cpp — see pgzip.h.  An own inflate (RFC 1951) that writes 16-bit symbols: 0..255 = a byte, 0x8000 | k = byte k of the 32 KB
// of text that precede the piece (unknown while the piece is decoded).  zlib supplies crc32 / crc32_combine only.
#include "pgzip.h"
#include "crc32_fast.h"
#include <immintrin.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <sys/mman.h>
#include <vector>

namespace {
constexpr int FB = 11;                 // bits of the one-step decoding tables
constexpr uint32_t WIN = 32768;
constexpr uint16_t MARK = 0x8000;

struct Bits {   // LSB-first bit reader over the whole file; reading past the end yields zeros (tell() > 8 n says so)
  const uint8_t* base; const uint8_t* end; const uint8_t* p; uint64_t buf = 0; int cnt = 0;
  Bits(const uint8_t* b, size_t n) : base(b), end(b + n), p(b) {}
  inline void refill() {
    if (__builtin_expect(p + 8 <= end, 1)) { uint64_t v; memcpy(&v, p, 8); buf |= v << cnt; const int adv = (63 - cnt) >> 3; p += adv; cnt += adv * 8; }
    else while (cnt <= 56) { buf |= (uint64_t)(p < end ? *p : 0) << cnt; ++p; cnt += 8; }
  }
  void seek(uint64_t bit) { p = base + (bit >> 3); buf = 0; cnt = 0; refill(); const int sh = (int)(bit & 7); buf >>= sh; cnt -= sh; }
  inline uint64_t tell() const { return (uint64_t)(p - base) * 8 - (uint64_t)cnt; }
  inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
  inline void drop(int n) { buf >>= n; cnt -= n; }
  inline uint32_t get(int n) { if (cnt < n) refill(); const uint32_t v = peek(n); drop(n); return v; }
};

template <int TB> struct HuffT {
  uint16_t fast[1 << TB];      // (symbol << 4) | length for codes of at most TB bits, 0 = longer (or unused)
  uint16_t count[16], symbol[288];
  bool complete = false; int used = 0;
  // canonical code from the lengths; false if over-subscribed.  `complete` = every bit string decodes
  bool build(const uint8_t* len, int n) {
    memset(count, 0, sizeof(count)); used = 0;
    for (int i = 0; i < n; ++i) { count[len[i]]++; used += len[i] != 0; }
    count[0] = 0;
    int left = 1; for (int l = 1; l <= 15; ++l) { left <<= 1; left -= count[l]; if (left < 0) return false; }
    complete = left == 0;
    uint16_t offs[16]; offs[1] = 0; for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int i = 0; i < n; ++i) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    memset(fast, 0, sizeof(fast));
    uint32_t code = 0; int idx = 0;
    for (int l = 1; l <= TB; ++l) {
      for (int k = 0; k < count[l]; ++k, ++idx, ++code) {
        uint32_t rev = 0; for (int b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b);
        const uint16_t e = (uint16_t)((symbol[idx] << 4) | l);
        for (uint32_t x = rev; x < (1u << TB); x += 1u << l) fast[x] = e;
      }
      code <<= 1;
    }
    return true;
  }
  inline int decode(Bits& br) const {   // the caller keeps >= 15 bits in the buffer
    const uint16_t e = fast[br.buf & ((1u << TB) - 1)];
    if (e) { br.drop(e & 15); return e >> 4; }
    int code = 0, first = 0, index = 0; uint64_t b = br.buf;
    for (int l = 1; l <= 15; ++l) {
      code |= (int)(b & 1); b >>= 1;
      const int c = count[l];
      if (code - c < first) { br.drop(l); return symbol[index + (code - first)]; }
      index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
  }
};

typedef HuffT<FB> Huff;
typedef HuffT<7> HuffCL;       // the code-length code: at most 7 bits

const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Out {   // symbols of a piece, preceded by the 32 K window symbols
  uint16_t* b = nullptr; size_t n = 0, cap = 0;
  ~Out() { free(b); }
  bool reserve(size_t want) { if (want <= cap) return true; size_t nc = std::max(want, cap + cap / 2 + (1u << 20)); uint16_t* nb = (uint16_t*)realloc(b, nc * 2); if (!nb) return false; b = nb; cap = nc;
    if (nc * 2 >= (4u << 20)) { const uintptr_t a = ((uintptr_t)b + 4095) & ~(uintptr_t)4095; (void)madvise((void*)a, (nc * 2 - (a - (uintptr_t)b)) & ~(size_t)4095, MADV_HUGEPAGE); }   // fewer first-touch faults
    return true; }
  bool start(size_t expect = 1u << 22) { n = 0; if (!reserve(WIN + expect)) return false; for (uint32_t k = 0; k < WIN; ++k) b[k] = (uint16_t)(MARK | k); n = WIN; return true; }
};

inline bool texty(int c) { return (c >= 32 && c < 127) || c == '\n' || c == '\r' || c == '\t'; }

// the code tables of a dynamic block (the reader stands behind the 3 header bits).  strict: what a searching piece demands of a candidate
bool read_dynamic(Bits& br, Huff& lit, Huff& dist, bool strict) {
  br.refill();
  const int hlit = (int)br.get(5) + 257, hdist = (int)br.get(5) + 1, hclen = (int)br.get(4) + 4;
  if (hlit > 286 || hdist > 30) return false;
  uint8_t cl[19] = {0};
  for (int i = 0; i < hclen; ++i) cl[CLORDER[i]] = (uint8_t)br.get(3);
  HuffCL clh; if (!clh.build(cl, 19)) return false;
  if (strict && !clh.complete) return false;
  uint8_t len[320]; int i = 0;
  while (i < hlit + hdist) {
    br.refill();
    const int s = clh.decode(br); if (s < 0) return false;
    if (s < 16) len[i++] = (uint8_t)s;
    else {
      int rep, v = 0;
      if (s == 16) { if (i == 0) return false; v = len[i - 1]; rep = 3 + (int)br.get(2); }
      else if (s == 17) rep = 3 + (int)br.get(3);
      else rep = 11 + (int)br.get(7);
      if (i + rep > hlit + hdist) return false;
      while (rep--) len[i++] = (uint8_t)v;
    }
  }
  if (len[256] == 0) return false;                               // no end-of-block code
  if (!lit.build(len, hlit)) return false;
  if (!dist.build(len + hlit, hdist)) return false;
  if (!lit.complete && (strict || lit.used != 1)) return false;   // zlib: incomplete codes only with a single symbol
  if (!dist.complete && dist.used > 1) return false;
  return true;
}

struct Fixed { Huff lit, dist; Fixed() { uint8_t l[288]; for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)); lit.build(l, 288); uint8_t d[30]; memset(d, 5, 30); dist.build(d, 30); } };
const Fixed& fixed_tables() { static const Fixed f; return f; }

enum { B_MORE = 0, B_FINAL = 1, B_BAD = 2 };
// one block from the reader's position.  text_only: reject bytes that cannot be in a FASTA/FASTQ file (a candidate block start is being
// tried); max_out bounds a trial
int inflate_block(Bits& br, Out& o, bool text_only, size_t max_out) {
  br.refill();
  const uint32_t fin = br.get(1), type = br.get(2);
  if (type == 3) return B_BAD;
  if (type == 0) {
    br.drop(br.cnt & 7);   // to the byte boundary
    br.refill();
    const uint32_t ln = br.get(16); br.refill(); const uint32_t nl = br.get(16);
    if ((ln ^ nl) != 0xFFFFu) return B_BAD;
    if (!o.reserve(o.n + ln + 8)) return B_BAD;
    const uint64_t at = br.tell() >> 3;                           // a byte boundary: the stored bytes are copied straight from the file
    if (at + ln > (uint64_t)(br.end - br.base)) return B_BAD;
    const uint8_t* src = br.base + at; uint16_t* dst = o.b + o.n;
    for (uint32_t i = 0; i < ln; ++i) { if (text_only && !texty(src[i])) return B_BAD; dst[i] = src[i]; }
    o.n += ln; br.seek((at + ln) * 8);
    return fin ? B_FINAL : B_MORE;
  }
  Huff dl, dd; const Huff* lit; const Huff* dist;
  if (type == 1) { lit = &fixed_tables().lit; dist = &fixed_tables().dist; }
  else { if (!read_dynamic(br, dl, dd, text_only)) return B_BAD; lit = &dl; dist = &dd; }
  const uint64_t end_bits = (uint64_t)(br.end - br.base) * 8;
  // the reader and the output cursor live in locals for the length of the block: their home objects sit next to other threads' (one store
  // per symbol into a shared cache line made four threads slower than one)
  Bits b = br; uint16_t* ob = o.b; size_t on = o.n, ocap = o.cap; int rc = -1;
  for (;;) {
    // [r4] a truncated or damaged stream can decode literals for ever (past the end of the input the reader yields zeros, and the all-zero
    // code may be a literal): the end of the input and the trial's output bound are looked at on every trip, not only after a match
    if (on > max_out || b.tell() > end_bits) { rc = B_BAD; break; }
    if (on + 300 > ocap) { o.n = on; if (!o.reserve(on + (1u << 20))) { rc = B_BAD; break; } ob = o.b; ocap = o.cap; }
    b.refill();                                    // >= 56 bits: a literal/length code (15) + extra (5) + distance code (15) + extra (13) = 48
    int s = lit->decode(b);
    if (s < 256) {
      if (s < 0 || (text_only && !texty(s))) { rc = B_BAD; break; }
      ob[on++] = (uint16_t)s;
      // more literals from the same refill while they are there (most symbols of sequence data are literals; a refill holds at least three codes)
      bool bad = false;
      while (b.cnt >= 15) { const uint16_t e = lit->fast[b.buf & ((1u << FB) - 1)]; if (!e || (e >> 4) >= 256) break; if (text_only && !texty(e >> 4)) { bad = true; break; } b.drop(e & 15); ob[on++] = (uint16_t)(e >> 4); }
      if (bad) { rc = B_BAD; break; }
      continue;
    }
    if (s == 256) break;
    s -= 257; if (s >= 29) { rc = B_BAD; break; }
    const uint32_t ln = LBASE[s] + b.get(LEXT[s]);
    const int ds = dist->decode(b); if (ds < 0 || ds >= 30) { rc = B_BAD; break; }
    if (b.cnt < 13) b.refill();
    const uint32_t d = DBASE[ds] + b.get(DEXT[ds]);
    if (d > WIN) { rc = B_BAD; break; }
    uint16_t* dst = ob + on; const uint16_t* src = dst - d;       // the window symbols in front make every distance valid
#if defined(__AVX2__)
    // [r4] 16 symbols per copy where the source lies at least that far back (in sequence data it nearly always does: a match is a piece of an earlier
    // record); the last copy may run up to 15 symbols past the match — into space the next symbols overwrite (the 300-symbol margin above covers it)
    if (d >= 16) { for (uint32_t i = 0; i < ln; i += 16) _mm256_storeu_si256((__m256i*)(dst + i), _mm256_loadu_si256((const __m256i*)(src + i))); }
    else if (ln >= 24) {   // [r5] long match, short distance (see inflate_raw_bytes): the period over 16 symbols, stored at strides of the largest multiple of d that fits
      alignas(32) uint16_t pat[16]; for (uint32_t i = 0; i < 16; ++i) pat[i] = src[i % d];
      const __m256i pv = _mm256_load_si256((const __m256i*)pat); const uint32_t step = (16u / d) * d;
      for (uint32_t i = 0; i < ln; i += step) _mm256_storeu_si256((__m256i*)(dst + i), pv);
    }
    else
#endif
    for (uint32_t i = 0; i < ln; ++i) dst[i] = src[i];
    on += ln;
    if (on > max_out || b.tell() > end_bits) { rc = B_BAD; break; }
  }
  br = b; o.n = on;
  if (rc == B_BAD || br.tell() > end_bits || o.n > max_out) return B_BAD;
  return fin ? B_FINAL : B_MORE;
}

// [r4] A whole raw deflate stream whose output size is known (a BGZF member: pgz_inflate_raw) straight into bytes: the decoder above without the 16-bit
// symbols — nothing of the stream's past is unknown here.  dst has room for `cap` bytes + 32 (a match is copied 32 bytes at a time and may run past its
// end; the caller places streams one behind the other or leaves that slack).  Returns the bytes written, -1 for a damaged stream.
static long inflate_raw_bytes(const uint8_t* in, size_t n, char* dst, size_t cap) {
  Bits b(in, n); const uint64_t end_bits = (uint64_t)n * 8; size_t on = 0;
  for (;;) {
    b.refill();
    const uint32_t fin = b.get(1), type = b.get(2);
    if (type == 3) return -1;
    if (type == 0) {
      b.drop(b.cnt & 7); b.refill();
      const uint32_t ln = b.get(16); b.refill(); const uint32_t nl = b.get(16);
      if ((ln ^ nl) != 0xFFFFu) return -1;
      const uint64_t at = b.tell() >> 3;
      if (at + ln > n || on + ln > cap) return -1;
      memcpy(dst + on, in + at, ln); on += ln; b.seek((at + ln) * 8);
    } else {
      Huff dl, dd; const Huff* lit; const Huff* dist;
      if (type == 1) { lit = &fixed_tables().lit; dist = &fixed_tables().dist; }
      else { if (!read_dynamic(b, dl, dd, false)) return -1; lit = &dl; dist = &dd; }
      for (;;) {
        if (b.tell() > end_bits) return -1;
        b.refill();
        int s = lit->decode(b);
        if (s < 256) {
          if (s < 0 || on >= cap) return -1;
          dst[on++] = (char)s;
          while (b.cnt >= 15 && on < cap) { const uint16_t e = lit->fast[b.buf & ((1u << FB) - 1)]; if (!e || (e >> 4) >= 256) break; b.drop(e & 15); dst[on++] = (char)(e >> 4); }
          continue;
        }
        if (s == 256) break;
        s -= 257; if (s >= 29) return -1;
        const uint32_t ln = LBASE[s] + b.get(LEXT[s]);
        const int ds = dist->decode(b); if (ds < 0 || ds >= 30) return -1;
        if (b.cnt < 13) b.refill();
        const uint32_t d = DBASE[ds] + b.get(DEXT[ds]);
        if (d > on || on + ln > cap) return -1;            // a distance before the stream's first byte, or more output than the stream said it holds
        char* o = dst + on; const char* src = o - d;
#if defined(__AVX2__)
        if (d >= 32) { for (uint32_t i = 0; i < ln; i += 32) _mm256_storeu_si256((__m256i*)(o + i), _mm256_loadu_si256((const __m256i*)(src + i))); }
        else if (ln >= 24) {
          // [r5] a long match at a short distance — a run of one quality value, a homopolymer, a tandem repeat: the source overlaps the destination, and the
          // bytewise copy that handled it made such text inflate slower than zlib.  The period is laid out once over 32 bytes and stored at strides of
          // the largest multiple of the distance that fits (the last store may run up to 31 bytes past the match, as the long-distance copy may)
          alignas(32) char pat[64]; for (uint32_t i = 0; i < 32; ++i) pat[i] = src[i % d];
          const __m256i pv = _mm256_load_si256((const __m256i*)pat); const uint32_t step = (32u / d) * d;
          for (uint32_t i = 0; i < ln; i += step) _mm256_storeu_si256((__m256i*)(o + i), pv);
        }
        else
#endif
        for (uint32_t i = 0; i < ln; ++i) o[i] = src[i];
        on += ln;
      }
    }
    if (fin) break;
    if (b.tell() > end_bits) return -1;
  }
  return b.tell() > end_bits ? -1 : (long)on;
}

// bits [bit, bit + n) of the file without a reader (n <= 32)
inline uint32_t bits_at(const uint8_t* base, size_t nbytes, uint64_t bit, int n) {
  const size_t by = (size_t)(bit >> 3); uint64_t v = 0;
  if (by + 8 <= nbytes) memcpy(&v, base + by, 8); else for (size_t i = 0; by + i < nbytes && i < 8; ++i) v |= (uint64_t)base[by + i] << (8 * i);
  return (uint32_t)((v >> (bit & 7)) & ((1ull << n) - 1));
}

// first bit in [from, to) where a non-final dynamic block starts whose whole content is text and which is followed by another plausible
// block header; ~0 if none
uint64_t find_block(const uint8_t* base, size_t n, uint64_t from, uint64_t to) {
  Out trial;
  for (uint64_t bit = from; bit < to; ++bit) {
    const uint32_t h = bits_at(base, n, bit, 17);
    if ((h & 7u) != 4u) continue;                                        // BFINAL 0, BTYPE 10
    if (((h >> 3) & 31u) > 29u || ((h >> 8) & 31u) > 29u) continue;        // HLIT, HDIST
    { // the code-length code must be complete: a few operations that stop ~99 % of what got this far
      const int hclen = (int)((h >> 13) & 15u) + 4; int cnt[8] = {0};
      const uint64_t lo = bits_at(base, n, bit + 17, 30), hi = bits_at(base, n, bit + 47, 27);
      for (int i = 0; i < hclen; ++i) cnt[i < 10 ? (lo >> (3 * i)) & 7u : (hi >> (3 * (i - 10))) & 7u]++;
      int left = 1; for (int l = 1; l <= 7; ++l) { left <<= 1; left -= cnt[l]; if (left < 0) break; }
      if (left != 0) continue; }
    { Bits b1(base, n); b1.seek(bit + 3); Huff a, b; if (!read_dynamic(b1, a, b, true)) continue; }   // both code tables valid and complete
    Bits br(base, n); br.seek(bit);
    if (!trial.start()) return ~0ULL;                                     // out of memory: no block start is claimed
    if (inflate_block(br, trial, true, WIN + (8u << 20)) != B_MORE) continue;
    if (trial.n < WIN + 1024) continue;                                   // a real block of a large file holds kilobytes
    // the next header must parse too
    const uint64_t nx = br.tell(); const uint32_t t2 = bits_at(base, n, nx + 1, 2);
    if (t2 == 3) continue;
    if (t2 == 2) { Bits b2(base, n); b2.seek(nx + 3); Huff a, b; if (!read_dynamic(b2, a, b, true)) continue; }
    return bit;
  }
  return ~0ull;
}

// symbols -> bytes: a symbol with MARK set is "byte k of the 32 KB in front of the piece" (wv), anything else is the byte itself.  [r4] Past a piece's
// first few hundred KB no marked symbols are left (every copy of one is resolved text by then), so 32 symbols at a time are tested for a mark and narrowed
// with one pack when there is none: ~8 GB/s against 1.3 for the symbol-by-symbol loop
static inline void resolve_symbols(const uint16_t* s, size_t L, const uint8_t* wv, char* d) {
  size_t k = 0;
#if defined(__AVX2__)
  const __m256i mark = _mm256_set1_epi16((short)MARK);
  for (; k + 32 <= L; k += 32) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(s + k)), b = _mm256_loadu_si256((const __m256i*)(s + k + 16));
    if (_mm256_testz_si256(_mm256_or_si256(a, b), mark)) { _mm256_storeu_si256((__m256i*)(d + k), _mm256_permute4x64_epi64(_mm256_packus_epi16(a, b), 0xD8)); continue; }
    for (size_t j = k; j < k + 32; ++j) { const uint16_t v = s[j]; d[j] = (char)(v & MARK ? wv[v & (WIN - 1)] : (uint8_t)v); }
  }
#endif
  for (; k < L; ++k) { const uint16_t v = s[k]; d[k] = (char)(v & MARK ? wv[v & (WIN - 1)] : (uint8_t)v); }
}
struct Text { char* p = nullptr; size_t n = 0, cap = 0; ~Text() { free(p); }   // a piece's text; buffers go round (a fresh page is the expensive part)
  bool size(size_t want) { if (want > cap) { free(p); cap = want + want / 8 + 4096; p = (char*)malloc(cap); if (!p) { cap = 0; return false; }
      if (cap >= (4u << 20)) { const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095; (void)madvise((void*)a, (cap - (a - (uintptr_t)p)) & ~(size_t)4095, MADV_HUGEPAGE); } }   // fewer first-touch faults (as Out)
    n = want; return true; } };
struct Recycler { std::mutex mu; std::vector<std::unique_ptr<Text>> spare; };   // outlives the stream while text buffers are still out
struct alignas(128) Piece {
  uint64_t start = ~0ull, end = 0; Out out; int status = B_BAD; bool oom = false; bool ran = false;     // status of the LAST block decoded: B_MORE = stopped at a boundary
  std::unique_ptr<Text> text; uint32_t crc = 0;
};
}  // namespace

long pgz_inflate_raw(const uint8_t* deflate, size_t n, char* dst, size_t isize) { return inflate_raw_bytes(deflate, n, dst, isize); }

struct Round {   // what the decoding stage hands to the finishing stage
  std::vector<std::unique_ptr<Piece>> pc; unsigned T = 0; std::vector<unsigned> chain;
  bool final_seen = false, starts_member = false, at_end = false; uint64_t end_bit = 0; std::string err;
};

struct PgzStream {
  const uint8_t* base; size_t n; std::function<void(std::function<void()>)> submit; unsigned threads; size_t piece_bytes;
  // ---- stage 1 (its own thread): where the pieces start, their symbols, the chain; needs only the bit position the round before ended on
  uint64_t bitpos = 0; bool next_starts_member = true; unsigned alone = 0;   // alone: rounds left to run as one piece (no other piece found a block start: stored or binary data)
  std::thread decoder; std::mutex qmu; std::condition_variable cv_q; std::deque<std::unique_ptr<Round>> decoded; bool dec_done = false, finishing = false;
  // (stage 1 of round r + 1 beside stage 2 of round r was measured on the 256-thread host and dropped: both stages want the whole pool, the overlap bought nothing —
  // 9.5 vs 10.7 M pairs/s on 2 x 24 M reads — and the extra buffers cost page faults on short files)
  std::mutex pmu; std::vector<std::unique_ptr<Piece>> free_pc;               // pieces go round: their symbol buffers are reused
  // ---- stage 2 (its own thread): the 32 KB windows in chain order, text and checksums, the member's trailer; then the text is published
  std::vector<uint8_t> tail; uint32_t crc = 0; uint64_t mlen = 0;            // of the current member
  std::deque<std::unique_ptr<Text>> ready; size_t ready_off = 0; std::shared_ptr<Recycler> rec = std::make_shared<Recycler>();
  std::mutex mu; std::condition_variable cv_ready, cv_room; std::thread producer; bool stop = false, failed = false, eof = false; size_t ready_bytes = 0, ahead_bytes = 0;
  pgz_counters ctr{0, 0, 0, 0};
  std::string err;

  void parallel(unsigned k, const std::function<void(unsigned)>& fn) {
    if (k <= 1) { if (k) fn(0); return; }
    std::mutex m; std::condition_variable c; unsigned left = k;
    for (unsigned i = 0; i < k; ++i) submit([&, i] { fn(i); std::lock_guard<std::mutex> lk(m); if (--left == 0) c.notify_one(); });
    std::unique_lock<std::mutex> lk(m); c.wait(lk, [&] { return left == 0; });
  }
  // gzip member header at byte `at` (RFC 1952); false if there is none
  bool member_header(size_t at, size_t* data_off) const {
    if (at + 18 > n || base[at] != 0x1f || base[at + 1] != 0x8b || base[at + 2] != 8) return false;
    const uint8_t flg = base[at + 3]; size_t q = at + 10;
    if (flg & 4) { if (q + 2 > n) return false; q += 2 + ((size_t)base[q] | ((size_t)base[q + 1] << 8)); }
    if (flg & 8) { while (q < n && base[q]) ++q; ++q; }
    if (flg & 16) { while (q < n && base[q]) ++q; ++q; }
    if (flg & 2) q += 2;
    if (q >= n) return false;
    *data_off = q; return true;
  }
  bool begin_member(size_t at) { size_t off; if (!member_header(at, &off)) return false; bitpos = (uint64_t)off * 8; next_starts_member = true; return true; }

  // stage 1, one round: up to `threads` pieces from bitpos
  void decode_round(Round& R) {
    const bool timing = getenv("SQ_TIMING") != nullptr; auto tm = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) { if (!timing) return; auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[pgz] %-10s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t - tm).count()); tm = t; };
    const uint64_t nbits = (uint64_t)n * 8, pb = (uint64_t)piece_bytes * 8;
    const unsigned T = alone ? 1u : (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(threads, (nbits - bitpos + pb - 1) / pb));
    if (alone) --alone;
    R.T = T; R.starts_member = next_starts_member; next_starts_member = false;
    { std::lock_guard<std::mutex> lk(pmu); while (R.pc.size() < T) { if (!free_pc.empty()) { R.pc.push_back(std::move(free_pc.back())); free_pc.pop_back(); } else R.pc.emplace_back(new Piece()); } }
    auto& pc = R.pc;
    for (auto& p : pc) { p->start = ~0ull; p->end = 0; p->status = B_BAD; p->ran = false; p->crc = 0; p->oom = false; }
    pc[0]->start = bitpos;
    const uint64_t round_end = bitpos + (uint64_t)T * pb;
    parallel(T - 1, [&](unsigned k) { const unsigned i = k + 1; const uint64_t from = bitpos + (uint64_t)i * pb; if (from < nbits) pc[i]->start = find_block(base, n, from, std::min(from + pb, nbits)); });
    mark("sync");
    if (T > 1) { bool any = false; for (unsigned i = 1; i < T; ++i) any |= pc[i]->start != ~0ull; if (!any) alone = 16; }
    // a piece runs until it stands exactly on the start of a later piece (a later piece whose start it runs past was not on a block boundary),
    // the last one to the first boundary past the end of the round
    parallel(T, [&](unsigned i) {
      Piece& P = *pc[i]; if (P.start == ~0ull) return;
      P.ran = true; Bits br(base, n); br.seek(P.start);
      if (!P.out.start(piece_bytes * 5)) { P.status = B_BAD; P.end = P.start; P.oom = true; return; }   // sequence data inflates ~3-4 x: no regrowth on the way
      unsigned nxt = i + 1;
      for (;;) {
        const uint64_t pos = br.tell();
        while (nxt < T && (pc[nxt]->start == ~0ull || pc[nxt]->start < pos)) ++nxt;
        if (pos != P.start) {   // a piece decodes at least one block
          if (nxt < T && pc[nxt]->start == pos) { P.status = B_MORE; P.end = pos; return; }
          if (nxt >= T && pos >= round_end) { P.status = B_MORE; P.end = pos; return; }
        }
        const int rc = inflate_block(br, P.out, false, ~(size_t)0);
        if (rc != B_MORE) { P.status = rc; P.end = br.tell(); return; }
      }
    });
    mark("decode");
    // the chain of pieces that follow one another exactly
    unsigned cur = 0;
    for (;;) {
      Piece& P = *pc[cur]; R.chain.push_back(cur);
      if (P.status == B_BAD) { R.err = P.oom ? std::string("out of memory while inflating (SQ_READER_PGZ_PIECE sets the piece size)") : "corrupt deflate data near byte " + std::to_string(P.end / 8); return; }
      if (P.status == B_FINAL) { R.final_seen = true; break; }
      unsigned k = cur + 1; while (k < T && pc[k]->start != P.end) ++k;
      if (k >= T) break;
      cur = k;
    }
    R.end_bit = pc[R.chain.back()]->end;
    if (R.final_seen) {   // the trailer (checked by stage 2), then maybe another member
      const size_t at = (size_t)((R.end_bit + 7) / 8);
      if (at + 8 > n) { R.err = "truncated gzip member (no trailer)"; return; }
      if (at + 8 >= n || !begin_member(at + 8)) R.at_end = true;     // like gzip: whatever follows the last member is ignored
    } else if (R.end_bit >= nbits) { R.err = "truncated gzip file (the last block is missing)"; return; }
    else bitpos = R.end_bit;
  }
  void decode_loop() {
    for (;;) {
      { std::unique_lock<std::mutex> lk(qmu); cv_q.wait(lk, [&] { return stop || (decoded.empty() && !finishing); });   // one round ahead of stage 2 (or none)
        if (stop) { dec_done = true; cv_q.notify_all(); return; } }
      std::unique_ptr<Round> R(new Round()); decode_round(*R);
      const bool last = !R->err.empty() || R->at_end;
      { std::lock_guard<std::mutex> lk(qmu); decoded.push_back(std::move(R)); if (last) dec_done = true; }
      cv_q.notify_all();
      if (last) return;
    }
  }

  // stage 2, one round
  bool finish_round(Round& R) {
    const bool timing = getenv("SQ_TIMING") != nullptr; auto tm = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) { if (!timing) return; auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[pgz] %-10s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t - tm).count()); tm = t; };
    ctr.rounds++;
    if (R.starts_member) { tail.assign(WIN, 0); crc = (uint32_t)crc32(0L, Z_NULL, 0); mlen = 0; ctr.members++; }
    if (!R.err.empty() && R.chain.empty()) { err = R.err; return false; }
    auto& pc = R.pc; auto& chain = R.chain;
    if (!R.err.empty()) { err = R.err; return false; }
    for (unsigned i = 1; i < R.T; ++i) if (pc[i]->ran && std::find(chain.begin(), chain.end(), i) == chain.end()) ctr.resynced++;
    ctr.pieces += chain.size();
    // windows: the 32 KB in front of every piece of the chain (sequential, 32 K symbols each), then text + checksums in parallel
    std::vector<std::vector<uint8_t>> win(chain.size());
    std::vector<uint8_t> w = tail;
    for (size_t t = 0; t < chain.size(); ++t) {
      win[t] = w;
      const Out& o = pc[chain[t]]->out; const size_t L = o.n - WIN;
      std::vector<uint8_t> nw(WIN);
      if (L >= WIN) { const uint16_t* s = o.b + o.n - WIN; for (uint32_t k = 0; k < WIN; ++k) nw[k] = s[k] & MARK ? w[s[k] & (WIN - 1)] : (uint8_t)s[k]; }
      else { memcpy(nw.data(), w.data() + L, WIN - L); const uint16_t* s = o.b + WIN; for (size_t k = 0; k < L; ++k) nw[WIN - L + k] = s[k] & MARK ? w[s[k] & (WIN - 1)] : (uint8_t)s[k]; }
      w.swap(nw);
    }
    tail = w;
    { std::lock_guard<std::mutex> lk(rec->mu);
      for (unsigned t : chain) { if (!rec->spare.empty()) { pc[t]->text = std::move(rec->spare.back()); rec->spare.pop_back(); } else pc[t]->text.reset(new Text()); } }
    mark("windows");
    parallel((unsigned)chain.size(), [&](unsigned t) {
      Piece& P = *pc[chain[t]]; const size_t L = P.out.n - WIN;
      if (!P.text->size(L)) { P.status = B_BAD; return; }
      const uint16_t* s = P.out.b + WIN; const uint8_t* wv = win[t].data(); char* d = P.text->p;
      resolve_symbols(s, L, wv, d);
      P.crc = sqcrc::crc32((uint32_t)crc32(0L, Z_NULL, 0), d, L);   // [r4] by carry-less multiplication (crc32_fast.h): ~6 GB/s against zlib's 1
    });
    mark("text+crc");
    for (unsigned t : chain) { Piece& P = *pc[t]; if (P.status == B_BAD) { err = "out of memory"; return false; }
      crc = (uint32_t)crc32_combine(crc, P.crc, (z_off_t)P.text->n); mlen += P.text->n; }
    if (R.final_seen) {   // trailer: CRC-32 and length of the member
      const size_t at = (size_t)((R.end_bit + 7) / 8);
      const uint32_t fcrc = (uint32_t)base[at] | ((uint32_t)base[at + 1] << 8) | ((uint32_t)base[at + 2] << 16) | ((uint32_t)base[at + 3] << 24);
      const uint32_t flen = (uint32_t)base[at + 4] | ((uint32_t)base[at + 5] << 8) | ((uint32_t)base[at + 6] << 16) | ((uint32_t)base[at + 7] << 24);
      if (fcrc != crc || flen != (uint32_t)mlen) { err = "gzip checksum mismatch (CRC-32 or length of a member)"; return false; }
    }
    { std::lock_guard<std::mutex> lk(mu);   // the text becomes visible only once its member's trailer (if it ended here) has been checked
      if (R.at_end) eof = true;
      for (unsigned t : chain) { Piece& P = *pc[t]; if (P.text->n) { ready_bytes += P.text->n; ready.push_back(std::move(P.text)); } else { std::lock_guard<std::mutex> l2(rec->mu); rec->spare.push_back(std::move(P.text)); } } }
    { std::lock_guard<std::mutex> lk(pmu); for (auto& p : pc) free_pc.push_back(std::move(p)); pc.clear(); }
    return true;
  }
  void produce() {
    for (;;) {
      { std::unique_lock<std::mutex> lk(mu); cv_room.wait(lk, [&] { return stop || ready_bytes < ahead_bytes; }); if (stop) return; }
      std::unique_ptr<Round> R;
      { std::unique_lock<std::mutex> lk(qmu); cv_q.wait(lk, [&] { return stop || !decoded.empty() || dec_done; });
        if (stop) return;
        if (decoded.empty()) { std::lock_guard<std::mutex> l2(mu); eof = true; cv_ready.notify_all(); return; }   // the decoder stopped without a last round (cannot happen; do not hang)
        R = std::move(decoded.front()); decoded.pop_front(); finishing = true; }
      cv_q.notify_all();
      const bool ok = finish_round(*R);
      { std::lock_guard<std::mutex> lk(qmu); finishing = false; } cv_q.notify_all();
      std::lock_guard<std::mutex> lk(mu);
      if (!ok) failed = true;
      cv_ready.notify_all();
      if (!ok || eof) return;
    }
  }
};

PgzStream* pgz_open(const uint8_t* data, size_t bytes, std::function<void(std::function<void()>)> submit, unsigned threads, size_t piece_bytes) {
  std::unique_ptr<PgzStream> s(new PgzStream());
  s->base = data; s->n = bytes; s->submit = std::move(submit); s->threads = std::max(1u, threads); s->piece_bytes = std::max<size_t>(piece_bytes, 1u << 16);
  if (!s->begin_member(0)) return nullptr;
  s->ahead_bytes = (size_t)s->threads * s->piece_bytes * 4;          // about one round of text waiting while the next is finished
  PgzStream* p = s.release();
  p->decoder = std::thread([p] { p->decode_loop(); });
  p->producer = std::thread([p] { p->produce(); });
  return p;
}
long pgz_read(PgzStream* s, char* dst, size_t want, std::string* err) {
  size_t got = 0;
  std::unique_lock<std::mutex> lk(s->mu);
  while (got < want) {
    if (s->ready.empty()) {
      if (s->failed) { if (err) *err = s->err; return -1; }
      if (s->eof) break;
      s->cv_ready.wait(lk, [&] { return !s->ready.empty() || s->failed || s->eof; });
      continue;
    }
    Text& f = *s->ready.front();
    const size_t take = std::min(want - got, f.n - s->ready_off);
    lk.unlock(); memcpy(dst + got, f.p + s->ready_off, take); lk.lock();      // the front buffer is the consumer's until it is returned
    got += take; s->ready_off += take;
    if (s->ready_off == f.n) { s->ready_bytes -= f.n; { std::lock_guard<std::mutex> l2(s->rec->mu); s->rec->spare.push_back(std::move(s->ready.front())); } s->ready.pop_front(); s->ready_off = 0; s->cv_room.notify_one(); }
  }
  return (long)got;
}
int pgz_next(PgzStream* s, PgzBuf* out, std::string* err) {
  std::unique_lock<std::mutex> lk(s->mu);
  s->cv_ready.wait(lk, [&] { return !s->ready.empty() || s->failed || s->eof; });
  if (s->ready.empty()) { if (s->failed) { if (err) *err = s->err; return -1; } return 0; }
  Text* t = s->ready.front().release(); s->ready.pop_front(); s->ready_bytes -= t->n; s->cv_room.notify_one();
  std::shared_ptr<Recycler> rec = s->rec;
  out->p = t->p; out->n = t->n;
  out->hold = std::shared_ptr<void>((void*)t, [rec](void* v) { std::lock_guard<std::mutex> l(rec->mu); rec->spare.emplace_back((Text*)v); });
  return 1;
}
void pgz_close(PgzStream* s) {
  if (!s) return;
  { std::lock_guard<std::mutex> lk(s->mu); std::lock_guard<std::mutex> l2(s->qmu); s->stop = true; } s->cv_room.notify_all(); s->cv_q.notify_all();
  if (s->decoder.joinable()) s->decoder.join();
  if (s->producer.joinable()) s->producer.join();
  delete s;
}
pgz_counters pgz_stats(const PgzStream* s) { return s->ctr; }
