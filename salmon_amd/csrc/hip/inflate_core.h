// hip/inflate_core.h — [r5] one raw deflate stream of known output size (a BGZF member: at most 64 KB of text) decoded by ONE wave.
//
// Why: the GPU boxes give a process 16 cores' worth of CPU time, and 16 cores inflate ~10 GB/s of FASTQ text — a quarter of what the PCIe link
// moves for plain files (profiles/r05_reader_compressed.txt).  A BGZF file is ~130 000 independent members per 20 M reads, and the compressed bytes
// are a fifth to a half of the text: inflated on the device, such a file crosses PCIe faster than a plain one.
//
// How it runs on the device (hip/inflate_dev.hip): a WAVE per member.  A deflate stream is a serial chain — a code's length says where the next one starts —
// so the wave is a scalar processor: positions, counters and what steers the control flow live in scalar registers, the 64-bit bit buffer in two vector
// registers whose lanes all hold the same value (the scalar port, one instruction per CU and cycle for all its waves, is what bounds the kernel: shifts,
// masks and table addresses go to the vector ALUs), the input comes through a 256-byte LDS window that the lanes fetch a block ahead with one coalesced load.
// One look-up per symbol: 32-bit peek tables of LIT_BITS / DIST_BITS bits whose entries carry the bits to drop, the literal (two, where two codes fit the peek)
// or the base length / distance with its number of extra bits; longer codes are walked bit by bit through the canonical counts as in zlib's contrib/puff
// (RFC 1951).  The LANES are used where bytes move: symbols wait as tokens, a lane each; 64 of them are laid out by a prefix sum over their lengths and stored
// by all lanes at once (a literal its bytes, a short match from text already in memory its copy); matches that reach into the batch's own text, repeat
// themselves or are long are copied one after the other by the whole wave (lane i takes byte i mod distance of the window, which also lays out the period of
// an overlapping match); the CRC-32 is taken over 64 stretches at once and folded (x^(8n) mod P).  Memory operations of one wave complete in order, so a lane
// may read what another lane of its wave stored by an earlier instruction.  Parallelism across members does the rest: a batch of 10^6 read pairs is ~13 000
// members = 13 000 waves.  What each of these steps bought, and the counters behind "the scalar port": profiles/r05_inflate_kernel.txt.
//
// The same source compiles for the host (SQ_HD; the lane parts have plain loops there), which is how tests/test_inflate.py checks tables, bit reader and
// the CRC folding against zlib on machines without a GPU; the host build is a test hook (sq_debug_inflate_core_host), not a path of the product.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "../sq_internal.h"

namespace sqinf {

// a value every lane holds alike, said so to the compiler (it then lives in a scalar register and the arithmetic on it is scalar)
#if defined(__HIP_DEVICE_COMPILE__)
#define SQ_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define SQ_INL SQ_HD __attribute__((always_inline))
#else
#define SQ_UNI(x) ((uint32_t)(x))
#define SQ_INL SQ_HD
#endif

constexpr int LIT_BITS = 10, DIST_BITS = 8;
struct Huff { uint16_t count[16]; uint16_t symbol[288]; };                 // codes of each length; symbols in code order
struct HuffS { uint16_t count[16]; uint16_t symbol[32]; };                 // the same for the 30 distance codes and the 19 code-length codes
// per wave (LDS on the device).  The peek tables answer a code of at most LIT_BITS / DIST_BITS bits with everything the decoder wants to know, so that a
// symbol is ONE look-up (entry 0: the code is longer — walked through the canonical counts):
//   lit[bits]:  bits 0-3 = bits to drop; K_LIT: a literal in bits 8-15, with K_PAIR a second one in bits 16-23 (two codes that fit the peek together);
//               else a length symbol: bits 8-16 = base length (0: end of block, > 258: no such symbol), bits 17-19 = number of extra bits
//   dist[bits]: bits 0-3 = bits to drop, bits 4-7 = number of extra bits, bits 8-24 = base distance (65 537: no such symbol — farther back than any member is long)
// (the code-length code of a dynamic block borrows dist[]: bits 0-3, symbol in bits 8-15)
constexpr uint32_t K_LIT = 16, K_PAIR = 32;
struct Tables { uint32_t lit[1 << LIT_BITS]; uint32_t dist[1 << DIST_BITS]; Huff hlit; HuffS hdist, hclen; uint8_t lengths[320]; uint16_t offs[16]; uint32_t win[64]; };   // win: the input window (Bits)

enum { INF_OK = 0, INF_EOF_INPUT = 1, INF_BAD_BLOCK = 2, INF_BAD_STORED = 3, INF_BAD_CODES = 4, INF_BAD_SYMBOL = 5, INF_BAD_DISTANCE = 6, INF_OUTPUT_SIZE = 7 };

// the input, least significant bit first: up to 64 bits wait in (hi, lo).  Words of four bytes are read at aligned addresses (scalar loads on the device); the
// bytes in front of the first aligned word and behind the last whole one go one by one.  Past the end of the input the buffer yields zeros (counted
// in `virt`): the callers look at bad() where a wrong symbol could do harm (before a match is copied, before literals are stored, at the end of a block).
// On the device the two words live in VECTOR registers (every lane the same value) although they are uniform: a compute unit issues one scalar instruction per
// cycle for all its waves, and with 32 decoding waves that port is what bounds the kernel — the shifts and masks of the bit buffer and the address of the
// table look-up go to the vector ALUs, which idle otherwise; only what steers the control flow (the code's length, the symbol) comes back to a scalar register.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline uint32_t vector_zero() { uint32_t z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); return z; }     // 0, in a vector register and opaque to the compiler
__device__ inline uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t k) { return __builtin_amdgcn_alignbit(hi, lo, k); }   // bits k .. k+31 of hi:lo (k < 32)
#else
inline uint32_t vector_zero() { return 0; }
inline uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t k) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (k & 31)); }
#endif
// How the input reaches the bit buffer: in BLOCKS of 256 bytes.  A word read from memory where the decoder needs it costs a trip to HBM (1 - 2 us with
// thousands of waves asking) per four bytes of input, and the decoder can do nothing until it is back: measured, that was what a member took.  Instead the 64
// lanes fetch a block with one coalesced load, a word each, a block AHEAD of the one being decoded (the words wait in a register, `pend`); when the decoder
// enters a block its words go to the wave's window in LDS, the load of the next block is issued, and the bit buffer is fed from the window.
struct Bits {
  const uint8_t* p; uint32_t n, pos, nblk; uint32_t lo, hi; uint32_t cnt, virt; uint32_t* win;      // win: 64 words (LDS on the device); virt: how many of the cnt bits lie behind the end of the input
  uint32_t mis;                                                                                     // bytes between the aligned address p and the stream's first byte
  SQ_INL uint64_t tell() const { return 8ull * (uint64_t)(pos - mis) - (uint64_t)cnt; }             // bits of the stream consumed so far (meaningless once bad())
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t pend, lane;
  __device__ void fetch(uint32_t b) { uint32_t off = b * 256u + lane * 4u; const uint32_t top = (n - 1u) & ~3u; off = off < top ? off : top; pend = *(const uint32_t*)(p + off); }   // no branch: lanes behind the input's end re-read its last word (never used); up to 3 bytes behind n are read (the buffers have slack)
  __device__ void enter() { win[lane] = pend; ++nblk; fetch(nblk); }      // (n > 0 here: positions below n are all that is ever asked for)
  __device__ void start() { lane = __lane_id(); pend = 0; if (n) fetch(0); }
#else
  void enter() { for (uint32_t l = 0; l < 64; ++l) { uint32_t w = 0; for (uint32_t k = 0; k < 4; ++k) { const uint32_t off = nblk * 256u + l * 4u + k; if (off < n) w |= (uint32_t)p[off] << (8 * k); } win[l] = w; } ++nblk; }
  void start() {}
#endif
  SQ_INL uint32_t word() { if ((pos >> 8) == nblk) enter(); return win[(pos >> 2) & 63u]; }      // the word that holds byte `pos` (a vector value on the device)
  SQ_INL void add(uint32_t v, int at) { const uint64_t w = (uint64_t)v << at; lo |= (uint32_t)w; hi |= (uint32_t)(w >> 32); }     // at <= 32
  SQ_INL void init(const uint8_t* p_, size_t n_, uint32_t* win_) {
    // p: the aligned address at or below the stream's first byte, positions count from there
    mis = (uint32_t)((uintptr_t)p_ & 3); p = p_ - mis; n = (uint32_t)n_ + mis; pos = 0; nblk = 0; win = win_; lo = vector_zero(); hi = lo; cnt = 0; virt = 0;
    start();
    if (n > mis) { const uint32_t nb = (n < 4u ? n : 4u) - mis; uint32_t w = word() >> (8 * mis); if (nb < 4) w &= (1u << (8 * nb)) - 1; add(w, 0); cnt = 8 * nb; pos = n < 4u ? n : 4u; }
  }
  // behind it at least 33 bits are there: a length code with its extra bits, or a distance code with its.  Past the end of the input the bits are zeros that
  // count as `virt`: a decoder that has eaten into them sees bad() — looked at where a wrong symbol could do harm, not per symbol
  SQ_INL void refill() {
    if (cnt > 32) return;
    if (pos + 4u <= n) { add(word(), (int)cnt); cnt += 32; pos += 4; }                        // (all but the stream's last word)
    else if (pos < n) { const uint32_t nb = n - pos; const uint32_t w = word() & ((1u << (8 * nb)) - 1); add(w, (int)cnt); cnt += 8 * nb; pos += nb; }
    else { cnt += 32; virt += 32; }
  }
  SQ_INL uint32_t peek(int k) const { return lo & ((1u << k) - 1); }                     // (a vector value on the device)
  SQ_INL void drop(int k) { lo = funnel(hi, lo, (uint32_t)k); hi >>= k; cnt -= (uint32_t)k; }      // k < 32, k <= cnt (the callers refill first)
  SQ_INL uint32_t take(int k) { const uint32_t v = SQ_UNI(peek(k)); drop(k); return v; } // k <= 16, refilled by the caller
  SQ_INL uint32_t get(int k) { refill(); return take(k); }
  SQ_INL bool bad() const { return cnt < virt; }
};

// canonical Huffman table from code lengths (puff.c construct): < 0 over-subscribed, 0 complete, > 0 incomplete
// (offs: where the next symbol of each length goes — 16 slots borrowed from the caller's table memory, not registers: they are indexed by a variable)
template <class H> SQ_INL int build(H& h, const uint8_t* length, int n, uint16_t* offs) {
  for (int l = 0; l < 16; ++l) h.count[l] = 0;
#pragma unroll 1
  for (int s = 0; s < n; ++s) { const uint32_t l = SQ_UNI(length[s]); h.count[l] = (uint16_t)(SQ_UNI(h.count[l]) + 1); }
  if ((int)SQ_UNI(h.count[0]) == n) return 0;
  int left = 1;
#pragma unroll 1
  for (int l = 1; l < 16; ++l) { left <<= 1; left -= (int)SQ_UNI(h.count[l]); if (left < 0) return left; }
  offs[1] = 0;
#pragma unroll 1
  for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(SQ_UNI(offs[l]) + SQ_UNI(h.count[l]));
#pragma unroll 1
  for (int s = 0; s < n; ++s) { const uint32_t l = SQ_UNI(length[s]); if (l) { const uint32_t at = SQ_UNI(offs[l]); offs[l] = (uint16_t)(at + 1); h.symbol[at] = (uint16_t)s; } }
  return left;
}
// base | extra bits << 12 of the 29 length symbols; base | extra bits << 16 of the 30 distance symbols
SQ_HD uint32_t len_base_extra(uint32_t s) {
  const uint32_t lenx[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11 | 1 << 12, 13 | 1 << 12, 15 | 1 << 12, 17 | 1 << 12, 19 | 2 << 12, 23 | 2 << 12, 27 | 2 << 12, 31 | 2 << 12, 35 | 3 << 12, 43 | 3 << 12,
                             51 | 3 << 12, 59 | 3 << 12, 67 | 4 << 12, 83 | 4 << 12, 99 | 4 << 12, 115 | 4 << 12, 131 | 5 << 12, 163 | 5 << 12, 195 | 5 << 12, 227 | 5 << 12, 258};
  return SQ_UNI(lenx[s]);
}
SQ_HD uint32_t dist_base_extra(uint32_t s) {
  const uint32_t distx[30] = {1, 2, 3, 4, 5 | 1 << 16, 7 | 1 << 16, 9 | 2 << 16, 13 | 2 << 16, 17 | 3 << 16, 25 | 3 << 16, 33 | 4 << 16, 49 | 4 << 16, 65 | 5 << 16, 97 | 5 << 16, 129 | 6 << 16,
                              193 | 6 << 16, 257 | 7 << 16, 385 | 7 << 16, 513 | 8 << 16, 769 | 8 << 16, 1025 | 9 << 16, 1537 | 9 << 16, 2049 | 10 << 16, 3073 | 10 << 16, 4097 | 11 << 16,
                              6145 | 11 << 16, 8193 | 12 << 16, 12289 | 12 << 16, 16385 | 13 << 16, 24577 | 13 << 16};
  return SQ_UNI(distx[s]);
}
// what the tables say about a symbol whose code has `len` bits (len = 0: the bits are gone already — the symbol came from decode_long)
struct LitEntry { SQ_INL uint32_t operator()(uint32_t sym, uint32_t len) const {
  if (sym < 256) return len | K_LIT | (sym << 8);
  if (sym == 256) return len;                                          // end of block: base length 0
  if (sym < 286) { const uint32_t lx = len_base_extra(sym - 257); return len | ((lx & 0xFFF) << 8) | ((lx >> 12) << 17); }
  return len | (0x1FFu << 8); } };                                     // 286, 287 and NO_SYMBOL: in no valid stream
struct DistEntry { SQ_INL uint32_t operator()(uint32_t sym, uint32_t len) const {
  if (sym < 30) { const uint32_t dx = dist_base_extra(sym); return len | ((dx >> 16) << 4) | ((dx & 0xFFFF) << 8); }
  return len | (0x10001u << 8); } };                                   // no such symbol: a distance no member is long enough for
struct ClenEntry { SQ_INL uint32_t operator()(uint32_t sym, uint32_t len) const { return len | (sym << 8); } };
// the peek table of a code: every `bits`-bit pattern that starts with a code of at most `bits` bits (codes are sent most significant bit first, the
// input is read least significant bit first: the pattern is the reversed code, and every setting of the bits behind it)
template <class H, class E> SQ_INL void build_fast(const H& h, uint32_t* fast, int bits, E entry) {
  uint64_t* z = (uint64_t*)fast;
#pragma unroll 4
  for (int i = 0; i < (1 << bits) / 2; ++i) z[i] = 0;
  uint32_t code = 0, index = 0;
#pragma unroll 1
  for (int len = 1; len <= bits; ++len) {
    const uint32_t count = SQ_UNI(h.count[len]);
#pragma unroll 1
    for (uint32_t j = 0; j < count; ++j) {
      const uint32_t r = __builtin_bitreverse32(code + j) >> (32 - len);
      const uint32_t e = entry(SQ_UNI(h.symbol[index + j]), (uint32_t)len);
#pragma unroll 1
      for (uint32_t k = r; k < (1u << bits); k += 1u << len) fast[k] = e;
    }
    index += count; code = (code + count) << 1;
  }
}
// two literals whose codes fit the peek together become one entry.  From the high patterns down: entry i looks at entry i >> (its code's length), which lies
// below it and has not been rewritten yet (the device takes 64 entries at once, every lane reading before any lane writes)
SQ_INL uint32_t pair_entry(const uint32_t* lit, uint32_t i) {
  const uint32_t e1 = lit[i], l1 = e1 & 15;
  if (!(e1 & K_LIT) || l1 >= (uint32_t)LIT_BITS) return e1;
  const uint32_t e2 = lit[i >> l1], l2 = e2 & 15;                        // the pattern behind the first code: its upper l1 bits are unknown (zeros here)
  if (!(e2 & K_LIT) || e2 == 0 || l1 + l2 > (uint32_t)LIT_BITS) return e1;   // ... which matters only to a code that needs them
  return (l1 + l2) | K_LIT | K_PAIR | (e1 & 0xFF00u) | ((e2 & 0xFF00u) << 8);
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline void pair_literals(uint32_t* lit) {
  const uint32_t lane = __lane_id();
  for (int c = (1 << LIT_BITS) / 64 - 1; c >= 0; --c) { const uint32_t i = (uint32_t)c * 64u + lane; const uint32_t e = pair_entry(lit, i); lit[i] = e; }
}
#else
inline void pair_literals(uint32_t* lit) { for (int i = (1 << LIT_BITS) - 1; i >= 0; --i) lit[i] = pair_entry(lit, (uint32_t)i); }
#endif
// a symbol whose code the peek table does not hold, bit by bit through the canonical counts (puff.c decode).  NO_SYMBOL (not a symbol of any alphabet) on a
// code that is not in the table: the callers' range checks catch it, the loop needs no way out of its own for it
constexpr uint32_t NO_SYMBOL = 511;
template <class H> SQ_INL uint32_t decode_long(Bits& b, const H& h) {
  int code = 0, first = 0, index = 0;
#pragma unroll 1
  for (int len = 1; len <= 15; ++len) {
    code |= (int)b.take(1);
    const int count = (int)SQ_UNI(h.count[len]);
    if (code - count < first) return SQ_UNI(h.symbol[index + (code - first)]);
    index += count; first += count; first <<= 1; code <<= 1;
  }
  return NO_SYMBOL;
}
// one entry (the caller has refilled): the table's, or the one made for a symbol with a long code; its bits are dropped
template <class H, class E> SQ_INL uint32_t decode(Bits& b, const H& h, const uint32_t* fast, int bits, E entry) {
  uint32_t e = SQ_UNI(fast[b.peek(bits)]);
  if (e == 0) e = entry(decode_long(b, h), 0);
  b.drop((int)(e & 15));
  return e;
}

SQ_INL void fixed_tables(Tables& T) {
  int s = 0;
  for (; s < 144; ++s) T.lengths[s] = 8;
  for (; s < 256; ++s) T.lengths[s] = 9;
  for (; s < 280; ++s) T.lengths[s] = 7;
  for (; s < 288; ++s) T.lengths[s] = 8;
  (void)build(T.hlit, T.lengths, 288, T.offs); build_fast(T.hlit, T.lit, LIT_BITS, LitEntry()); pair_literals(T.lit);
  for (s = 0; s < 30; ++s) T.lengths[s] = 5;
  (void)build(T.hdist, T.lengths, 30, T.offs); build_fast(T.hdist, T.dist, DIST_BITS, DistEntry());
}
SQ_INL int dynamic_tables(Bits& b, Tables& T) {
  const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
  if (b.bad()) return INF_EOF_INPUT;
  if (nlen > 286 || ndist > 30) return INF_BAD_CODES;
  int i = 0;
  for (; i < ncode; ++i) T.lengths[SQ_UNI(order[i])] = (uint8_t)b.get(3);
  for (; i < 19; ++i) T.lengths[SQ_UNI(order[i])] = 0;
  if (b.bad()) return INF_EOF_INPUT;
  if (build(T.hclen, T.lengths, 19, T.offs) != 0) return INF_BAD_CODES;      // the code-length code must be complete
  build_fast(T.hclen, T.dist, 7, ClenEntry());                                // (its peek table borrows the distance table's place: that one is built below)
  i = 0;
  while (i < nlen + ndist) {
    b.refill();
    const uint32_t sym = decode(b, T.hclen, T.dist, 7, ClenEntry()) >> 8;
    if (b.bad()) return INF_EOF_INPUT;
    if (sym > 18) return INF_BAD_CODES;
    if (sym < 16) T.lengths[i++] = (uint8_t)sym;
    else {
      int len = 0, rep;
      if (sym == 16) { if (i == 0) return INF_BAD_CODES; len = (int)SQ_UNI(T.lengths[i - 1]); rep = 3 + (int)b.take(2); }
      else if (sym == 17) rep = 3 + (int)b.take(3);
      else rep = 11 + (int)b.take(7);
      if (b.bad()) return INF_EOF_INPUT;
      if (i + rep > nlen + ndist) return INF_BAD_CODES;
      while (rep--) T.lengths[i++] = (uint8_t)len;
    }
  }
  if (SQ_UNI(T.lengths[256]) == 0) return INF_BAD_CODES;                      // no end-of-block code
  int err = build(T.hlit, T.lengths, nlen, T.offs);
  if (err < 0 || (err > 0 && nlen - (int)SQ_UNI(T.hlit.count[0]) != 1)) return INF_BAD_CODES;
  build_fast(T.hlit, T.lit, LIT_BITS, LitEntry()); pair_literals(T.lit);
  // the distance lengths follow the literal/length lengths in the same array: shifted to its start for build
  for (int s = 0; s < ndist; ++s) T.lengths[s] = (uint8_t)SQ_UNI(T.lengths[nlen + s]);
  err = build(T.hdist, T.lengths, ndist, T.offs);
  if (err < 0 || (err > 0 && ndist - (int)SQ_UNI(T.hdist.count[0]) != 1)) return INF_BAD_CODES;
  build_fast(T.hdist, T.dist, DIST_BITS, DistEntry());
  return INF_OK;
}

// where the text goes.  Device: the decoder does not touch memory per symbol.  What it decodes — a literal entry as the table gave it, or a match as
// 1 << 31 | length << 16 | distance — waits as a TOKEN in `tokv`, lane k holding token k; at 64 tokens (or the end of a block) apply() lays them out
// with one prefix sum over their lengths and all lanes store at once: a literal its byte(s), a match of up to 16 bytes whose source lies in text already in
// memory its copy (the usual case in FASTQ text: short repeats of sequence far back).  Matches that reach into the batch's own text, repeat themselves or are
// longer go one after the other behind that, in token order, the whole wave copying each (lane i takes byte i mod distance).  Per symbol the scalar unit — a
// compute unit issues one scalar instruction per cycle for all its waves, and that port is what bounds this kernel — sees a table entry, a shift count and a
// counter; the exec-mask work of a store is paid once per 64 tokens.  Host: bytes, in order.  Nothing is ever stored at or behind `cap` (the next member's text
// begins there); whether the stream wanted to is seen from size().
#if defined(__HIP_DEVICE_COMPILE__)
// the tokens of a batch into memory (Out::apply).  NOT inlined: it is the only code with lane-dependent branches, and with it out of the way the decoding loop
// is a region of uniform branches that the compiler leaves as plain scalar jumps instead of structuring it into flags and masks
// element idx of a span's output; [r6] symbols only: in front of the output lies the unknown window, whose byte k is the symbol SYM_MARK | k (k = SPAN_WINDOW + idx)
template <class T> __device__ inline T span_src(const T* out, int32_t idx) {
  if constexpr (sizeof(T) == 2) { if (idx < 0) return (T)(0x8000u | (uint32_t)(32768 + idx)); }
  return out[idx];
}
// Memory ordering: a match reads text that OTHER lanes of this wave stored a few instructions earlier (the literals and simple matches above it, the round of the
// `rest` loop before it).  On gfx9 / CDNA a wave's vector stores and loads to the same address are served in issue order by the one vector-memory pipe and its L1
// (one counter, vmcnt, for both): a later load of the wave sees an earlier store of the wave.  gfx10 and later count stores separately (vscnt) and give no such
// order without a wait — this file is built for gfx950 only and says so:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "inflate_core.h: apply_tokens relies on gfx9's in-order vector memory pipe (see the note above it)"
#endif
template <class T>
__device__ __attribute__((noinline)) void apply_tokens(T* out, uint32_t on, uint32_t ntok, uint32_t tokv) {
  const uint32_t lane = __lane_id();
  const uint32_t t = lane < ntok ? tokv : 0u; const bool m = (t >> 31) != 0;
  const uint32_t len = m ? (t >> 16) & 0x1FFu : (t & K_LIT) ? ((t & K_PAIR) ? 2u : 1u) : 0u;
  uint32_t inc = len;                                                              // where each token's text begins: a prefix sum across the lanes
  for (int d = 1; d < 64; d <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= (uint32_t)d) inc += up; }
  const uint32_t pos = on + inc - len, dist = t & 0xFFFFu;
  const bool simple = m && len <= 16 && pos + len <= on + dist;                    // its source is in memory already (possibly in front of `out`: the window of a span, [r6])
  if (!m) { if (len >= 1) out[pos] = (T)((t >> 8) & 0xFFu); if (len == 2) out[pos + 1] = (T)((t >> 16) & 0xFFu); }
  if (simple) {
    const int32_t s0 = (int32_t)pos - (int32_t)dist; T* dst = out + pos; T by[16];
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) if (k < len) by[k] = span_src<T>(out, s0 + (int32_t)k);
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) if (k < len) dst[k] = by[k];
  }
  uint64_t rest = __ballot(m && !simple);
  while (rest) {                                                                   // (uniform: every lane sees the same mask)
    const int k = __builtin_ctzll(rest); rest &= rest - 1;
    const uint32_t tk = (uint32_t)__builtin_amdgcn_readlane((int)t, k), pk = (uint32_t)__builtin_amdgcn_readlane((int)pos, k), dk = tk & 0xFFFFu, lk = (tk >> 16) & 0x1FFu;
    const int32_t s0 = (int32_t)pk - (int32_t)dk; T* dst = out + pk;
    if (dk >= lk) { for (uint32_t i = lane; i < lk; i += 64) dst[i] = span_src<T>(out, s0 + (int32_t)i); }
    else { for (uint32_t i = lane; i < lk; i += 64) dst[i] = span_src<T>(out, s0 + (int32_t)(i % dk)); }        // the window's last `dist` bytes, repeated
  }
}
#endif
// T: uint8_t (text) or, [r6], uint16_t (the symbols of a span of a gzip stream whose preceding 32 KB are not known yet: 0..255 = a byte, 0x8000 | k = byte k of that window)
template <class T> struct OutT {
  T* out; uint32_t on, pend, ntok, cap;      // on: elements in memory; pend: elements the waiting tokens stand for
  bool dry = false;                          // [r6] a trial of a candidate block start: everything is decoded and counted, nothing is stored
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t lane, tokv;
  __device__ void init(T* o, uint32_t cap_) { out = o; on = 0; pend = 0; ntok = 0; cap = cap_; tokv = 0; lane = __lane_id(); }
  __device__ void token(uint32_t t, uint32_t nbytes) { tokv = lane == ntok ? t : tokv; ++ntok; pend += nbytes; }      // the caller applies at 64
  __device__ void apply() { if (!dry) apply_tokens<T>(out, on, ntok, tokv); on += pend; pend = 0; ntok = 0; }      // the caller has checked size() <= cap
  __device__ void lit(uint32_t e) { token(e, (e & K_PAIR) ? 2u : 1u); }
  __device__ void match(uint32_t dist, uint32_t len) { token(0x80000000u | (len << 16) | dist, len); }   // the caller has checked dist <= size() (+ the window), size() + len <= cap
#else
  void init(T* o, uint32_t cap_) { out = o; on = 0; pend = 0; ntok = 0; cap = cap_; }
  void apply() { ntok = 0; }
  void lit(uint32_t e) { if (on < cap && !dry) out[on] = (T)((e >> 8) & 0xFFu); ++on; if (e & K_PAIR) { if (on < cap && !dry) out[on] = (T)((e >> 16) & 0xFFu); ++on; } ++ntok; }
  void match(uint32_t dist, uint32_t len) { if (!dry) { const int32_t s0 = (int32_t)on - (int32_t)dist; T* dst = out + on;
      for (uint32_t i = 0; i < len; ++i) { const int32_t idx = s0 + (int32_t)(i % dist); dst[i] = (sizeof(T) == 2 && idx < 0) ? (T)(0x8000u | (uint32_t)(32768 + idx)) : out[idx]; } } on += len; ++ntok; }
#endif
  SQ_HD uint32_t size() const { return on + pend; }
};
typedef OutT<uint8_t> Out;

// the whole stream: `isize` bytes of output are expected (a BGZF member's trailer says how many).  Returns INF_OK or what was wrong.
SQ_INL int inflate_member(const uint8_t* in, size_t n, uint8_t* out, uint32_t isize, Tables& T) {
  if (n == 0) return INF_EOF_INPUT;   // (Bits::init counts positions from the aligned address below `in`: with no input at all the bytes in front of it would pass for the stream)
  Bits b; b.init(in, n, T.win); Out o; o.init(out, isize); int rc = INF_OK;
  for (;;) {
    const uint32_t last = b.get(1), type = b.take(2);
    if (b.bad()) return INF_EOF_INPUT;
    if (type == 3) return INF_BAD_BLOCK;
    if (type == 0) {
      b.take((int)(b.cnt & 7));                                       // to the next byte boundary; the buffer then holds whole bytes
      const uint32_t len = b.get(16), nlen = b.get(16);
      if (b.bad()) return INF_EOF_INPUT;
      if ((len ^ nlen) != 0xFFFFu) return INF_BAD_STORED;
      if (o.size() + len > isize) return INF_OUTPUT_SIZE;
#pragma unroll 1
      for (uint32_t i = 0; i < len; ++i) { const uint32_t v = b.get(8); if (b.bad()) return INF_EOF_INPUT; o.lit(K_LIT | (v << 8)); if (o.ntok == 64) o.apply(); }
    } else {
      if (type == 1) fixed_tables(T);
      else { rc = dynamic_tables(b, T); if (rc) return rc; }
      // the loop that does the work: one table entry (a literal, two, or a length symbol followed by its distance) per turn; every way out of it leaves through
      // the one test behind it.  Whether the input ended or the text overflows is looked at per batch of tokens and per match, not per literal (zeros decode to
      // something harmless until then: nothing is stored at or behind the member's end)
#pragma unroll 1
      for (;;) {
        b.refill();
        const uint32_t e = decode(b, T.hlit, T.lit, LIT_BITS, LitEntry());
        if (e & K_LIT) o.lit(e);
        else {
          const uint32_t base = (e >> 8) & 0x1FFu;
          if (base == 0) { if (b.bad()) rc = INF_EOF_INPUT; break; }       // the end of the block
          if (base > 258) { rc = INF_BAD_SYMBOL; break; }
          const uint32_t len = base + b.take((int)((e >> 17) & 7u));
          b.refill();
          const uint32_t d = decode(b, T.hdist, T.dist, DIST_BITS, DistEntry());
          const uint32_t dist = (d >> 8) + b.take((int)((d >> 4) & 15u));
          // (whether the input had ended is looked at before the tokens are applied: a length or distance decoded from the zeros behind it is held to the same two tests)
          if (dist > o.size() || o.size() + len > isize) { rc = b.bad() ? INF_EOF_INPUT : (d >> 8) > 0x10000u ? INF_BAD_SYMBOL : o.size() + len > isize ? INF_OUTPUT_SIZE : INF_BAD_DISTANCE; break; }
          o.match(dist, len);
        }
        if (o.ntok == 64) { if (b.bad() || o.size() > isize) { rc = b.bad() ? INF_EOF_INPUT : INF_OUTPUT_SIZE; break; } o.apply(); }
      }
      if (rc) return rc;
    }
    if (last) break;
  }
  if (b.bad()) return INF_EOF_INPUT;
  if (o.size() != isize) return INF_OUTPUT_SIZE;
  o.apply();
  return INF_OK;
}

// ---- [r6] a SPAN of an ordinary gzip stream (hip/gzip_dev.hip) ------------------------------------------------------------------------------------------------------
// A plain .gz file is one deflate stream: where its blocks begin is only known by decoding, and a block may copy from the 32 KB of text before it.  As in host/pgzip.cpp
// (after pugz, Kerbiriou & Chikhi 2019) the stream is cut at block starts FOUND by trying bit offsets (find_block_start), every span between two found starts is decoded
// by a wave into 16-bit symbols in which "byte k of the 32 KB in front of the span" is a symbol of its own (the span's output is preceded by SPAN_WINDOW such symbols, so a
// copy out of the unknown window is an ordinary copy), and the windows are resolved afterwards, span after span.  Nothing is taken on trust: a span must end exactly on
// the bit where the next one was found to start, and the member's CRC-32 and length are checked against its trailer (gzip_dev.hip).
constexpr uint32_t SPAN_WINDOW = 32768, SYM_MARK = 0x8000u;
enum { INF_OVERRUN = 9 /* a block ended behind the bit where the next span starts: that start was not a block boundary */, INF_NOT_TEXT = 10 };
SQ_INL bool text_byte(uint32_t c) { return c == 10u || c == 13u || c == 9u || (c >= 32u && c < 127u); }
// from bit `start_bit` of `in` (n bytes in all) to the block boundary at `stop_bit` (~0: none) or to the end of a final block, whichever comes first.
// T = uint16_t: symbols, `out` preceded by SPAN_WINDOW marker symbols (window = SPAN_WINDOW); T = uint8_t with window = 0: plain text.
// max_sym > 0: a TRIAL of a candidate block start — stops (INF_OK) behind max_sym symbols, every literal must be a byte a FASTA / FASTQ file can hold.
template <class T> SQ_INL int inflate_span(const uint8_t* in, size_t n, uint64_t start_bit, uint64_t stop_bit, T* out, uint32_t cap, uint32_t window, Tables& Tb,
                                           uint32_t* out_n, uint64_t* end_bit, uint32_t* ended_final, uint32_t max_sym) {
  const size_t sb = (size_t)(start_bit >> 3); *ended_final = 0; *out_n = 0; *end_bit = start_bit;
  if (sb >= n) return INF_EOF_INPUT;
  Bits b; b.init(in + sb, n - sb, Tb.win); OutT<T> o; o.init(out, cap); o.dry = max_sym != 0; int rc = INF_OK; uint32_t nsym = 0;
  if (start_bit & 7) (void)b.get((int)(start_bit & 7));
  const uint64_t base = (uint64_t)sb * 8ull;
  for (;;) {
    if (!b.bad() && stop_bit != ~0ull) { const uint64_t at = base + b.tell(); if (at == stop_bit) break; if (at > stop_bit) return INF_OVERRUN; }
    const uint32_t last = b.get(1), type = b.take(2);
    if (b.bad()) return INF_EOF_INPUT;
    if (type == 3) return INF_BAD_BLOCK;
    if (type == 0) {
      b.take((int)(b.cnt & 7));
      const uint32_t len = b.get(16), nlen = b.get(16);
      if (b.bad()) return INF_EOF_INPUT;
      if ((len ^ nlen) != 0xFFFFu) return INF_BAD_STORED;
      if (o.size() + len > cap) return INF_OUTPUT_SIZE;
#pragma unroll 1
      for (uint32_t i = 0; i < len; ++i) { const uint32_t v = b.get(8); if (b.bad()) return INF_EOF_INPUT; if (max_sym && !text_byte(v)) return INF_NOT_TEXT; o.lit(K_LIT | (v << 8)); if (o.ntok == 64) o.apply(); }
      nsym += len;
    } else {
      if (type == 1) fixed_tables(Tb);
      else { rc = dynamic_tables(b, Tb); if (rc) return rc; }
#pragma unroll 1
      for (;;) {
        b.refill();
        const uint32_t e = decode(b, Tb.hlit, Tb.lit, LIT_BITS, LitEntry());
        if (e & K_LIT) { if (max_sym && (!text_byte((e >> 8) & 0xFFu) || ((e & K_PAIR) && !text_byte((e >> 16) & 0xFFu)))) { rc = INF_NOT_TEXT; break; } o.lit(e); }
        else {
          const uint32_t bl = (e >> 8) & 0x1FFu;
          if (bl == 0) { if (b.bad()) rc = INF_EOF_INPUT; break; }
          if (bl > 258) { rc = INF_BAD_SYMBOL; break; }
          const uint32_t len = bl + b.take((int)((e >> 17) & 7u));
          b.refill();
          const uint32_t d = decode(b, Tb.hdist, Tb.dist, DIST_BITS, DistEntry());
          const uint32_t dist = (d >> 8) + b.take((int)((d >> 4) & 15u));
          if (dist > o.size() + window || dist > SPAN_WINDOW || o.size() + len > cap) { rc = b.bad() ? INF_EOF_INPUT : (d >> 8) > 0x10000u ? INF_BAD_SYMBOL : o.size() + len > cap ? INF_OUTPUT_SIZE : INF_BAD_DISTANCE; break; }
          o.match(dist, len);
        }
        ++nsym;
        if (o.ntok == 64) { if (b.bad() || o.size() > cap) { rc = b.bad() ? INF_EOF_INPUT : INF_OUTPUT_SIZE; break; } o.apply(); if (max_sym && nsym >= max_sym) break; }
      }
      if (rc) return rc;
    }
    if (max_sym && nsym >= max_sym) break;
    if (last) { *ended_final = 1; break; }
  }
  if (b.bad()) return INF_EOF_INPUT;
  if (o.size() > cap) return INF_OUTPUT_SIZE;
  o.apply();
  *out_n = o.size(); *end_bit = base + b.tell();
  return INF_OK;
}
// bits [bit, bit + k) of the stream, k <= 32, straight from memory (positions behind the end read as zeros)
SQ_INL uint32_t peek_bits(const uint8_t* in, size_t n, uint64_t bit, int k) {
  const size_t by = (size_t)(bit >> 3); uint64_t w = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) w |= (by + i < n ? (uint64_t)in[by + i] : 0ull) << (8 * i);
  return (uint32_t)((w >> (bit & 7)) & (k == 32 ? 0xFFFFFFFFull : ((1ull << k) - 1)));
}
// what a block start must look like before it is worth a trial: not the final block, dynamic codes, counts in range, and a COMPLETE code-length code (Kraft sum = 1)
SQ_INL bool plausible_dynamic_header(const uint8_t* in, size_t n, uint64_t bit) {
  // the 74 bits of header a dynamic block starts with, from two unaligned 8-byte reads (the caller's buffer has 16 bytes of slack behind n)
  const size_t by = (size_t)(bit >> 3); if (by + 10 > n) return false;
  uint64_t w0, w1; __builtin_memcpy(&w0, in + by, 8); __builtin_memcpy(&w1, in + by + 8, 8);
  const uint32_t sh = (uint32_t)(bit & 7); const uint64_t lo = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0, hi = w1 >> sh;
  const uint32_t h = (uint32_t)lo & 0x1FFFFu;
  if ((h & 7u) != 4u) return false;                                   // BFINAL = 0, BTYPE = 2 (sent least significant bit first: 0, then 01 -> value 0b100)
  const uint32_t hlit = (h >> 3) & 31u, hdist = (h >> 8) & 31u, hclen = ((h >> 13) & 15u) + 4u;
  if (hlit > 29u || hdist > 29u) return false;
  uint32_t kraft = 0, used = 0;
#pragma unroll
  for (uint32_t i = 0; i < 19u; ++i) {
    const uint32_t o = 17u + 3u * i;
    const uint32_t l = i < hclen ? (uint32_t)((o < 64u ? (lo >> o) | (o > 61u ? hi << (64u - o) : 0ull) : hi >> (o - 64u)) & 7ull) : 0u;
    if (l) { kraft += 128u >> l; ++used; }
  }
  return kraft == 128u && used >= 2u;
}
// the first bit in [lo, hi) at which a non-final dynamic block starts whose first 512 symbols decode and are text; ~0 if there is none.  A trial stores nothing (its copies
// may reach back into the unknown window); whether the start is real is settled later: the span in front of it must end exactly there (inflate_span: INF_OVERRUN)
SQ_INL uint64_t find_block_start(const uint8_t* in, size_t n, uint64_t lo, uint64_t hi, Tables& Tb) {
  typedef uint16_t T; T* const scratch = nullptr; const uint32_t scratch_cap = 0x7FFFFFFFu;
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t lane = __lane_id();
  for (uint64_t base = lo; base < hi; base += 64) {
    const uint64_t bit = base + lane;
    uint64_t cand = __ballot(bit < hi && plausible_dynamic_header(in, n, bit));
    while (cand) {                                                     // (uniform)
      const int k = __builtin_ctzll(cand); cand &= cand - 1;
      uint32_t on, fin; uint64_t eb;
      if (inflate_span<T>(in, n, base + (uint64_t)k, ~0ull, scratch, scratch_cap, SPAN_WINDOW, Tb, &on, &eb, &fin, 512u) == INF_OK) return base + (uint64_t)k;
    }
  }
#else
  for (uint64_t bit = lo; bit < hi; ++bit) {
    if (!plausible_dynamic_header(in, n, bit)) continue;
    uint32_t on, fin; uint64_t eb;
    if (inflate_span<T>(in, n, bit, ~0ull, scratch, scratch_cap, SPAN_WINDOW, Tb, &on, &eb, &fin, 512u) == INF_OK) return bit;
  }
#endif
  return ~0ull;
}

// CRC-32 (the gzip polynomial, reflected) of n bytes, a byte at a time through a 256-entry table
SQ_HD uint32_t crc32_bytes(const uint32_t* table, const uint8_t* p, uint32_t n) {
  uint32_t c = 0xFFFFFFFFu;
  for (uint32_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
SQ_HD uint32_t crc32_entry(uint32_t i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1); return c; }
// a(x) * b(x) mod P in the reflected representation (bit 31 = x^0), and x^(8n) mod P: crc(A || B) = crc(A) * x^(8 |B|) + crc(B), which is how the CRCs
// of 64 stretches of a member, taken by 64 lanes at once, fold into the member's (zlib's crc32_combine states the same identity)
SQ_HD uint32_t crc_mul(uint32_t a, uint32_t b) {
  uint32_t p = 0;
#pragma unroll 1
  for (int i = 0; i < 32; ++i) { if (a & (0x80000000u >> i)) p ^= b; b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1; }
  return p;
}
SQ_HD uint32_t crc_xpow8(uint32_t n) {
  uint32_t r = 0x80000000u, sq = 0x00800000u;      // x^0; x^8
  for (; n; n >>= 1) { if (n & 1) r = crc_mul(sq, r); sq = crc_mul(sq, sq); }
  return r;
}
// the stretches: lane 0 takes n / 64 + n % 64 bytes, every other lane n / 64
SQ_HD void crc_stretch(uint32_t n, uint32_t lane, uint32_t* start, uint32_t* len) {
  const uint32_t q = n / 64, first = q + n % 64;
  *start = lane ? first + (lane - 1) * q : 0; *len = lane ? q : first;
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline uint32_t crc32_wave(const uint32_t* table, const uint8_t* p, uint32_t n) {
  const uint32_t lane = __lane_id(); uint32_t start, len; crc_stretch(n, lane, &start, &len);
  uint32_t c = crc32_bytes(table, p + start, len), X = crc_xpow8(n / 64);
  for (uint32_t s = 1; s < 64; s <<= 1) {       // lanes i and i + s hold the CRCs of s stretches each: behind the step lane i holds that of 2 s
    const uint32_t other = (uint32_t)__shfl_down((int)c, (int)s);
    if ((lane & (2 * s - 1)) == 0) c = crc_mul(X, c) ^ other;
    X = crc_mul(X, X);
  }
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
}
#else
inline uint32_t crc32_wave(const uint32_t* table, const uint8_t* p, uint32_t n) {      // the same stretches, folded left to right
  const uint32_t X = crc_xpow8(n / 64); uint32_t c = 0;
  for (uint32_t lane = 0; lane < 64; ++lane) { uint32_t start, len; crc_stretch(n, lane, &start, &len); const uint32_t ci = crc32_bytes(table, p + start, len); c = lane ? crc_mul(X, c) ^ ci : ci; }
  return c;
}
#endif

}  // namespace sqinf
