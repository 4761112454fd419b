// hip/gzip_dev.hip — [r6] an ordinary gzip file inflated on the device ("parallel gzip decode"; the reference reads .gz through one zlib stream per
// file, include/salmon/internal/io/FastxReader.hpp:13-32, and its parser threads are what bounds it, src/quant/SalmonQuantify.cpp:2419-2443).
//
// A gzip member is ONE deflate stream: no table says where its blocks begin, and a block may copy from the 32 KB of text in front of it.  The scheme is host/pgzip.cpp's
// (after pugz, Kerbiriou & Chikhi 2019), laid out for the device.  The file is taken in SEGMENTS of compressed bytes (64 MB: some four thousand spans — a launch that fills
// the chip); per segment
//   k_gz_find       a wave per 16 KB of compressed bytes looks for the first bit at which a dynamic block starts (64 bit offsets are tested at a time, a lane each: header
//                   fields in range and a COMPLETE code-length code; a candidate that passes is decoded for 512 symbols by the whole wave, which must all be text);
//   k_gz_decode     a wave per SPAN — from one found start to the next — decodes into 16-bit symbols (inflate_core.h: inflate_span): a byte, or "byte k of the 32 KB in
//                   front of this span", unknown at this point; a span must end exactly on the bit where the next one starts (a found start that was no block boundary is
//                   struck out and the segment decoded again);
//   k_gz_tails + k_gz_chain   ONE block walks the spans in order: the window in front of span u + 1 is the last 32 KB of (window of span u, span u resolved with it);
//   k_gz_translate  symbols -> bytes, every span with its window, into the caller's text buffer;   k_gz_crc: a wave per 64 KB of that text takes its CRC-32, the host folds
//                   them (one multiplication mod P each) and holds every member to its trailer (CRC-32 and length).
// A member that ends inside a segment ends the segment (its trailer and the next member's header are read by the host).  The last found start of a segment begins the
// next one: its span ends where the next segment finds its first start.
#include <hip/hip_runtime.h>
#include <zlib.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../host/index.h"
#include "inflate_core.h"
#include "gzip_dev.h"

namespace {
constexpr int GZ_WAVES = 4;                 // waves (sub-chunks / spans) per block
constexpr uint32_t GZ_SUB = 16384;          // compressed bytes a searching wave looks through
constexpr uint32_t GZ_TILE = 8192;          // symbols per block of the translation
struct GzUnit { uint64_t start_bit, stop_bit; uint64_t sym_off; uint32_t cap, _pad; };
struct GzUnitOut { uint64_t end_bit; uint32_t n_sym; uint32_t rc, final_, _pad; };

__global__ void __launch_bounds__(64 * GZ_WAVES) k_gz_find(const uint8_t* __restrict__ comp, uint32_t n, uint64_t first_bit, uint32_t nsub, uint64_t* __restrict__ found) {
  __shared__ sqinf::Tables s_tab[GZ_WAVES];
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), w = blockIdx.x * GZ_WAVES + wave;
  if (w >= nsub) return;
  uint64_t lo = (uint64_t)w * GZ_SUB * 8ull, hi = std::min<uint64_t>((uint64_t)(w + 1) * GZ_SUB * 8ull, (uint64_t)n * 8ull);
  if (lo < first_bit) lo = first_bit;
  const uint64_t f = lo < hi ? sqinf::find_block_start(comp, n, lo, hi, s_tab[wave]) : ~0ull;
  if ((threadIdx.x & 63) == 0) found[w] = f;
}
__global__ void __launch_bounds__(64 * GZ_WAVES) k_gz_decode(const uint8_t* __restrict__ comp, uint32_t n, const GzUnit* __restrict__ units, uint32_t nunits, uint16_t* __restrict__ sym,
                                                              GzUnitOut* __restrict__ out) {
  __shared__ sqinf::Tables s_tab[GZ_WAVES];
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), u = blockIdx.x * GZ_WAVES + wave;
  if (u >= nunits) return;
  const GzUnit U = units[u]; uint32_t on = 0, fin = 0; uint64_t eb = 0;
  const int rc = sqinf::inflate_span<uint16_t>(comp, n, U.start_bit, U.stop_bit, sym + U.sym_off, U.cap, sqinf::SPAN_WINDOW, s_tab[wave], &on, &eb, &fin, 0u);
  if ((threadIdx.x & 63) == 0) { GzUnitOut o; o.end_bit = eb; o.n_sym = on; o.rc = (uint32_t)rc; o.final_ = fin; o._pad = 0; out[u] = o; }
}
// What a span does to the window, as symbols: tails[u][j] = what lies at place j of the 32 KB in front of span u + 1, in terms of the 32 KB in front of span u — the span's
// own symbol (a byte, or a marker into that window), or, for a span shorter than the window, "byte n + j of it".  One block per span; the chain below then reads fixed addresses.
__global__ void __launch_bounds__(256) k_gz_tails(const GzUnit* __restrict__ units, const GzUnitOut* __restrict__ uo, const uint16_t* __restrict__ sym, uint16_t* __restrict__ tails) {
  constexpr uint32_t WN = sqinf::SPAN_WINDOW;
  const uint32_t u = blockIdx.x, n = uo[u].n_sym; const uint16_t* S = sym + units[u].sym_off; uint16_t* F = tails + (size_t)u * WN;
  for (uint32_t j = threadIdx.x; j < WN; j += 256) F[j] = ((uint64_t)n + j < WN) ? (uint16_t)(sqinf::SYM_MARK | (n + j)) : S[(size_t)n + j - WN];
}
// windows: (nunits + 1) x 32 KB; windows[0] is the text in front of the segment's first span (given), windows[u + 1] what lies in front of span u + 1.
// The chain is serial in the spans — some four thousand steps per segment — so a step must cost next to nothing: the window lives in LDS (two buffers of 32 KB: a step reads
// one and writes the other), a thread resolves four neighbouring places at a time (one 8-byte load, one 4-byte store to LDS and to memory), the tails of the next two spans
// are on their way while this one is resolved, and one barrier separates two steps.  (The first version kept the window in global memory with a fence per step and read each
// span's bounds where it needed them: 19 us per span, 76 ms per segment — ten times what decoding the segment takes.)
__global__ void __launch_bounds__(1024) k_gz_chain(uint32_t nunits, const uint16_t* __restrict__ tails, uint8_t* __restrict__ windows) {
  extern __shared__ uint8_t s_win[];      // [2][SPAN_WINDOW]
  constexpr uint32_t WN = sqinf::SPAN_WINDOW, PER = WN / 4 / 1024;      // 8 groups of four places per thread
  for (uint32_t j = threadIdx.x; j < WN / 4; j += 1024) ((uint32_t*)s_win)[j] = ((const uint32_t*)windows)[j];
  uint2 a[PER], b[PER];                                                   // the tails of span u (a) and u + 1 (b)
  auto fetch = [&](uint2* d, uint32_t u) { const uint2* F = (const uint2*)(tails + (size_t)u * WN);
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) d[i] = F[threadIdx.x + 1024u * i]; };
  if (nunits > 0) fetch(a, 0);
  if (nunits > 1) fetch(b, 1);
  __syncthreads();
  for (uint32_t u = 0; u < nunits; ++u) {
    const uint8_t* cur = s_win + (u & 1u) * WN; uint32_t* nxt = (uint32_t*)(s_win + ((u & 1u) ^ 1u) * WN); uint32_t* Wn = (uint32_t*)(windows + (size_t)(u + 1) * WN);
    uint2 c[PER];
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) { c[i] = a[i]; a[i] = b[i]; }
    if (u + 2 < nunits) fetch(b, u + 2);
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) {
      const uint32_t s0 = c[i].x & 0xFFFFu, s1 = c[i].x >> 16, s2 = c[i].y & 0xFFFFu, s3 = c[i].y >> 16;
      const uint32_t v0 = (s0 & sqinf::SYM_MARK) ? cur[s0 & 0x7FFFu] : (s0 & 0xFFu), v1 = (s1 & sqinf::SYM_MARK) ? cur[s1 & 0x7FFFu] : (s1 & 0xFFu);
      const uint32_t v2 = (s2 & sqinf::SYM_MARK) ? cur[s2 & 0x7FFFu] : (s2 & 0xFFu), v3 = (s3 & sqinf::SYM_MARK) ? cur[s3 & 0x7FFFu] : (s3 & 0xFFu);
      const uint32_t v = v0 | (v1 << 8) | (v2 << 16) | (v3 << 24); const uint32_t jj = threadIdx.x + 1024u * i;
      nxt[jj] = v; Wn[jj] = v;
    }
    __syncthreads();
  }
}
// tile t: symbols [tile_first[t], + GZ_TILE) of span tile_unit[t]
__global__ void __launch_bounds__(256) k_gz_translate(const GzUnit* __restrict__ units, const GzUnitOut* __restrict__ uo, const uint64_t* __restrict__ toff, const uint32_t* __restrict__ tile_unit,
                                                       const uint32_t* __restrict__ tile_first, const uint16_t* __restrict__ sym, const uint8_t* __restrict__ windows, uint8_t* __restrict__ text) {
  const uint32_t u = tile_unit[blockIdx.x], first = tile_first[blockIdx.x], n = uo[u].n_sym;
  const uint16_t* S = sym + units[u].sym_off; const uint8_t* W = windows + (size_t)u * sqinf::SPAN_WINDOW; uint8_t* T = text + toff[u];
  const uint32_t end = std::min(n, first + GZ_TILE);
  for (uint32_t i = first + threadIdx.x; i < end; i += 256) { const uint16_t s = S[i]; T[i] = (s & sqinf::SYM_MARK) ? W[s & 0x7FFFu] : (uint8_t)s; }
}
// the CRC-32 of the segment's text in stretches of GZ_CRC_TILE bytes, a wave each: stretches of ONE length fold with one multiplication each on the host
// (crc(A || B) = crc(A) * x^(8 |B|) + crc(B)), where folding spans of their own lengths costs a power of x per span
constexpr uint32_t GZ_CRC_TILE = 65536;
__global__ void __launch_bounds__(64 * GZ_WAVES) k_gz_crc(const uint8_t* __restrict__ text, uint64_t n, uint32_t ntiles, uint32_t* __restrict__ crc) {
  __shared__ uint32_t s_crc[256];
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_crc[i] = sqinf::crc32_entry(i);
  __syncthreads();
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), t = blockIdx.x * GZ_WAVES + wave;
  if (t >= ntiles) return;
  const uint64_t a = (uint64_t)t * GZ_CRC_TILE; const uint32_t len = (uint32_t)std::min<uint64_t>(GZ_CRC_TILE, n - a);
  const uint32_t c = sqinf::crc32_wave(s_crc, text + a, len);
  if ((threadIdx.x & 63) == 0) crc[t] = c;
}

const char* gz_rc_text(uint32_t rc) {
  switch (rc) {
    case sqinf::INF_EOF_INPUT: return "the gzip stream ends in the middle of a block (truncated file?)";
    case sqinf::INF_OUTPUT_SIZE: return "a span holds more text than its buffer";
    case sqinf::INF_OVERRUN: return "a block runs across a presumed block start";
    default: return "corrupt deflate data";
  }
}
template <class T> struct DevBuf { T* p = nullptr; size_t cap = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int need(size_t n) { if (n <= cap) return 0; if (p) (void)hipFree(p); p = nullptr; cap = 0; const size_t c = n + n / 4 + 64; if (hipMalloc((void**)&p, c * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); return -1; } cap = c; return 0; } };
}  // namespace

// A segment's work is two chains: FIND + DECODE (stream A, the host reads their results) and TAILS + CHAIN + TRANSLATE + CRC (stream B, nothing for the host to decide until
// the checksum).  The second chain of segment k runs while the first chain of segment k + 1 does: two SLOTS of buffers alternate, and a slot's checksum is settled — folded into
// the member's, the member held to its trailer — when the slot is needed again, when a member ends, or at the end of the file (the serial window chain, a third of a segment's
// device time on one compute unit, hides behind the next segment's decoding that way).
struct GzSlot {
  DevBuf<uint8_t> d_win; DevBuf<uint64_t> d_toff; DevBuf<GzUnit> d_units; DevBuf<GzUnitOut> d_uout; DevBuf<uint16_t> d_sym, d_tails; DevBuf<uint32_t> d_tile_unit, d_tile_first, d_crc;
  uint8_t* h_tab = nullptr; size_t h_tab_cap = 0;      // page-locked: the small tables up, the stretches' checksums down
  std::vector<GzUnit> units; std::vector<GzUnitOut> uout; std::vector<uint64_t> toff; size_t text_n = 0; uint32_t nct = 0; size_t crc_at = 0;
  bool decoded = false, in_flight = false, ends_member = false; uint32_t trailer_crc = 0, trailer_isize = 0; hipEvent_t ev = nullptr;
  ~GzSlot() { if (h_tab) (void)hipHostFree(h_tab); if (ev) (void)hipEventDestroy(ev); }
  int tab(size_t n) { if (n <= h_tab_cap) return 0; if (h_tab) (void)hipHostFree(h_tab); h_tab = nullptr; h_tab_cap = 0; if (hipHostMalloc((void**)&h_tab, n + n / 4, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return -1; } h_tab_cap = n + n / 4; return 0; }
};
struct sq_gzdev {
  const uint8_t* data = nullptr; size_t bytes = 0; int device = 0; hipStream_t st = nullptr, stb = nullptr; size_t SEG = 64u << 20;
  size_t hdr_at = 0; bool in_member = false, eof = false; uint64_t pos_bit = 0; uint32_t m_crc = 0; uint64_t m_len = 0; uint32_t ratio = 10;
  DevBuf<uint8_t> d_comp, d_carry; DevBuf<uint64_t> d_found;
  uint8_t* h_pin = nullptr; size_t h_pin_cap = 0;      // page-locked staging of a segment's compressed bytes
  GzSlot slot[2]; uint64_t nseg = 0; int cur = -1;     // cur: the slot decoded by the last next(), waiting for its emit()
  sq_gzdev_counters ctr = {0, 0, 0, 0};
  double t_copy = 0, t_find = 0, t_decode = 0, t_settle = 0; uint64_t text_total = 0;      // SQ_READER_STATS=1: where the host waited
  ~sq_gzdev() {
    if (stb) { (void)hipStreamSynchronize(stb); (void)hipStreamDestroy(stb); }
    if (getenv("SQ_READER_STATS")) fprintf(stderr, "[sq_gzdev] %.3f GB of text, %llu segments, %llu spans, %llu members, %llu retries: the host waited for the staging copy %.3f s, the block search %.3f s, the decoding %.3f s, windows + text + checksums %.3f s\n",
        (double)text_total / 1e9, (unsigned long long)ctr.segments, (unsigned long long)ctr.spans, (unsigned long long)ctr.members, (unsigned long long)ctr.retries, t_copy, t_find, t_decode, t_settle);
    if (h_pin) (void)hipHostFree(h_pin); }
  int pin(size_t n) { if (n <= h_pin_cap) return 0; if (h_pin) (void)hipHostFree(h_pin); h_pin = nullptr; h_pin_cap = 0; if (hipHostMalloc((void**)&h_pin, n, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return -1; } h_pin_cap = n; return 0; }
};
static double gz_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// a member header at `at` (RFC 1952): the offset of its deflate stream, 0 if there is none
static size_t gz_member_header(const uint8_t* p, size_t n, size_t at) {
  if (at + 18 > n || p[at] != 0x1f || p[at + 1] != 0x8b || p[at + 2] != 8) return 0;
  const uint8_t flg = p[at + 3]; size_t q = at + 10;
  if (flg & 4) { if (q + 2 > n) return 0; q += 2 + ((size_t)p[q] | ((size_t)p[q + 1] << 8)); }
  if (flg & 8) { while (q < n && p[q]) ++q; ++q; }
  if (flg & 16) { while (q < n && p[q]) ++q; ++q; }
  if (flg & 2) q += 2;
  return q + 8 <= n ? q : 0;
}

int sq_gzdev_open(const uint8_t* data, size_t bytes, int device, hipStream_t st, size_t seg_bytes, sq_gzdev** out, std::string* err) {
  if (!gz_member_header(data, bytes, 0)) { *err = "does not start with a gzip member"; return SQ_ERR_IO; }
  sq_gzdev* g = new sq_gzdev(); g->data = data; g->bytes = bytes; g->device = device; g->st = st; if (seg_bytes) g->SEG = std::max<size_t>(seg_bytes, 4 * GZ_SUB);
  bool ok = hipStreamCreateWithFlags(&g->stb, hipStreamNonBlocking) == hipSuccess;
  for (auto& S : g->slot) ok = ok && hipEventCreateWithFlags(&S.ev, hipEventDisableTiming) == hipSuccess;
  if (!ok || g->d_carry.need(sqinf::SPAN_WINDOW) || hipMemsetAsync(g->d_carry.p, 0, sqinf::SPAN_WINDOW, g->stb) != hipSuccess) { delete g; *err = "device allocation failed (gzip decoder)"; return SQ_ERR_NOMEM; }
  *out = g; return SQ_OK;
}
void sq_gzdev_close(sq_gzdev* g) { delete g; }
sq_gzdev_counters sq_gzdev_stats(const sq_gzdev* g) { return g->ctr; }

// a slot whose second chain is in flight: wait for it, fold its stretches' checksums into the member's, and hold the member to its trailer if it ended there
static int gz_settle(sq_gzdev* g, GzSlot& S, std::string* err) {
  if (!S.in_flight) return SQ_OK;
  const double t0 = gz_now();
  if (hipEventSynchronize(S.ev) != hipSuccess) { *err = std::string("device failure in the gzip decoder (text): ") + hipGetErrorString(hipGetLastError()); return SQ_ERR_DEVICE; }
  g->t_settle += gz_now() - t0; S.in_flight = false;
  if (S.text_n) {      // all stretches but the last have one length: one power of x serves them all
    const uint32_t* crcs = (const uint32_t*)(S.h_tab + S.crc_at); const uint32_t X = sqinf::crc_xpow8(GZ_CRC_TILE); uint32_t c = g->m_crc;
    for (uint32_t t = 0; t < S.nct; ++t) { const uint64_t a = (uint64_t)t * GZ_CRC_TILE; const uint32_t len = (uint32_t)std::min<uint64_t>(GZ_CRC_TILE, S.text_n - a);
      c = sqinf::crc_mul(len == GZ_CRC_TILE ? X : sqinf::crc_xpow8(len), c) ^ crcs[t]; }
    g->m_crc = c; g->m_len += S.text_n;
  }
  if (S.ends_member) {
    if (g->m_crc != S.trailer_crc) { *err = "gzip checksum mismatch (CRC-32 of the inflated text against the member's trailer)"; return SQ_ERR_IO; }
    if ((uint32_t)g->m_len != S.trailer_isize) { *err = "gzip length mismatch (the inflated text against the member's trailer)"; return SQ_ERR_IO; }
    g->m_crc = (uint32_t)crc32(0L, Z_NULL, 0); g->m_len = 0; S.ends_member = false;
  }
  return SQ_OK;
}
static int gz_settle_all(sq_gzdev* g, std::string* err) {      // in segment order: the older slot first
  const int older = (int)(g->nseg & 1);      // the slot the NEXT segment would take holds the older of the two
  for (int k = 0; k < 2; ++k) { const int rc = gz_settle(g, g->slot[(older + k) & 1], err); if (rc) return rc; }
  return SQ_OK;
}
int sq_gzdev_wait(sq_gzdev* g, std::string* err) { return gz_settle_all(g, err); }

int sq_gzdev_next(sq_gzdev* g, size_t* n_out, std::string* err) {
  auto dev_fail = [&](const char* what) { *err = std::string("device failure in the gzip decoder (") + what + "): " + hipGetErrorString(hipGetLastError()); return SQ_ERR_DEVICE; };
  *n_out = 0; g->cur = -1;
  for (;;) {
    if (g->eof) return gz_settle_all(g, err);
    if (!g->in_member) {
      { const int rc = gz_settle_all(g, err); if (rc) return rc; }      // the member before is checked before the next one's sums begin
      // (zero bytes behind the last member are padding, as gzip itself takes them)
      size_t at = g->hdr_at; while (at < g->bytes && g->data[at] == 0) ++at;
      if (at >= g->bytes) { g->eof = true; return SQ_OK; }
      const size_t ds = gz_member_header(g->data, g->bytes, at);
      if (!ds) { *err = "not a gzip member at offset " + std::to_string(at) + " (damaged or truncated file)"; return SQ_ERR_IO; }
      g->pos_bit = (uint64_t)ds * 8ull; g->in_member = true; g->m_crc = (uint32_t)crc32(0L, Z_NULL, 0); g->m_len = 0; ++g->ctr.members;
    }
    GzSlot& S = g->slot[g->nseg & 1];
    { const int rc = gz_settle(g, S, err); if (rc) return rc; }        // (its buffers are free again behind this)
    size_t seg = g->SEG; std::vector<uint64_t> banned;
    for (int attempt = 0;; ++attempt) {
      if (attempt > 64) { *err = "the gzip decoder does not find this file's block boundaries"; return SQ_ERR_IO; }
      const size_t c0 = (size_t)(g->pos_bit >> 3) & ~(size_t)3, c1 = std::min(g->bytes, (size_t)(g->pos_bit >> 3) + seg), nb = c1 - c0;
      const bool to_eof = c1 == g->bytes;
      if (nb >= 0xFFFFFFF0ull) { *err = "gzip segment larger than 4 GB"; return SQ_ERR_STATE; }
      const uint32_t nsub = (uint32_t)((nb + GZ_SUB - 1) / GZ_SUB);
      if (g->pin(nb + 64) || g->d_comp.need(nb + 64) || g->d_found.need(nsub + 8)) { *err = "allocation failed (gzip segment)"; return SQ_ERR_NOMEM; }
      double t0 = gz_now();
      memcpy(g->h_pin, g->data + c0, nb); memset(g->h_pin + nb, 0, 64); g->t_copy += gz_now() - t0; t0 = gz_now();
      if (hipMemcpyAsync(g->d_comp.p, g->h_pin, nb + 64, hipMemcpyHostToDevice, g->st) != hipSuccess) return dev_fail("upload");
      const uint64_t start_rel = g->pos_bit - (uint64_t)c0 * 8ull;
      k_gz_find<<<(nsub + GZ_WAVES - 1) / GZ_WAVES, 64 * GZ_WAVES, 0, g->st>>>(g->d_comp.p, (uint32_t)nb, start_rel + 1, nsub, g->d_found.p);
      std::vector<uint64_t> found(nsub);
      if (hipMemcpyAsync(found.data(), g->d_found.p, (size_t)nsub * 8, hipMemcpyDeviceToHost, g->st) != hipSuccess || hipStreamSynchronize(g->st) != hipSuccess) return dev_fail("block search");
      g->t_find += gz_now() - t0; t0 = gz_now();
      std::vector<uint64_t> starts; starts.push_back(start_rel);
      for (uint64_t f : found) if (f != ~0ull && f > starts.back() && std::find(banned.begin(), banned.end(), f + (uint64_t)c0 * 8ull) == banned.end()) starts.push_back(f);
      if (starts.size() == 1 && !to_eof) { seg *= 2; ++g->ctr.retries; continue; }      // no boundary in the whole segment: a longer one
      S.units.clear(); uint64_t so = 0;
      const size_t nu = to_eof ? starts.size() : starts.size() - 1;
      for (size_t i = 0; i < nu; ++i) {
        GzUnit U; U.start_bit = starts[i]; U.stop_bit = i + 1 < starts.size() ? starts[i + 1] : ~0ull;
        const uint64_t cb = (U.stop_bit == ~0ull ? (uint64_t)nb * 8ull : U.stop_bit) - U.start_bit;
        U.cap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, (cb / 8 + 1) * g->ratio + 4096); U.sym_off = so; U._pad = 0; so += (U.cap + 63) & ~63ull; S.units.push_back(U);
      }
      const uint32_t K = (uint32_t)S.units.size();
      if (S.tab(((size_t)K + 1) * (sizeof(GzUnit) + sizeof(GzUnitOut)) + 64) || S.d_units.need(K + 1) || S.d_uout.need(K + 1) || S.d_sym.need((size_t)so + 64)) { *err = "allocation failed (gzip symbols, " + std::to_string(so >> 19) + " MB)"; return SQ_ERR_NOMEM; }
      memcpy(S.h_tab, S.units.data(), (size_t)K * sizeof(GzUnit)); GzUnitOut* h_uo = (GzUnitOut*)(S.h_tab + ((size_t)K + 1) * sizeof(GzUnit));
      if (hipMemcpyAsync(S.d_units.p, S.h_tab, (size_t)K * sizeof(GzUnit), hipMemcpyHostToDevice, g->st) != hipSuccess) return dev_fail("span table");
      k_gz_decode<<<(K + GZ_WAVES - 1) / GZ_WAVES, 64 * GZ_WAVES, 0, g->st>>>(g->d_comp.p, (uint32_t)nb, S.d_units.p, K, S.d_sym.p, S.d_uout.p);
      if (hipMemcpyAsync(h_uo, S.d_uout.p, (size_t)K * sizeof(GzUnitOut), hipMemcpyDeviceToHost, g->st) != hipSuccess || hipStreamSynchronize(g->st) != hipSuccess) return dev_fail("decode");
      S.uout.assign(h_uo, h_uo + K);
      g->t_decode += gz_now() - t0;
      // what the spans say, in order
      uint32_t keep = 0; bool again = false; S.ends_member = false;
      for (uint32_t i = 0; i < K && !again; ++i) {
        const GzUnitOut& o = S.uout[i];
        if (o.rc == sqinf::INF_OUTPUT_SIZE) { if (g->ratio >= 1100) { *err = "a span of the gzip stream expands more than deflate can"; return SQ_ERR_IO; } g->ratio *= 2; again = true; break; }
        if (o.rc == sqinf::INF_OVERRUN && i + 1 < starts.size()) { banned.push_back(starts[i + 1] + (uint64_t)c0 * 8ull); again = true; break; }   // that start was no block boundary
        if (o.rc != sqinf::INF_OK) { *err = std::string(gz_rc_text(o.rc)) + " (near compressed offset " + std::to_string(c0 + (size_t)(S.units[i].start_bit >> 3)) + ")"; return SQ_ERR_IO; }
        keep = i + 1;
        if (o.final_) {      // the member ends here: its trailer, then (next call) the next member's header
          const size_t tr = c0 + (size_t)((o.end_bit + 7) >> 3);
          if (tr + 8 > g->bytes) { *err = "the gzip file ends inside a member's trailer (truncated)"; return SQ_ERR_IO; }
          S.trailer_crc = (uint32_t)g->data[tr] | ((uint32_t)g->data[tr + 1] << 8) | ((uint32_t)g->data[tr + 2] << 16) | ((uint32_t)g->data[tr + 3] << 24);
          S.trailer_isize = (uint32_t)g->data[tr + 4] | ((uint32_t)g->data[tr + 5] << 8) | ((uint32_t)g->data[tr + 6] << 16) | ((uint32_t)g->data[tr + 7] << 24);
          S.ends_member = true; g->hdr_at = tr + 8; break;
        }
      }
      if (again) { ++g->ctr.retries; continue; }
      if (!S.ends_member) {
        if (to_eof) { *err = "the gzip stream ends without a final block (truncated file?)"; return SQ_ERR_IO; }
        g->pos_bit = starts.back() + (uint64_t)c0 * 8ull;      // the last found start begins the next segment
      } else g->in_member = false;
      S.units.resize(keep); S.uout.resize(keep);
      break;
    }
    const uint32_t K = (uint32_t)S.units.size();
    S.toff.assign((size_t)K + 1, 0); for (uint32_t i = 0; i < K; ++i) S.toff[i + 1] = S.toff[i] + S.uout[i].n_sym;
    S.text_n = (size_t)S.toff[K]; ++g->ctr.segments; g->ctr.spans += K; S.decoded = true; g->cur = (int)(g->nseg & 1); ++g->nseg;
    if (S.text_n == 0) {   // (an empty member, or spans without text): settled at once
      std::string e2; const int rc = sq_gzdev_emit(g, nullptr, nullptr, &e2); if (rc) { *err = e2; return rc; }
      continue;
    }
    *n_out = S.text_n; return SQ_OK;
  }
}

// the second chain of the segment next() has just decoded, queued on the decoder's own stream: windows, text into d_dst, the stretches' checksums.  Returns at once;
// `done` (may be null) is recorded behind the text.  The checksums are looked at by a later next() (or sq_gzdev_wait): a damaged file is refused a segment late, never accepted
int sq_gzdev_emit(sq_gzdev* g, uint8_t* d_dst, hipEvent_t done, std::string* err) {
  auto dev_fail = [&](const char* what) { *err = std::string("device failure in the gzip decoder (") + what + "): " + hipGetErrorString(hipGetLastError()); return SQ_ERR_DEVICE; };
  if (g->cur < 0 || !g->slot[g->cur].decoded) { *err = "internal: sq_gzdev_emit without a decoded segment"; return SQ_ERR_STATE; }
  GzSlot& S = g->slot[g->cur]; S.decoded = false; g->cur = -1; g->text_total += S.text_n;
  const uint32_t K = (uint32_t)S.units.size(); S.nct = (uint32_t)((S.text_n + GZ_CRC_TILE - 1) / GZ_CRC_TILE);
  if (K && S.text_n) {
    size_t ntile = 0; for (uint32_t u = 0; u < K; ++u) ntile += (S.uout[u].n_sym + GZ_TILE - 1) / GZ_TILE;
    // the page-locked table of the slot: [units K + 1][span results K + 1] (next()), then [toff (K + 1) x 8][tile_unit][tile_first][checksums nct]
    const size_t base = ((size_t)K + 1) * (sizeof(GzUnit) + sizeof(GzUnitOut)), b_toff = ((size_t)K + 1) * 8, b_tile = ntile * 4; S.crc_at = base + b_toff + 2 * b_tile;
    if (S.tab(S.crc_at + (size_t)S.nct * 4 + 64)) { *err = "allocation failed (gzip text)"; return SQ_ERR_NOMEM; }
    if (S.d_toff.need((size_t)K + 1) || S.d_tile_unit.need(ntile + 1) || S.d_tile_first.need(ntile + 1) || S.d_crc.need((size_t)S.nct + 1) || S.d_win.need(((size_t)K + 1) * sqinf::SPAN_WINDOW) ||
        S.d_tails.need((size_t)K * sqinf::SPAN_WINDOW + 64)) { *err = "device allocation failed (gzip text)"; return SQ_ERR_NOMEM; }
    uint64_t* h_toff = (uint64_t*)(S.h_tab + base); uint32_t* h_tu = (uint32_t*)(S.h_tab + base + b_toff); uint32_t* h_tf = h_tu + ntile; size_t q = 0;
    memcpy(h_toff, S.toff.data(), b_toff);
    for (uint32_t u = 0; u < K; ++u) for (uint32_t f = 0; f < S.uout[u].n_sym; f += GZ_TILE) { h_tu[q] = u; h_tf[q] = f; ++q; }
    hipStream_t B = g->stb;
    if (hipMemcpyAsync(S.d_win.p, g->d_carry.p, sqinf::SPAN_WINDOW, hipMemcpyDeviceToDevice, B) != hipSuccess) return dev_fail("window");
    k_gz_tails<<<K, 256, 0, B>>>(S.d_units.p, S.d_uout.p, S.d_sym.p, S.d_tails.p);
    k_gz_chain<<<1, 1024, 2 * sqinf::SPAN_WINDOW, B>>>(K, S.d_tails.p, S.d_win.p);
    if (hipMemcpyAsync(g->d_carry.p, S.d_win.p + (size_t)K * sqinf::SPAN_WINDOW, sqinf::SPAN_WINDOW, hipMemcpyDeviceToDevice, B) != hipSuccess) return dev_fail("window");
    if (hipMemcpyAsync(S.d_toff.p, h_toff, b_toff, hipMemcpyHostToDevice, B) != hipSuccess || (ntile && (hipMemcpyAsync(S.d_tile_unit.p, h_tu, b_tile, hipMemcpyHostToDevice, B) != hipSuccess ||
        hipMemcpyAsync(S.d_tile_first.p, h_tf, b_tile, hipMemcpyHostToDevice, B) != hipSuccess))) return dev_fail("tables");
    if (ntile) k_gz_translate<<<(uint32_t)ntile, 256, 0, B>>>(S.d_units.p, S.d_uout.p, S.d_toff.p, S.d_tile_unit.p, S.d_tile_first.p, S.d_sym.p, S.d_win.p, d_dst);
    if (done && hipEventRecord(done, B) != hipSuccess) return dev_fail("text");
    k_gz_crc<<<(S.nct + GZ_WAVES - 1) / GZ_WAVES, 64 * GZ_WAVES, 0, B>>>(d_dst, (uint64_t)S.text_n, S.nct, S.d_crc.p);
    if (hipMemcpyAsync(S.h_tab + S.crc_at, S.d_crc.p, (size_t)S.nct * 4, hipMemcpyDeviceToHost, B) != hipSuccess) return dev_fail("text");
  } else if (done && hipEventRecord(done, g->stb) != hipSuccess) return dev_fail("text");
  if (hipEventRecord(S.ev, g->stb) != hipSuccess) return dev_fail("text");
  S.in_flight = true;
  if (!S.text_n) return gz_settle_all(g, err);      // (in segment order: the member's earlier segments first)
  return SQ_OK;
}

// ---- test hook: a whole gzip file (host memory) through the device decoder, the text back to the host ------------------------------------------------------------
extern "C" int sq_debug_gzip_inflate(int device, const uint8_t* gz, uint64_t gz_bytes, uint64_t seg_bytes, uint8_t* text, uint64_t text_cap, uint64_t* text_n, uint64_t* counters4) {
  if (!gz || !text || !text_n) { sq_set_error("sq_debug_gzip_inflate: bad arguments"); return SQ_ERR_ARG; }
  if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); sq_set_error("no HIP device %d", device); return SQ_ERR_DEVICE; }
  hipStream_t st = nullptr; if (hipStreamCreate(&st) != hipSuccess) { sq_set_error("stream creation failed"); return SQ_ERR_DEVICE; }
  sq_gzdev* g = nullptr; std::string e; int rc = sq_gzdev_open(gz, (size_t)gz_bytes, device, st, (size_t)seg_bytes, &g, &e);
  uint64_t total = 0; void* d = nullptr; size_t dcap = 0;
  while (!rc) {
    size_t n = 0; rc = sq_gzdev_next(g, &n, &e); if (rc || !n) break;
    if (n > dcap) { if (d) (void)hipFree(d); d = nullptr; dcap = 0; if (hipMalloc(&d, n + n / 4 + 64) != hipSuccess) { rc = SQ_ERR_NOMEM; e = "device allocation failed"; break; } dcap = n + n / 4; }
    rc = sq_gzdev_emit(g, (uint8_t*)d, nullptr, &e); if (rc) break;
    rc = sq_gzdev_wait(g, &e); if (rc) break;      // (one buffer here: its text goes to the host before the next segment's is written)
    if (total + n > text_cap) { rc = SQ_ERR_OVERFLOW; e = "text buffer too small"; break; }
    if (hipMemcpy(text + total, d, n, hipMemcpyDeviceToHost) != hipSuccess) { rc = SQ_ERR_DEVICE; e = "copy failed"; break; }
    total += n;
  }
  if (g && counters4) { const sq_gzdev_counters c = sq_gzdev_stats(g); counters4[0] = c.segments; counters4[1] = c.spans; counters4[2] = c.members; counters4[3] = c.retries; }
  if (g) sq_gzdev_close(g);
  if (d) (void)hipFree(d);
  (void)hipStreamDestroy(st);
  *text_n = total;
  if (rc) sq_set_error("%s", e.c_str());
  return rc;
}
// the decoder's source on the host (inflate_core.h compiled for the CPU): a span between two bit positions into symbols, and the search for a block start — how
// tests check both against zlib where there is no GPU.  Not a path of the product.
extern "C" int sq_debug_inflate_span_host(const uint8_t* comp, uint64_t n, uint64_t start_bit, uint64_t stop_bit, uint16_t* sym, uint32_t cap, uint32_t* n_sym, uint64_t* end_bit, uint32_t* ended_final) {
  static sqinf::Tables T;
  return sqinf::inflate_span<uint16_t>(comp, (size_t)n, start_bit, stop_bit, sym, cap, sqinf::SPAN_WINDOW, T, n_sym, end_bit, ended_final, 0u);
}
extern "C" uint64_t sq_debug_find_block_start_host(const uint8_t* comp, uint64_t n, uint64_t lo, uint64_t hi) { static sqinf::Tables T; return sqinf::find_block_start(comp, (size_t)n, lo, hi, T); }
