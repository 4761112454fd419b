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
//   k_gz_chain      ONE block walks the spans in order: the window in front of span u + 1 is the last 32 KB of (window of span u, span u resolved with it);
//   k_gz_translate  symbols -> bytes, every span with its window, into the caller's text buffer;   k_gz_crc: a wave per span takes the CRC-32 of its text, the host folds
//                   them (crc32_combine) and holds every member to its trailer (CRC-32 and length).
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
__global__ void __launch_bounds__(64 * GZ_WAVES) k_gz_crc(const GzUnitOut* __restrict__ uo, const uint64_t* __restrict__ toff, uint32_t nunits, const uint8_t* __restrict__ text, uint32_t* __restrict__ crc) {
  __shared__ uint32_t s_crc[256];
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_crc[i] = sqinf::crc32_entry(i);
  __syncthreads();
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), u = blockIdx.x * GZ_WAVES + wave;
  if (u >= nunits) return;
  const uint32_t c = sqinf::crc32_wave(s_crc, text + toff[u], uo[u].n_sym);
  if ((threadIdx.x & 63) == 0) crc[u] = c;
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

struct sq_gzdev {
  const uint8_t* data = nullptr; size_t bytes = 0; int device = 0; hipStream_t st = nullptr; size_t SEG = 64u << 20;
  size_t hdr_at = 0; bool in_member = false, eof = false; uint64_t pos_bit = 0; uint32_t m_crc = 0; uint64_t m_len = 0; uint32_t ratio = 10;
  DevBuf<uint8_t> d_comp, d_win; DevBuf<uint64_t> d_found, d_toff; DevBuf<GzUnit> d_units; DevBuf<GzUnitOut> d_uout; DevBuf<uint16_t> d_sym, d_tails; DevBuf<uint32_t> d_tile_unit, d_tile_first, d_crc;
  uint8_t* h_pin = nullptr; size_t h_pin_cap = 0;      // page-locked staging: the segment's compressed bytes up, the small tables down
  uint8_t carry[sqinf::SPAN_WINDOW];                   // (host copy not needed: the carried window stays on the device, d_carry)
  DevBuf<uint8_t> d_carry;
  // the segment between next() and emit()
  std::vector<GzUnit> units; std::vector<GzUnitOut> uout; std::vector<uint64_t> toff; size_t text_n = 0; bool pending = false;
  bool seg_ends_member = false; uint32_t trailer_crc = 0, trailer_isize = 0;
  sq_gzdev_counters ctr = {0, 0, 0, 0};
  double t_copy = 0, t_find = 0, t_decode = 0, t_chain_emit = 0; uint64_t text_total = 0;      // SQ_READER_STATS=1: where the host waited
  ~sq_gzdev() {
    if (getenv("SQ_READER_STATS")) fprintf(stderr, "[sq_gzdev] %.3f GB of text, %llu segments, %llu spans, %llu members, %llu retries: staging copy %.3f s, block search %.3f s, decode %.3f s, windows + text + crc %.3f s\n",
        (double)text_total / 1e9, (unsigned long long)ctr.segments, (unsigned long long)ctr.spans, (unsigned long long)ctr.members, (unsigned long long)ctr.retries, t_copy, t_find, t_decode, t_chain_emit);
    if (h_pin) (void)hipHostFree(h_pin); }
  int pin(size_t n) { if (n <= h_pin_cap) return 0; if (h_pin) (void)hipHostFree(h_pin); h_pin = nullptr; h_pin_cap = 0; if (hipHostMalloc((void**)&h_pin, n, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return -1; } h_pin_cap = n; return 0; }
};

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
  if (g->d_carry.need(sqinf::SPAN_WINDOW) || hipMemsetAsync(g->d_carry.p, 0, sqinf::SPAN_WINDOW, st) != hipSuccess) { delete g; *err = "device allocation failed (gzip window)"; return SQ_ERR_NOMEM; }
  *out = g; return SQ_OK;
}
void sq_gzdev_close(sq_gzdev* g) { delete g; }
sq_gzdev_counters sq_gzdev_stats(const sq_gzdev* g) { return g->ctr; }

int sq_gzdev_next(sq_gzdev* g, size_t* n_out, std::string* err) {
  auto dev_fail = [&](const char* what) { *err = std::string("device failure in the gzip decoder (") + what + "): " + hipGetErrorString(hipGetLastError()); return SQ_ERR_DEVICE; };
  *n_out = 0; g->pending = false;
  for (;;) {
    if (g->eof) return SQ_OK;
    if (!g->in_member) {
      // (zero bytes behind the last member are padding, as gzip itself takes them)
      size_t at = g->hdr_at; while (at < g->bytes && g->data[at] == 0) ++at;
      if (at >= g->bytes) { g->eof = true; return SQ_OK; }
      const size_t ds = gz_member_header(g->data, g->bytes, at);
      if (!ds) { *err = "not a gzip member at offset " + std::to_string(at) + " (damaged or truncated file)"; return SQ_ERR_IO; }
      g->pos_bit = (uint64_t)ds * 8ull; g->in_member = true; g->m_crc = (uint32_t)crc32(0L, Z_NULL, 0); g->m_len = 0; ++g->ctr.members;
    }
    size_t seg = g->SEG; std::vector<uint64_t> banned;
    for (int attempt = 0;; ++attempt) {
      if (attempt > 64) { *err = "the gzip decoder does not find this file's block boundaries"; return SQ_ERR_IO; }
      const size_t c0 = (size_t)(g->pos_bit >> 3) & ~(size_t)3, c1 = std::min(g->bytes, (size_t)(g->pos_bit >> 3) + seg), nb = c1 - c0;
      const bool to_eof = c1 == g->bytes;
      if (nb >= 0xFFFFFFF0ull) { *err = "gzip segment larger than 4 GB"; return SQ_ERR_STATE; }
      const uint32_t nsub = (uint32_t)((nb + GZ_SUB - 1) / GZ_SUB);
      if (g->pin(std::max<size_t>(nb + 64, (size_t)nsub * 64 + 4096)) || g->d_comp.need(nb + 64) || g->d_found.need(nsub + 8)) { *err = "allocation failed (gzip segment)"; return SQ_ERR_NOMEM; }
      auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }; double t0 = tnow();
      memcpy(g->h_pin, g->data + c0, nb); memset(g->h_pin + nb, 0, 64); g->t_copy += tnow() - t0; t0 = tnow();
      if (hipMemcpyAsync(g->d_comp.p, g->h_pin, nb + 64, hipMemcpyHostToDevice, g->st) != hipSuccess) return dev_fail("upload");
      const uint64_t start_rel = g->pos_bit - (uint64_t)c0 * 8ull;
      k_gz_find<<<(nsub + GZ_WAVES - 1) / GZ_WAVES, 64 * GZ_WAVES, 0, g->st>>>(g->d_comp.p, (uint32_t)nb, start_rel + 1, nsub, g->d_found.p);
      if (hipStreamSynchronize(g->st) != hipSuccess) return dev_fail("block search");      // (the staging buffer is free again behind this)
      g->t_find += tnow() - t0; t0 = tnow();
      std::vector<uint64_t> found(nsub);
      if (hipMemcpy(found.data(), g->d_found.p, (size_t)nsub * 8, hipMemcpyDeviceToHost) != hipSuccess) return dev_fail("block search");
      std::vector<uint64_t> starts; starts.push_back(start_rel);
      for (uint64_t f : found) if (f != ~0ull && f > starts.back() && std::find(banned.begin(), banned.end(), f + (uint64_t)c0 * 8ull) == banned.end()) starts.push_back(f);
      if (starts.size() == 1 && !to_eof) { seg *= 2; ++g->ctr.retries; continue; }      // no boundary in the whole segment: a longer one
      g->units.clear(); uint64_t so = 0;
      const size_t nu = to_eof ? starts.size() : starts.size() - 1;
      for (size_t i = 0; i < nu; ++i) {
        GzUnit U; U.start_bit = starts[i]; U.stop_bit = i + 1 < starts.size() ? starts[i + 1] : ~0ull;
        const uint64_t cb = (U.stop_bit == ~0ull ? (uint64_t)nb * 8ull : U.stop_bit) - U.start_bit;
        U.cap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, (cb / 8 + 1) * g->ratio + 4096); U.sym_off = so; U._pad = 0; so += (U.cap + 63) & ~63ull; g->units.push_back(U);
      }
      const uint32_t K = (uint32_t)g->units.size();
      if (g->d_units.need(K + 1) || g->d_uout.need(K + 1) || g->d_sym.need((size_t)so + 64)) { *err = "device allocation failed (gzip symbols, " + std::to_string(so >> 19) + " MB)"; return SQ_ERR_NOMEM; }
      if (hipMemcpyAsync(g->d_units.p, g->units.data(), (size_t)K * sizeof(GzUnit), hipMemcpyHostToDevice, g->st) != hipSuccess) return dev_fail("span table");
      k_gz_decode<<<(K + GZ_WAVES - 1) / GZ_WAVES, 64 * GZ_WAVES, 0, g->st>>>(g->d_comp.p, (uint32_t)nb, g->d_units.p, K, g->d_sym.p, g->d_uout.p);
      g->uout.resize(K);
      if (hipMemcpyAsync(g->uout.data(), g->d_uout.p, (size_t)K * sizeof(GzUnitOut), hipMemcpyDeviceToHost, g->st) != hipSuccess || hipStreamSynchronize(g->st) != hipSuccess) return dev_fail("decode");
      g->t_decode += tnow() - t0;
      // what the spans say, in order
      uint32_t keep = 0; bool again = false; g->seg_ends_member = false;
      for (uint32_t i = 0; i < K && !again; ++i) {
        const GzUnitOut& o = g->uout[i];
        if (o.rc == sqinf::INF_OUTPUT_SIZE) { if (g->ratio >= 1100) { *err = "a span of the gzip stream expands more than deflate can"; return SQ_ERR_IO; } g->ratio *= 2; again = true; break; }
        if (o.rc == sqinf::INF_OVERRUN && i + 1 < starts.size()) { banned.push_back(starts[i + 1] + (uint64_t)c0 * 8ull); again = true; break; }   // that start was no block boundary
        if (o.rc != sqinf::INF_OK) { *err = std::string(gz_rc_text(o.rc)) + " (near compressed offset " + std::to_string(c0 + (size_t)(g->units[i].start_bit >> 3)) + ")"; return SQ_ERR_IO; }
        keep = i + 1;
        if (o.final_) {      // the member ends here: its trailer, then (next call) the next member's header
          const size_t tr = c0 + (size_t)((o.end_bit + 7) >> 3);
          if (tr + 8 > g->bytes) { *err = "the gzip file ends inside a member's trailer (truncated)"; return SQ_ERR_IO; }
          g->trailer_crc = (uint32_t)g->data[tr] | ((uint32_t)g->data[tr + 1] << 8) | ((uint32_t)g->data[tr + 2] << 16) | ((uint32_t)g->data[tr + 3] << 24);
          g->trailer_isize = (uint32_t)g->data[tr + 4] | ((uint32_t)g->data[tr + 5] << 8) | ((uint32_t)g->data[tr + 6] << 16) | ((uint32_t)g->data[tr + 7] << 24);
          g->seg_ends_member = true; g->hdr_at = tr + 8; break;
        }
      }
      if (again) { ++g->ctr.retries; continue; }
      if (!g->seg_ends_member) {
        if (to_eof) { *err = "the gzip stream ends without a final block (truncated file?)"; return SQ_ERR_IO; }
        g->pos_bit = starts.back() + (uint64_t)c0 * 8ull;      // the last found start begins the next segment
      } else g->in_member = false;
      g->units.resize(keep); g->uout.resize(keep);
      break;
    }
    const uint32_t K = (uint32_t)g->units.size();
    g->toff.assign((size_t)K + 1, 0); for (uint32_t i = 0; i < K; ++i) g->toff[i + 1] = g->toff[i] + g->uout[i].n_sym;
    g->text_n = (size_t)g->toff[K]; ++g->ctr.segments; g->ctr.spans += K;
    if (K) {   // the windows: the carried one in front, then span after span
      if (g->d_win.need(((size_t)K + 1) * sqinf::SPAN_WINDOW)) { *err = "device allocation failed (gzip windows)"; return SQ_ERR_NOMEM; }
      if (hipMemcpyAsync(g->d_win.p, g->d_carry.p, sqinf::SPAN_WINDOW, hipMemcpyDeviceToDevice, g->st) != hipSuccess) return dev_fail("window");
      if (g->d_tails.need((size_t)K * sqinf::SPAN_WINDOW + 64)) { *err = "device allocation failed (gzip windows)"; return SQ_ERR_NOMEM; }
      k_gz_tails<<<K, 256, 0, g->st>>>(g->d_units.p, g->d_uout.p, g->d_sym.p, g->d_tails.p);
      k_gz_chain<<<1, 1024, 2 * sqinf::SPAN_WINDOW, g->st>>>(K, g->d_tails.p, g->d_win.p);
      if (hipMemcpyAsync(g->d_carry.p, g->d_win.p + (size_t)K * sqinf::SPAN_WINDOW, sqinf::SPAN_WINDOW, hipMemcpyDeviceToDevice, g->st) != hipSuccess) return dev_fail("window");
    }
    g->pending = true;
    if (g->text_n == 0) {   // (an empty member, or a span without text): settle it and go on
      std::string e2; const int rc = sq_gzdev_emit(g, nullptr, &e2); if (rc) { *err = e2; return rc; }
      continue;
    }
    *n_out = g->text_n; return SQ_OK;
  }
}

int sq_gzdev_emit(sq_gzdev* g, uint8_t* d_dst, std::string* err) {
  auto dev_fail = [&](const char* what) { *err = std::string("device failure in the gzip decoder (") + what + "): " + hipGetErrorString(hipGetLastError()); return SQ_ERR_DEVICE; };
  if (!g->pending) { *err = "internal: sq_gzdev_emit without a decoded segment"; return SQ_ERR_STATE; }
  g->pending = false; const auto te0 = std::chrono::steady_clock::now(); g->text_total += g->text_n;
  struct Tm { sq_gzdev* g; std::chrono::steady_clock::time_point t0; ~Tm() { g->t_chain_emit += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } tm{g, te0};
  const uint32_t K = (uint32_t)g->units.size(); std::vector<uint32_t> crcs(K, 0);
  if (K && g->text_n) {
    std::vector<uint32_t> tu, tf;
    for (uint32_t u = 0; u < K; ++u) for (uint32_t f = 0; f < g->uout[u].n_sym; f += GZ_TILE) { tu.push_back(u); tf.push_back(f); }
    if (g->d_toff.need((size_t)K + 1) || g->d_tile_unit.need(tu.size() + 1) || g->d_tile_first.need(tf.size() + 1) || g->d_crc.need((size_t)K + 1)) { *err = "device allocation failed (gzip text)"; return SQ_ERR_NOMEM; }
    if (hipMemcpyAsync(g->d_toff.p, g->toff.data(), ((size_t)K + 1) * 8, hipMemcpyHostToDevice, g->st) != hipSuccess || hipMemcpyAsync(g->d_tile_unit.p, tu.data(), tu.size() * 4, hipMemcpyHostToDevice, g->st) != hipSuccess ||
        hipMemcpyAsync(g->d_tile_first.p, tf.data(), tf.size() * 4, hipMemcpyHostToDevice, g->st) != hipSuccess || hipStreamSynchronize(g->st) != hipSuccess) return dev_fail("tables");   // (pageable sources: done with them before they go out of scope)
    if (!tu.empty()) k_gz_translate<<<(uint32_t)tu.size(), 256, 0, g->st>>>(g->d_units.p, g->d_uout.p, g->d_toff.p, g->d_tile_unit.p, g->d_tile_first.p, g->d_sym.p, g->d_win.p, d_dst);
    k_gz_crc<<<(K + GZ_WAVES - 1) / GZ_WAVES, 64 * GZ_WAVES, 0, g->st>>>(g->d_uout.p, g->d_toff.p, K, d_dst, g->d_crc.p);
    if (hipMemcpyAsync(crcs.data(), g->d_crc.p, (size_t)K * 4, hipMemcpyDeviceToHost, g->st) != hipSuccess || hipStreamSynchronize(g->st) != hipSuccess) return dev_fail("text");
  }
  for (uint32_t u = 0; u < K; ++u) if (g->uout[u].n_sym) { g->m_crc = (uint32_t)crc32_combine(g->m_crc, crcs[u], (z_off_t)g->uout[u].n_sym); g->m_len += g->uout[u].n_sym; }
  if (g->seg_ends_member) {
    if (g->m_crc != g->trailer_crc) { *err = "gzip checksum mismatch (CRC-32 of the inflated text against the member's trailer)"; return SQ_ERR_IO; }
    if ((uint32_t)g->m_len != g->trailer_isize) { *err = "gzip length mismatch (the inflated text against the member's trailer)"; return SQ_ERR_IO; }
    g->seg_ends_member = false;
  }
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
    rc = sq_gzdev_emit(g, (uint8_t*)d, &e); if (rc) break;
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
