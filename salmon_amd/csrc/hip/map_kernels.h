// hip/map_kernels.h — device kernels of the mapping pipeline (seam B1), one stage per kernel:
//   k_pack      ASCII reads -> 2-bit words + N mask                        (K1 of SURVEY.md §7.4)
//   k_seed      per read end: SSHash lookup + uni-MEM extension            (K2+K3; row a1)
//   k_project_list  large ends only: uni-MEM x contig-table occurrences -> MEM sort records (row a2; everything else: mem_kernels.h)
//   [radix sort by (read end, global reference position)]
//   k_join      per fragment: pair / orphan candidates                     (K6;    row a3)
//   k_score     per candidate: selective-alignment score, fast path        (K7;    row a4)
//   k_dp        queued banded affine-gap DP regions                        (K7)
//   k_select    per fragment: best-per-transcript, decoys, estAlnProb      (K8;    rows a7-a9)
// Everything is integer / fixed-order arithmetic so results equal the CPU checker bit for bit.
#pragma once
#include "ctx.h"

namespace sqk {

typedef uint32_t sq_u32x4 __attribute__((ext_vector_type(4)));
typedef uint64_t sq_u64x2 __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(8))) sq_u32x4_a8 { sq_u32x4 v; };   // 16 bytes at an 8-byte aligned address: still one global_load_dwordx4
struct __attribute__((aligned(8))) sq_u64x2_a8 { sq_u64x2 v; };

struct ReadView { const uint64_t* w; const uint64_t* nm; int L; };

// Same-address atomics serialise at ~4-12 ns each on gfx950 (one per THREAD turned a 1M-thread kernel
// into 12 ms of atomic traffic).  Counters are therefore summed across the wave first; ALL 64 lanes
// must call this (kernels keep out-of-range lanes alive with a zero contribution).
__device__ inline void wave_stat_add(unsigned long long* p, unsigned long long v) {
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_down(v, s, 64);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(p, v);
}

__device__ inline uint64_t fetch_bits(const uint64_t* m, uint32_t p, uint32_t n) {  // n <= 32 one-bit flags from p
  uint32_t w = p >> 6, sh = p & 63;
  uint64_t lo = m[w] >> sh;
  if (sh + n > 64) lo |= m[w + 1] << (64 - sh);
  return lo & ((n >= 64) ? ~0ULL : ((1ULL << n) - 1));
}
__device__ inline uint32_t rd_base(const ReadView& r, int i) {
  if ((r.nm[i >> 6] >> (i & 63)) & 1) return 4;
  return (uint32_t)(r.w[i >> 5] >> ((i & 31) * 2)) & 3u;
}
__device__ inline uint32_t norm_base(const ReadView& r, bool fw, int x) {  // strand-normalised read
  if (fw) return rd_base(r, x);
  uint32_t b = rd_base(r, r.L - 1 - x);
  return b > 3 ? 4u : 3u - b;
}
__device__ inline ReadView read_view(const uint64_t* rpack, const uint64_t* rnmask, const uint16_t* rlen, uint32_t e, uint32_t rw) {   // rw = the context's packing stride (words)
  ReadView r; r.w = rpack + (size_t)e * rw; r.nm = rnmask + (size_t)e * (rw >> 1); r.L = rlen[e]; return r;
}

// [r2] One same-address atomic per WAVE is still too many when a launch has 10^5 waves: L2 retires them one per 4-12 ns, and 95 000
// waves x ~5 atomics were the whole 3.1 ms of k_score while every counter said "waiting for memory".  Per-launch counters and queue
// cursors are therefore summed in LDS and leave with one atomic per BLOCK.  All 64 lanes call this; `s_slot` is in LDS.
__device__ inline void block_stat_add(unsigned long long* s_slot, unsigned long long v) {
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_down(v, s, 64);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(s_slot, v);
}

// unique slot for every calling lane with ONE atomic per wave (works in divergent code: the ballot
// only sees the active lanes)
__device__ inline uint32_t wave_alloc(uint32_t* ctr) {
  const unsigned long long m = __ballot(1);
  const int leader = __ffsll((long long)m) - 1, lane = (int)(threadIdx.x & 63);
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(ctr, (uint32_t)__popcll(m));
  base = (uint32_t)__shfl((int)base, leader, 64);
  return base + (uint32_t)__popcll(m & ((1ULL << lane) - 1));
}

// ------------------------------------------------------------------------------------------------
// [r3] one thread per read end: its (up to) eight 32-base words are loaded with 16-byte loads that are all in flight together, packed
// in registers and stored as 16-byte pairs.  (Round 2 used 8 threads per end, one word each: 128 M threads per 8 M pairs whose two
// dependent round trips — offsets, then bases — set the pace at 2.2 ms; here a wave covers 64 ends and there are 8x fewer waves.)
template <uint32_t RW>   // packing stride in words: 8 (reads of up to 256 bases, the default), 16, 32
__device__ __forceinline__ void pack_end(const uint8_t* __restrict__ seq, const uint64_t* __restrict__ seq_off, uint32_t nrec, uint32_t e,
                       uint64_t* __restrict__ rpack, uint64_t* __restrict__ rnmask, uint16_t* __restrict__ rlen, uint8_t* __restrict__ rany, unsigned long long* __restrict__ stats) {
  const uint64_t a = seq_off[e], b = seq_off[e + 1];
  uint32_t L = (uint32_t)(b - a);
  // [r4] a read that does not fit the stride is reported, not cut: the host packs the batch again with a wider stride, or refuses it (> SQ_MAX_READ_LEN)
  if (L > 32u * RW || L > SQ_MAX_READ_LEN) { atomicMax(&stats[ST_MAXLEN], (unsigned long long)(b - a)); atomicAdd(&stats[ST_TRUNC], 1ULL); L = 32u * RW < SQ_MAX_READ_LEN ? 32u * RW : SQ_MAX_READ_LEN; }
  const uint8_t* s = seq + a;
  struct __attribute__((packed, aligned(4))) Q4 { uint32_t v[4]; };
  // reads start at any byte (2x150: every other record is 2 mod 4): dwords are loaded from the aligned address below and funnel-shifted;
  // the dword past a full word is only touched when it lies inside the batch's buffer (not for the last record)
  const uint32_t mis = (uint32_t)(((uintptr_t)s) & 3); const uint8_t* s0 = s - mis; const bool wide = mis == 0 || e + 1 < nrec;
  uint64_t cw[RW]; uint32_t cn[RW];
#pragma unroll
  for (uint32_t w = 0; w < RW; ++w) {
    cw[w] = 0; cn[w] = 0;
    const uint32_t lo = 32 * w; const uint32_t cnt = lo >= L ? 0 : (L - lo < 32 ? L - lo : 32);
    // branch-free base code: upper-case, then (x >> 1) & 3 maps A,C,T,G -> 0,1,2,3; x ^ (x >> 1) swaps the last two
    auto put = [&](uint32_t i, uint32_t ch) {
      const uint32_t x = ch & 0xDFu;
      const uint32_t ok = (x == 0x41u) | (x == 0x43u) | (x == 0x47u) | (x == 0x54u);
      const uint32_t c2 = (x >> 1) & 3u; const uint32_t c = c2 ^ (c2 >> 1);
      cw[w] |= (uint64_t)(ok ? c : 0u) << (i * 2); cn[w] |= (ok ^ 1u) << i;
    };
    if (cnt == 32 && wide) {
      const Q4 q0 = *(const Q4*)(s0 + lo), q1 = *(const Q4*)(s0 + lo + 16);
      uint32_t d[9] = {q0.v[0], q0.v[1], q0.v[2], q0.v[3], q1.v[0], q1.v[1], q1.v[2], q1.v[3], 0u};
      if (mis) d[8] = *(const uint32_t*)(s0 + lo + 32);
#pragma unroll
      for (uint32_t q = 0; q < 8; ++q) {
        const uint32_t v = mis ? (uint32_t)((((uint64_t)d[q + 1] << 32) | d[q]) >> (8 * mis)) : d[q];
        put(4 * q, v & 0xFF); put(4 * q + 1, (v >> 8) & 0xFF); put(4 * q + 2, (v >> 16) & 0xFF); put(4 * q + 3, v >> 24);
      }
    } else if (cnt) {
      for (uint32_t i = 0; i < cnt; ++i) put(i, s[lo + i]);
    }
  }
  sq_u64x2* rp = (sq_u64x2*)(rpack + (size_t)e * RW);
#pragma unroll
  for (uint32_t w = 0; w < RW; w += 2) { sq_u64x2 v; v.x = cw[w]; v.y = cw[w + 1]; rp[w / 2] = v; }
  sq_u64x2* np = (sq_u64x2*)(rnmask + (size_t)e * (RW / 2));
#pragma unroll
  for (uint32_t w = 0; w < RW / 2; w += 2) { sq_u64x2 v; v.x = (uint64_t)cn[2 * w] | ((uint64_t)cn[2 * w + 1] << 32); v.y = (uint64_t)cn[2 * w + 2] | ((uint64_t)cn[2 * w + 3] << 32); np[w / 2] = v; }
  rlen[e] = (uint16_t)L;
  { uint32_t any = 0;
#pragma unroll
    for (uint32_t w = 0; w < RW; ++w) any |= cn[w];
    rany[e] = any ? 1 : 0; }   // [r6] k_seed2 asks this byte (64 consecutive ends: one sector) instead of reading the end's 32-byte mask
}
template <uint32_t RW>
__global__ void __launch_bounds__(256) k_pack(const uint8_t* __restrict__ seq, const uint64_t* __restrict__ seq_off, uint32_t nrec,
                       uint64_t* __restrict__ rpack, uint64_t* __restrict__ rnmask, uint16_t* __restrict__ rlen, uint8_t* __restrict__ rany, unsigned long long* __restrict__ stats) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nrec) pack_end<RW>(seq, seq_off, nrec, e, rpack, rnmask, rlen, rany, stats);
}

// ------------------------------------------------------------------------------------------------
// [r6] k_pack for the eight-word stride with the text of a wave's 64 read ends staged through LDS.  The thread-per-end kernel above asks for its end's bytes with 16-byte loads
// at a 100- or 150-byte stride and stores 16-byte pieces at a 64-byte stride: the counters show 52.7 M read and 50.5 M write requests to L2 per 10^7 ends (3 x the lines the
// bytes occupy) beside 1 536 vector instructions per end (profiles/r06_pack_counters.txt).  Here a wave copies the contiguous text of its ends into LDS with whole-line loads, every
// lane takes its end out of LDS (dwords at an odd stride: no bank conflicts) and packs four characters at a time, the packed words go back through LDS and leave as whole lines.
// A wave whose text does not fit PK_WCAP bytes (reads of more than ~159 bases) or holds an end beyond the stride runs the thread-per-end code.  Same words, masks, lengths, flags.
#define PK_WCAP 9728u    // 64 ends of 150 bases and the alignment slack; four blocks of four waves per CU
// 32 bases from nine dwords of LDS (the text from byte `mis` of sw[0] on) -> 2-bit codes + "not a base" bits
__device__ __forceinline__ void pack_word32(const uint32_t* sw, uint32_t mis, uint64_t& cw, uint32_t& cn) {
  cw = 0; cn = 0;
  uint32_t prev = sw[0];
#pragma unroll
  for (uint32_t q = 0; q < 8; ++q) {
    const uint32_t cur = sw[q + 1];
    const uint32_t v = mis ? (uint32_t)((((uint64_t)cur << 32) | prev) >> (8 * mis)) : prev;
    prev = cur;
    // upper-case; c2 = (x >> 1) & 3 maps A, C, T, G -> 0, 1, 2, 3; the character that code stands for is 0x41 + 2 c2 (+ 0x0F for T): any other byte is not a base
    const uint32_t x = v & 0xDFDFDFDFu;
    const uint32_t c2 = (x >> 1) & 0x03030303u;
    const uint32_t isT = (c2 >> 1) & ~c2 & 0x01010101u;
    const uint32_t expect = 0x41414141u + (c2 << 1) + (isT << 4) - isT;
    const uint32_t diff = x ^ expect;
    const uint32_t nz = ((((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) >> 7) & 0x01010101u;      // bit 0 of every byte that is not A, C, G or T
    const uint32_t code = (c2 ^ (c2 >> 1)) & 0x03030303u & ~(nz | (nz << 1));             // swap T and G (k_pack's code); a non-base packs as 0
    cw |= (uint64_t)((code | (code >> 6) | (code >> 12) | (code >> 18)) & 0xFFu) << (8 * q);
    cn |= ((nz | (nz >> 7) | (nz >> 14) | (nz >> 21)) & 0xFu) << (4 * q);
  }
}
__global__ void __launch_bounds__(256) k_pack8_staged(const uint8_t* __restrict__ seq, const uint64_t* __restrict__ seq_off, uint32_t nrec,
                       uint64_t* __restrict__ rpack, uint64_t* __restrict__ rnmask, uint16_t* __restrict__ rlen, uint8_t* __restrict__ rany, unsigned long long* __restrict__ stats) {
  __shared__ sq_u32x4 s_buf[4][(PK_WCAP + 64) / 16];
  const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t e0 = (blockIdx.x * 4 + wv) * 64;
  if (e0 >= nrec) return;
  const uint32_t ne = nrec - e0 < 64 ? nrec - e0 : 64;
  const uint32_t e = e0 + lane; const bool act = lane < ne;
  const uint64_t a = act ? seq_off[e] : 0, b = act ? seq_off[e + 1] : 0;
  const uint64_t A = __shfl(a, 0, 64), B = __shfl(b, (int)ne - 1, 64);
  const uintptr_t gA = (uintptr_t)seq + A; const uint32_t pre = (uint32_t)(gA & 15);
  const uint64_t span = pre + (B - A);
  const bool fits = __ballot(act && b - a > 256) == 0 && span <= PK_WCAP;
  if (!fits) { if (act) pack_end<8>(seq, seq_off, nrec, e, rpack, rnmask, rlen, rany, stats); return; }
  uint8_t* sb = reinterpret_cast<uint8_t*>(s_buf[wv]);
  { // the wave's text, whole 16-byte pieces (the first and the last piece of the batch's buffer byte by byte: nothing outside [seq, seq + seq_off[nrec]) is touched)
    const uint8_t* g16 = reinterpret_cast<const uint8_t*>(gA - pre); const uint8_t* gend = seq + seq_off[nrec];
    for (uint32_t off = lane * 16; off < (uint32_t)span; off += 1024) {
      const uint8_t* p = g16 + off; sq_u32x4 v;
      if (p >= seq && p + 16 <= gend) v = *reinterpret_cast<const sq_u32x4*>(p);
      else {
        auto word = [&](uint32_t k0) { uint32_t t = 0; for (uint32_t k = k0; k < k0 + 4; ++k) if (p + k >= seq && p + k < gend) t |= (uint32_t)p[k] << (8 * (k & 3)); return t; };
        v.x = word(0); v.y = word(4); v.z = word(8); v.w = word(12);
      }
      *reinterpret_cast<sq_u32x4*>(sb + off) = v;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  uint64_t cw[8]; uint32_t cn[8]; const uint32_t L = (uint32_t)(b - a);
  { const uint32_t o = pre + (uint32_t)(a - A), mis = o & 3; const uint32_t* sw = reinterpret_cast<const uint32_t*>(sb + (o & ~3u));
#pragma unroll
    for (uint32_t w = 0; w < 8; ++w) {
      cw[w] = 0; cn[w] = 0;
      const uint32_t lo = 32 * w; const uint32_t cnt = (!act || lo >= L) ? 0 : (L - lo < 32 ? L - lo : 32);
      if (cnt) {
        pack_word32(sw + 8 * w, mis, cw[w], cn[w]);
        if (cnt < 32) { cw[w] &= (1ull << (2 * cnt)) - 1; cn[w] &= (1u << cnt) - 1; }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  // the packed words back through LDS (every lane is done with the text): the wave's 64 x 64 bytes of words and 64 x 32 bytes of masks are contiguous in memory
  { sq_u64x2* lw = reinterpret_cast<sq_u64x2*>(sb + lane * 64);
#pragma unroll
    for (uint32_t w = 0; w < 8; w += 2) { sq_u64x2 v; v.x = cw[w]; v.y = cw[w + 1]; lw[w / 2] = v; }
    sq_u64x2* lm = reinterpret_cast<sq_u64x2*>(sb + 4096 + lane * 32);
#pragma unroll
    for (uint32_t w = 0; w < 4; w += 2) { sq_u64x2 v; v.x = (uint64_t)cn[2 * w] | ((uint64_t)cn[2 * w + 1] << 32); v.y = (uint64_t)cn[2 * w + 2] | ((uint64_t)cn[2 * w + 3] << 32); lm[w / 2] = v; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  { sq_u32x4* gw = reinterpret_cast<sq_u32x4*>(rpack + (size_t)e0 * 8); const sq_u32x4* lw = reinterpret_cast<const sq_u32x4*>(sb);
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) { const uint32_t x = i * 64 + lane; if (x < ne * 4) gw[x] = lw[x]; }
    sq_u32x4* gm = reinterpret_cast<sq_u32x4*>(rnmask + (size_t)e0 * 4); const sq_u32x4* lm = reinterpret_cast<const sq_u32x4*>(sb + 4096);
#pragma unroll
    for (uint32_t i = 0; i < 2; ++i) { const uint32_t x = i * 64 + lane; if (x < ne * 2) gm[x] = lm[x]; }
  }
  if (act) {
    rlen[e] = (uint16_t)L;
    uint32_t any = 0;
#pragma unroll
    for (uint32_t w = 0; w < 8; ++w) any |= cn[w];
    rany[e] = any ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// a1 — MemCollector::operator() (reference call site SalmonQuantify.cpp:1266-1275); SPEC §a1.
// Lane-level dynamic scheduling: the number of dictionary probes per read end ranges from 1 (a clean
// read inside one unitig) to ~70 (an unmappable read), and a wave waits for its slowest lane.  So a
// lane does ONE probe step per loop trip and, when its read end is finished, takes the next one
// from a global cursor (one atomic per wave via ballot).  The persistent grid keeps every lane busy
// until the batch is drained; results are keyed by read end, so they do not depend on scheduling.
// SEED_SPEC = probe positions laid out per trip (a template argument since round 3; 2 by default, SQ_SEED_SPEC=1|3|4 selects another instantiation for measurements)
// round 2, word-per-k-mer filter (measured per 4x10^6 pairs: 1 -> 7.7 ms, 2 -> 5.5 ms, 3 -> 5.8 ms, 4 -> 6.2 ms, 8 -> 7.7 ms: wider costs registers and wasted filter words)
template <int KT, int MT, int SEED_SPEC = 2>   // k, minimizer length fixed at compile time (0 = runtime): the kernel is instruction-issue bound
__global__ void k_seed(sq_dict_view d, const uint64_t* __restrict__ ctab_off, sq_map_params P, uint32_t nends,
                       const uint64_t* __restrict__ rpack, const uint64_t* __restrict__ rnmask, const uint16_t* __restrict__ rlen,
                       sq_unimem_dev* __restrict__ um, uint32_t* __restrict__ n_uni, uint32_t* __restrict__ n_proj,
                           unsigned long long* __restrict__ stats,
                           uint32_t* __restrict__ cursor, uint32_t rw, uint32_t us /* uni-MEM slots per end */) {
  const int k = KT ? KT : (int)P.k; const int alt = (int)P.alt_skip;
  const int lane = (int)(threadIdx.x & 63);
  uint32_t e = 0xFFFFFFFFu; bool have = false, drained = false;
  ReadView r; r.w = nullptr; r.nm = nullptr; r.L = 0;
  int pos = 0, skip_until = -1; uint32_t nu = 0, np = 0;
  unsigned long long tot_nu = 0, tot_look = 0;
  sq_unimem_dev* out = nullptr;
  uint32_t pool_next = 0, pool_end = 0; bool global_drained = false;   // wave-uniform: the wave's private run of read ends
  for (;;) {
    // refill: lanes without a read end take the next ones of the wave's run; the run is renewed 64 at a time with ONE
    // atomic (a per-trip atomic on the single cursor serialised the whole grid: ~700 k same-address atomics per batch)
    unsigned long long want = __ballot(!have && !drained);
    while (want) {
      if (pool_next == pool_end) {
        if (global_drained) break;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(cursor, 64u);
        base = (uint32_t)__shfl((int)base, 0, 64);
        if (base >= nends) { global_drained = true; break; }
        pool_next = base; pool_end = base + 64u < nends ? base + 64u : nends;
      }
      const uint32_t avail = pool_end - pool_next, nw = (uint32_t)__popcll(want);
      const uint32_t take = avail < nw ? avail : nw;
      const uint32_t rank = (uint32_t)__popcll(want & ((1ULL << lane) - 1));
      if (!have && !drained && rank < take) {
        e = pool_next + rank; have = true;
        r = read_view(rpack, rnmask, rlen, e, rw); pos = 0; skip_until = -1; nu = 0; np = 0; out = um + (size_t)e * us;
      }
      pool_next += take;
      want = __ballot(!have && !drained);
    }
    if (global_drained && !have) drained = true;
    if (!__ballot(have)) break;
    if (have) {
      const int L = r.L; bool done = false;
      if (!(L >= k && pos + k <= L && nu < us)) { done = true; if (nu >= us && L >= k && pos + k <= L) atomicAdd(&stats[ST_UNIOVER], 1ULL); }   // the walk is not over but the slots are: the host widens the slab and seeds the batch again
      else {
        // [r2] Most probes are misses that end at the membership filter, and where the walk goes after a miss does not depend on
        // memory: the next SEED_SPEC probe positions (N skips applied) are laid out first and their filter words requested
        // together; the walk then takes the first position the filter lets through.  Positions behind it are simply laid out again
        // on the next trip, so the look-ups performed — and counted — are those of the one-at-a-time walk.
        const int nspec = d.kfilter ? SEED_SPEC : 1;   // without a filter: one probe per trip, as before
        int cp[SEED_SPEC]; uint64_t ckm[SEED_SPEC], crc[SEED_SPEC], cmini[SEED_SPEC]; uint32_t cat[SEED_SPEC]; bool cv[SEED_SPEC], cpass[SEED_SPEC]; int p = pos; bool ended = false;
#pragma unroll
        for (int s2 = 0; s2 < SEED_SPEC; ++s2) {
          cv[s2] = false; cp[s2] = p; ckm[s2] = 0; crc[s2] = 0; cmini[s2] = 0; cat[s2] = 0; cpass[s2] = false;
          if (s2 < nspec) {
            while (!ended) {
              if (p + k > L) { ended = true; break; }
              const uint64_t nb = fetch_bits(r.nm, (uint32_t)p, (uint32_t)k);
              if (nb) { p = p + (63 - __clzll((long long)nb)) + 1; continue; }
              cv[s2] = true; break;
            }
            if (cv[s2]) {
              cp[s2] = p; ckm[s2] = sq_fetch_bases(r.w, (uint64_t)p, (uint32_t)k);
              if (p < skip_until) { int np2 = p + alt; if (np2 > skip_until) np2 = skip_until; p = np2; } else p += 1;
            }
          }
        }
        // [r3] the minimizer scan comes first: it picks the filter block (consecutive probes share it, so their words sit in one 64-byte
        // line) and is reused by the dictionary walk of the probe that passes
#pragma unroll
        for (int s2 = 0; s2 < SEED_SPEC; ++s2) {
          if (cv[s2]) {
            crc[s2] = sq_revcomp(ckm[s2], (uint32_t)k);
            sq_min_scan<KT, MT>(d, ckm[s2], crc[s2], &cmini[s2], &cat[s2]);
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < SEED_SPEC; ++s2) {
          if (cv[s2]) {
            if (d.kfilter) { const uint64_t h = sq_kf_hash(ckm[s2] < crc[s2] ? ckm[s2] : crc[s2]), msk = sq_kf_mask(h);
              cpass[s2] = (d.kfilter[sq_kf_word_of(cmini[s2], h, d.kfilter_words / SQ_KF_BLOCK_WORDS)] & msk) == msk; }
            else cpass[s2] = true;
          }
        }
        int pick = -1; uint32_t looked = 0;
#pragma unroll
        for (int s2 = 0; s2 < SEED_SPEC; ++s2) if (pick < 0 && cv[s2]) { ++looked; if (cpass[s2]) pick = s2; }
        tot_look += looked;
        uint64_t km = 0, krc = 0, kmini = 0; uint32_t kat = 0; int ppos = pos;
#pragma unroll
        for (int s2 = 0; s2 < SEED_SPEC; ++s2) if (s2 == pick) { km = ckm[s2]; krc = crc[s2]; kmini = cmini[s2]; kat = cat[s2]; ppos = cp[s2]; }
        if (pick < 0) {   // every laid-out probe was a miss, or the read ran out: the walk continues behind them
          pos = p;
          if (ended) done = true;
        } else {
          pos = ppos;
          uint64_t u; uint32_t off; int fw;
          if (!sq_dict_lookup_pre<KT, MT>(d, km, krc, kmini, kat, &u, &off, &fw)) {
            if (pos < skip_until) { int npos = pos + alt; if (npos > skip_until) npos = skip_until; pos = npos; } else pos += 1;
          } else {
            uint64_t ub, ue, ca, cb;   // the sector sq_dict_try has just brought in: both tables' bounds of unitig u
            if (d.uinfo) { sq_ld_pair(d.uinfo + 2 * u, &ub, &ca); sq_ld_pair(d.uinfo + 2 * u + 2, &ue, &cb); }
            else { sq_ld_pair(d.uoff + u, &ub, &ue); sq_ld_pair(ctab_off + u, &ca, &cb); }
            const int ulen = (int)(ue - ub);
            int len = k;
            int avail = fw ? min(L - (pos + len), ulen - ((int)off + len)) : min(L - (pos + len), (int)off - (len - k));
            bool mism = false;
            while (avail > 0) {
              int c = avail < 32 ? avail : 32;
              uint64_t rc_ = sq_fetch_bases(r.w, (uint64_t)(pos + len), (uint32_t)c);
              uint64_t nn = fetch_bits(r.nm, (uint32_t)(pos + len), (uint32_t)c);
              uint64_t uc;
              if (fw) uc = sq_fetch_bases(d.useq, ub + off + len, (uint32_t)c);
              else {
                int up = (int)off - 1 - (len - k);
                uc = sq_revcomp(sq_fetch_bases(d.useq, ub + (uint64_t)(up - c + 1), (uint32_t)c), (uint32_t)c);
              }
              uint64_t x = rc_ ^ uc; uint64_t mm = (x | (x >> 1)) & 0x5555555555555555ULL;
              int i1 = mm ? (__ffsll((long long)mm) - 1) / 2 : 64; int i2 = nn ? (__ffsll((long long)nn) - 1) : 64;
              int im = i1 < i2 ? i1 : i2;
              if (im < c) { len += im; mism = true; break; }
              len += c; avail -= c;
            }
            const bool rend = (pos + len >= L);
            const bool uend = !rend && !mism;
            sq_unimem_dev m;
            m.unitig = (uint32_t)u;
            m.qpos = (uint16_t)pos;
            m.len = (uint16_t)len;
            m.fw = (uint8_t)fw;
            m.ustart = fw ? off : (uint32_t)((int)off - (len - k));
            m.pad[0] = m.pad[1] = m.pad[2] = 0;
            const uint64_t occ = cb - ca;
            m.ctab_a = ca; m.cnt = occ <= P.max_occ ? (uint32_t)occ : 0u; m.ulen = (uint32_t)ulen;
            out[nu++] = m;
            if (occ <= P.max_occ) np += (uint32_t)occ;
            if (rend) done = true;
            else { int ee = pos + len; pos = pos + len - k + 1; skip_until = uend ? -1 : ee + 1; }
          }
        }
      }
      if (done) { n_uni[e] = nu; n_proj[e] = np; tot_nu += nu; have = false; }
    }
  }
  wave_stat_add(&stats[ST_SEEDS], tot_nu); wave_stat_add(&stats[ST_LOOKUPS], tot_look);
}

// ------------------------------------------------------------------------------------------------
// [r5] k_seed2 — the same walk with the lane's state where the lane can reach it.  Round 4's counters said what k_seed waits for: a lane
// holds a read end for ~5 trips of the loop, a few microseconds apart, and between two trips the other ~37 000 lanes of its XCD pull
// several 64-byte sectors each through a 4 MB L2 — so the lane's own read words and the filter block it asked a moment ago are gone
// when it comes back, and every trip began with two or three dependent trips to memory for data the lane had already had (PMC: 1.31 x
// the byte model).  Here:
//   * the packed words of the read end are loaded ONCE, at refill, into the lane's column of LDS (LW words; the N-mask stays in HBM and
//     is consulted only by the rare lane whose read has an N: one flag in a register);
//   * the 64-byte filter block of the lane's current minimizer is kept in LDS too, brought there by LDS-DMA (global_load_lds_dwordx4: the
//     block never passes through registers).  Consecutive probes of a walk mostly share their minimizer — that is what the blocked
//     filter was built for — so the run of misses across a sequencing error, or along an unmappable read, asks memory once per
//     minimizer instead of once per probe;
//   * the slot record of the minimizer comes from the minimizer table (sq_mtab_find: one sector) instead of pilot -> slot (two).
// Measured on MI355X, c2, 5x10^6 pairs per launch (profiles/r05_seed_variants.txt): k_seed 6.61 ms; read words in LDS 4.70; + minimizer
// table 4.56; + filter block through registers into LDS 4.71 (the eight ds_write cost what the saved loads gave), by LDS-DMA 4.40.
// The walk, and therefore every uni-MEM, is k_seed's: tests hold both kernels to the checker.  Read ends longer than 32 * LW bases raise
// ST_SEEDLW and the host seeds the batch again with the next wider instantiation (map.hip).
#define SEED_TB 64
template <int LW>
__device__ inline uint64_t seed_lds_bases(const uint64_t (*rd)[SEED_TB], uint32_t tx, uint32_t p, uint32_t n) {
  const uint32_t w = p >> 5, sh = (p & 31) * 2;
  const uint64_t a = rd[w][tx];
  const uint64_t b = (w + 1 < (uint32_t)LW) ? rd[w + 1][tx] : 0ULL;   // bits beyond the read are masked off below
  const uint64_t lo = (a >> sh) | (sh ? (b << (64 - sh)) : 0ULL);
  return lo & sq_kmask(n);
}
template <int KT, int MT, int SEED_SPEC, int LW>
__global__ void __launch_bounds__(SEED_TB) __attribute__((amdgpu_waves_per_eu(7))) k_seed2(sq_dict_view d, sq_map_params P, uint32_t nends,
                       const uint64_t* __restrict__ rpack, const uint64_t* __restrict__ rnmask, const uint16_t* __restrict__ rlen, const uint8_t* __restrict__ rany,
                       sq_unimem_dev* __restrict__ um, uint32_t* __restrict__ n_uni, uint32_t* __restrict__ n_proj,
                       unsigned long long* __restrict__ stats, uint32_t* __restrict__ cursor, uint32_t rw, uint32_t us) {
  static_assert(KT > 0 && MT > 0 && SEED_SPEC >= 1 && SEED_SPEC <= 2, "k_seed2 is the specialised kernel");
  __shared__ uint64_t s_rd[LW][SEED_TB];
  __shared__ sq_u64x2 s_fq[SQ_KF_BLOCK_WORDS / 2][SEED_TB];   // quarter q of lane t's filter block at [q][t] — the layout the LDS-DMA writes (wave base + lane x 16 bytes)
  constexpr int k = KT; const int alt = (int)P.alt_skip;
  const uint32_t tx = threadIdx.x; const int lane = (int)(tx & 63);
  const uint64_t nfb = d.kfilter_words / SQ_KF_BLOCK_WORDS;
  uint32_t e = 0xFFFFFFFFu; bool have = false, drained = false, anyN = false;
  int L = 0, pos = 0, skip_until = -1; uint32_t nu = 0, np = 0;
  uint64_t cur_blk = ~0ULL;   // the filter block in this lane's column of s_fq
  unsigned long long tot_nu = 0, tot_look = 0, tot_fill = 0;
  uint32_t pool_next = 0, pool_end = 0; bool global_drained = false;   // wave-uniform: the wave's private run of read ends
  for (;;) {
    unsigned long long want = __ballot(!have && !drained);
    while (want) {
      if (pool_next == pool_end) {
        if (global_drained) break;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(cursor, 64u);
        base = (uint32_t)__shfl((int)base, 0, 64);
        if (base >= nends) { global_drained = true; break; }
        pool_next = base; pool_end = base + 64u < nends ? base + 64u : nends;
      }
      const uint32_t avail = pool_end - pool_next, nw = (uint32_t)__popcll(want);
      const uint32_t take = avail < nw ? avail : nw;
      const uint32_t rank = (uint32_t)__popcll(want & ((1ULL << lane) - 1));
      if (!have && !drained && rank < take) {
        e = pool_next + rank; have = true; pos = 0; skip_until = -1; nu = 0; np = 0;
        L = rlen[e];
        // the read end moves into the lane's LDS column: all loads in flight together, 16 bytes each
        const sq_u64x2* rp = (const sq_u64x2*)(rpack + (size_t)e * rw);
        sq_u64x2 v[(LW + 1) / 2];         // (an odd LW reads one word more than it keeps: still inside the end's row of rw >= 8 words)
#pragma unroll
        for (int w = 0; w < (LW + 1) / 2; ++w) v[w] = rp[w];
        const uint8_t hasN = rany[e];   // [r6] k_pack's byte "the mask has a bit set": the 32-byte mask itself stays where it is unless the end has an N
        if (L > 32 * LW) { atomicMax(&stats[ST_SEEDLW], (unsigned long long)L); L = 0; }   // does not fit this instantiation: nothing is seeded, the host runs a wider one (it learns the longest such end)
#pragma unroll
        for (int w = 0; w < (LW + 1) / 2; ++w) { s_rd[2 * w][tx] = v[w].x; if (2 * w + 1 < LW) s_rd[2 * w + 1][tx] = v[w].y; }
        anyN = hasN != 0;
      }
      pool_next += take;
      want = __ballot(!have && !drained);
    }
    if (global_drained && !have) drained = true;
    if (!__ballot(have)) break;
    if (have) {
      bool done = false;
      const uint64_t* nm = rnmask + (size_t)e * (rw >> 1);   // only lanes with anyN look at it
      if (!(L >= k && pos + k <= L && nu < us)) { done = true; if (nu >= us && L >= k && pos + k <= L) atomicAdd(&stats[ST_UNIOVER], 1ULL); }
      else {
        int cp[SEED_SPEC]; uint64_t ckm[SEED_SPEC], crc[SEED_SPEC], cmini[SEED_SPEC]; uint32_t cat[SEED_SPEC]; bool cv[SEED_SPEC], cpass[SEED_SPEC]; int p = pos; bool ended = false;
#pragma unroll
        for (int s2 = 0; s2 < SEED_SPEC; ++s2) {
          cv[s2] = false; cp[s2] = p; ckm[s2] = 0; crc[s2] = 0; cmini[s2] = 0; cat[s2] = 0; cpass[s2] = false;
          if (!ended) {
            if (p + k > L) ended = true;
            else if (anyN) {
              while (!ended) {
                if (p + k > L) { ended = true; break; }
                const uint64_t nb = fetch_bits(nm, (uint32_t)p, (uint32_t)k);
                if (nb) { p = p + (63 - __clzll((long long)nb)) + 1; continue; }
                cv[s2] = true; break;
              }
            } else cv[s2] = true;
          }
          if (cv[s2]) {
            cp[s2] = p; ckm[s2] = seed_lds_bases<LW>(s_rd, tx, (uint32_t)p, (uint32_t)k);
            if (p < skip_until) { int np2 = p + alt; if (np2 > skip_until) np2 = skip_until; p = np2; } else p += 1;
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < SEED_SPEC; ++s2) {
          if (cv[s2]) {
            crc[s2] = sq_revcomp(ckm[s2], (uint32_t)k);
            sq_min_scan<KT, MT>(d, ckm[s2], crc[s2], &cmini[s2], &cat[s2]);
          }
        }
        // filter words.  The block of the LAST candidate laid out becomes the lane's block (the walk moves forward: that minimizer is the one the next
        // trips will ask about) and is fetched whole unless the lane already holds it; the candidate before it reads the block the lane held so far,
        // or the new one, or — rarely, a third minimizer — asks memory for its one word.  In a run of k-mers that share a minimizer no trip loads anything.
        uint64_t fblk[SEED_SPEC], fmsk[SEED_SPEC], fword[SEED_SPEC]; uint32_t fwi[SEED_SPEC];
#pragma unroll
        for (int s2 = 0; s2 < SEED_SPEC; ++s2) {
          fblk[s2] = 0; fmsk[s2] = 0; fwi[s2] = 0; fword[s2] = 0;
          if (cv[s2]) { const uint64_t h = sq_kf_hash(ckm[s2] < crc[s2] ? ckm[s2] : crc[s2]); fmsk[s2] = sq_kf_mask(h);
            fblk[s2] = sq_kf_word(sq_mix64(cmini[s2] ^ 0x6A09E667F3BCC909ULL), nfb); fwi[s2] = (uint32_t)(h >> 24) & (SQ_KF_BLOCK_WORDS - 1); }
        }
        {
          const bool two = SEED_SPEC > 1 && cv[SEED_SPEC - 1];
          const uint64_t blast = two ? fblk[SEED_SPEC - 1] : fblk[0]; const uint32_t wlast = two ? fwi[SEED_SPEC - 1] : fwi[0];
          const bool old0 = two && fblk[0] == cur_blk, far0 = two && !old0 && fblk[0] != blast;
          auto lds_word = [&](uint32_t wi) -> uint64_t { return ((const uint64_t*)&s_fq[wi >> 1][tx])[wi & 1]; };
          if (old0) fword[0] = lds_word(fwi[0]);
          if (far0) { fword[0] = d.kfilter[fblk[0] * SQ_KF_BLOCK_WORDS + fwi[0]]; ++tot_fill; }
          const bool fill = cv[0] && blast != cur_blk;
          __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the read of the block held so far is done before the DMA may overwrite it
          if (fill) {
            typedef __attribute__((address_space(3))) void* lds_vp; typedef const __attribute__((address_space(1))) void* glb_vp;
            const char* g = (const char*)(d.kfilter + blast * SQ_KF_BLOCK_WORDS);
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_amdgcn_global_load_lds((glb_vp)(g + 16 * q), (lds_vp)&s_fq[q][tx & ~63u], 16, 0, 0);
            cur_blk = blast; ++tot_fill;
          }
          __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the block has landed
          if (cv[0]) { const uint64_t w = lds_word(wlast); if (two) fword[SEED_SPEC - 1] = w; else fword[0] = w; }
          if (two && !old0 && !far0) fword[0] = lds_word(fwi[0]);
        }
#pragma unroll
        for (int s2 = 0; s2 < SEED_SPEC; ++s2) cpass[s2] = cv[s2] && (fword[s2] & fmsk[s2]) == fmsk[s2];
        int pick = -1; uint32_t looked = 0;
#pragma unroll
        for (int s2 = 0; s2 < SEED_SPEC; ++s2) if (pick < 0 && cv[s2]) { ++looked; if (cpass[s2]) pick = s2; }
        tot_look += looked;
        uint64_t km = 0, krc = 0, kmini = 0; uint32_t kat = 0; int ppos = pos;
#pragma unroll
        for (int s2 = 0; s2 < SEED_SPEC; ++s2) if (s2 == pick) { km = ckm[s2]; krc = crc[s2]; kmini = cmini[s2]; kat = cat[s2]; ppos = cp[s2]; }
        if (pick < 0) {   // every laid-out probe was a miss, or the read ran out: the walk continues behind them
          pos = p;
          if (ended) done = true;
        } else {
          pos = ppos;
          uint64_t u; uint32_t off; int fw;
          const uint64_t rec = sq_mtab_find(d.mtab, d.mtab_buckets, kmini);
          if (!sq_dict_lookup_rec<KT, MT>(d, km, krc, rec, kat, &u, &off, &fw)) {
            if (pos < skip_until) { int npos = pos + alt; if (npos > skip_until) npos = skip_until; pos = npos; } else pos += 1;
          } else {
            uint64_t ub, ue, ca, cb;   // the sector sq_dict_try has just brought in: both tables' bounds of unitig u
            sq_ld_pair(d.uinfo + 2 * u, &ub, &ca); sq_ld_pair(d.uinfo + 2 * u + 2, &ue, &cb);
            const int ulen = (int)(ue - ub);
            int len = k;
            int avail = fw ? min(L - (pos + len), ulen - ((int)off + len)) : min(L - (pos + len), (int)off - (len - k));
            bool mism = false;
            while (avail > 0) {
              const int c = avail < 32 ? avail : 32;
              const uint64_t rc_ = seed_lds_bases<LW>(s_rd, tx, (uint32_t)(pos + len), (uint32_t)c);
              const uint64_t nn = anyN ? fetch_bits(nm, (uint32_t)(pos + len), (uint32_t)c) : 0ULL;
              uint64_t uc;
              if (fw) uc = sq_fetch_bases(d.useq, ub + off + len, (uint32_t)c);
              else {
                const int up = (int)off - 1 - (len - k);
                uc = sq_revcomp(sq_fetch_bases(d.useq, ub + (uint64_t)(up - c + 1), (uint32_t)c), (uint32_t)c);
              }
              const uint64_t x = rc_ ^ uc; const uint64_t mm = (x | (x >> 1)) & 0x5555555555555555ULL;
              const int i1 = mm ? (__ffsll((long long)mm) - 1) / 2 : 64; const int i2 = nn ? (__ffsll((long long)nn) - 1) : 64;
              const int im = i1 < i2 ? i1 : i2;
              if (im < c) { len += im; mism = true; break; }
              len += c; avail -= c;
            }
            const bool rend = (pos + len >= L);
            const bool uend = !rend && !mism;
            sq_unimem_dev m;
            m.unitig = (uint32_t)u;
            m.qpos = (uint16_t)pos;
            m.len = (uint16_t)len;
            m.fw = (uint8_t)fw;
            m.ustart = fw ? off : (uint32_t)((int)off - (len - k));
            m.pad[0] = m.pad[1] = m.pad[2] = 0;
            const uint64_t occ = cb - ca;
            m.ctab_a = ca; m.cnt = occ <= P.max_occ ? (uint32_t)occ : 0u; m.ulen = (uint32_t)ulen;
            um[(size_t)e * us + nu] = m; ++nu;
            if (occ <= P.max_occ) np += (uint32_t)occ;
            if (rend) done = true;
            else { const int ee = pos + len; pos = pos + len - k + 1; skip_until = uend ? -1 : ee + 1; }
          }
        }
      }
      if (done) { n_uni[e] = nu; n_proj[e] = np; tot_nu += nu; have = false; }
    }
  }
  wave_stat_add(&stats[ST_SEEDS], tot_nu); wave_stat_add(&stats[ST_LOOKUPS], tot_look); wave_stat_add(&stats[ST_FILLS], tot_fill);
}

// val layout: len[0,10) q[10,20) fw[20] tid[32,64)
__device__ inline uint64_t mem_pack_val(uint32_t tid, uint32_t q, uint32_t len, uint32_t fw) {
  return ((uint64_t)tid << 32) | ((uint64_t)fw << 20) | ((uint64_t)q << 10) | len;
}
struct MemD { uint32_t tid; int32_t rpos; int32_t q; int32_t len; bool fw; };
__device__ inline MemD mem_decode(uint64_t key, uint64_t val, const uint64_t* ref_accum) {
  MemD m; m.tid = (uint32_t)(val >> 32); m.fw = (val >> 20) & 1; m.q = (int32_t)((val >> 10) & 1023); m.len = (int32_t)(val & 1023);
  m.rpos = (int32_t)((key & ((1ULL << 40) - 1)) - ref_accum[m.tid]); return m;
}

// a3 — joinReadsAndFilter; SPEC §a3. Two-phase (count / fill) enumeration.
__device__ inline bool pair_ok(const sq_map_params& P, const sq_chain_dev& x, const sq_chain_dev& y, int32_t* fl, bool* dove) {
  if (x.fw == y.fw) return false;  // mpol.noDiscordant
  const sq_chain_dev& fwc = x.fw ? x : y; const sq_chain_dev& rcc = x.fw ? y : x;
  if (rcc.pos < fwc.pos) { *dove = true; if (!P.allow_dovetail) return false; }
  int32_t f = rcc.pos + (int32_t)rcc.read_len - fwc.pos;
  if (f <= 0 || f > (int32_t)P.frag_len_max) return false;
  *fl = f; return true;
}
__device__ inline void cand_init(sq_cand_dev& c, double cov, uint32_t tid, uint32_t lc, uint32_t rc, uint32_t fl, uint8_t ms) {
  c.cov = cov; c.tid = tid; c.lc = lc; c.rc = rc; c.frag_len = fl; c.lscore = c.rscore = SQ_INVALID_SCORE; c.mate_status = ms;
  c.valid = 0; c.compat = 0; c.lfail = c.rfail = 0; c.pad[0] = c.pad[1] = c.pad[2] = 0; c.pad2 = 0;
}
template <bool FILL>
__device__ inline uint32_t join_fragment(const sq_map_params& P, const sq_chain_dev* lc, uint32_t nl, uint32_t lbase,
    const sq_chain_dev* rc, uint32_t nr, uint32_t rbase,
                                         sq_cand_dev* out, bool* dovetail) {
  uint32_t cnt = 0; bool dove = false;
  // pass 1: global best coverage over concordant pairs
  double best = -1.0;
  for (uint32_t i = 0, j = 0; i < nl && j < nr;) {
    uint32_t ti = lc[i].tid, tj = rc[j].tid;
    if (ti < tj) { ++i; continue; }
    if (ti > tj) { ++j; continue; }
    uint32_t i1 = i, j1 = j; while (i1 < nl && lc[i1].tid == ti) ++i1; while (j1 < nr && rc[j1].tid == ti) ++j1;
    for (uint32_t a = i; a < i1; ++a) for (uint32_t b = j; b < j1; ++b) {
      int32_t fl;
      if (!pair_ok(P, lc[a], rc[b], &fl, &dove)) continue;
      double cov = lc[a].score + rc[b].score;
      if (cov > best) best = cov;
    }
    i = i1; j = j1;
  }
  *dovetail = dove;
  if (best >= 0.0) {
    const double thr = P.consensus_frac * best;
    for (uint32_t i = 0, j = 0; i < nl && j < nr;) {
      uint32_t ti = lc[i].tid, tj = rc[j].tid;
      if (ti < tj) { ++i; continue; }
      if (ti > tj) { ++j; continue; }
      uint32_t i1 = i, j1 = j; while (i1 < nl && lc[i1].tid == ti) ++i1; while (j1 < nr && rc[j1].tid == ti) ++j1;
      double bt = 0.0; bool d2;
      for (uint32_t a = i; a < i1; ++a) for (uint32_t b = j; b < j1; ++b) {
        int32_t fl;
        if (!pair_ok(P, lc[a], rc[b], &fl, &d2)) continue;
        double cov = lc[a].score + rc[b].score;
        if (cov < thr) continue;
        if (cov > bt) bt = cov;
      }
      const double pthr = P.post_thr * bt;
      for (uint32_t a = i; a < i1; ++a) for (uint32_t b = j; b < j1; ++b) {
        int32_t fl;
        if (!pair_ok(P, lc[a], rc[b], &fl, &d2)) continue;
        double cov = lc[a].score + rc[b].score;
        if (cov < thr || cov < pthr) continue;
        if (FILL) cand_init(out[cnt], cov, ti, lbase + a, rbase + b, (uint32_t)fl, SQ_MS_PAIRED_END_PAIRED);
        ++cnt;
      }
      i = i1; j = j1;
    }
    return cnt;
  }
  if (!P.allow_orphans) return 0;
  double ob = 0.0;
  for (uint32_t a = 0; a < nl; ++a) if (lc[a].score > ob) ob = lc[a].score;
  for (uint32_t b = 0; b < nr; ++b) if (rc[b].score > ob) ob = rc[b].score;
  const double othr = P.orphan_thr * ob;
  // pad[0] remembers which end anchors an orphan candidate (1 left, 2 right): orphan recovery (k_recover) may turn the
  // candidate into a pair, and k_select still needs the two tid-sorted runs (left-anchored, then right-anchored)
  for (uint32_t a = 0; a < nl; ++a) if (lc[a].score >= othr) {
    if (FILL) {
      cand_init(out[cnt], lc[a].score, lc[a].tid, lbase + a, 0xFFFFFFFFu, 0, SQ_MS_PAIRED_END_LEFT);
      out[cnt].pad[0] = 1;
    }
    ++cnt;
  }
  for (uint32_t b = 0; b < nr; ++b) if (rc[b].score >= othr) {
    if (FILL) {
      cand_init(out[cnt], rc[b].score, rc[b].tid, 0xFFFFFFFFu, rbase + b, 0, SQ_MS_PAIRED_END_RIGHT);
      out[cnt].pad[0] = 2;
    }
    ++cnt;
  }
  return cnt;
}

// Single-pass join: the chains of a fragment are read from HBM once.  Concordant pairs (almost always
// < 8 per fragment) are held in registers for the consensus / post-merge filters; the candidate block of
// the fragment is carved out of one global array with a block-aggregated cursor (one atomic per block), so
// there is no count kernel, no scan and no second enumeration.
// [r2] Two kernels.  k_join2 takes the usual fragment — at most JB chains per end, at most JP concordant pairs, or orphans only —
// with straight-line code: the transcripts of all chains are requested at once and matched in registers.  Anything else (repeat
// families, single-end libraries) is put on a list and k_join2_rest runs the general merge (join_fragment<>) over that list, a lane
// per fragment among its own kind: one slow lane used to hold the other 63 of its wave (SQ counters: 21 % of the lanes active).
// (Tried in round 2: copying the compact form of a fragment's chains into thread-private LDS columns first — three loads per chain —
// and enumerating from there: 1.95 -> 2.46 ms per 10^6 pairs; the 40 KB of LDS per 128 threads cost more occupancy than the loads saved.)
#define JP 8
#define JB 8   // chains per end whose transcripts are matched in registers

// the block's share of the candidate array: wave totals meet in LDS, one cursor atomic per block (see block_stat_add)
__device__ inline uint64_t join_alloc(uint32_t cnt, unsigned long long* cursor) {
  __shared__ uint32_t s_wt[16]; __shared__ unsigned long long s_base;
  const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6), nwv = (int)((blockDim.x + 63) >> 6);
  uint32_t incl = cnt;
  for (int sft = 1; sft < 64; sft <<= 1) { uint32_t o = __shfl_up(incl, sft, 64); if (lane >= sft) incl += o; }
  if (lane == 63) s_wt[wv] = incl;
  __syncthreads();
  if (threadIdx.x == 0) { unsigned long long tot = 0; for (int i = 0; i < nwv; ++i) tot += s_wt[i]; s_base = tot ? atomicAdd(cursor, tot) : 0ULL; }
  __syncthreads();
  unsigned long long base = s_base;
  for (int i = 0; i < wv; ++i) base += s_wt[i];
  return base + (incl - cnt);
}

__global__ void __attribute__((amdgpu_waves_per_eu(5))) k_join2(sq_map_params P, uint32_t nfrag, uint32_t paired, const uint64_t* __restrict__ chain_off,
    const sq_chain_dev* __restrict__ chains,
    const uint32_t* __restrict__ n_chains,
                        uint32_t* __restrict__ n_cand, uint64_t* __restrict__ cand_start, sq_cand_dev* __restrict__ cands,
                            uint32_t* __restrict__ cand_frag,
                            uint64_t cand_cap,
                        uint8_t* __restrict__ frag_flags, unsigned long long* __restrict__ cursor, uint32_t* __restrict__ rest, uint32_t* __restrict__ nrest) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = f < nfrag;
  const int lane = (int)(threadIdx.x & 63);
  uint32_t cnt = 0; bool dove = false; int mode = 0;   // mode 1: pairs in registers, 3: orphans; `later`: left to k_join2_rest
  bool later = false;
  double pc[JP]; uint32_t pt[JP], pa[JP], pb[JP], pf[JP]; uint32_t np = 0; double best = -1.0, othr = 0.0;
  const sq_chain_dev* lc = nullptr; const sq_chain_dev* rc = nullptr; uint32_t nl = 0, nr = 0, lbase = 0, rbase = 0;
#pragma unroll
  for (int i = 0; i < JP; ++i) { pc[i] = 0.0; pt[i] = 0; pa[i] = 0; pb[i] = 0; pf[i] = 0; }
  if (act && !paired) later = true;
  else if (act) {
    const uint32_t e0 = 2 * f;
    { const sq_u64x2 co = *reinterpret_cast<const sq_u64x2*>(chain_off + e0); lbase = (uint32_t)co.x; rbase = (uint32_t)co.y;   // e0 is even: one 16-byte load
      const uint2 nc2 = *reinterpret_cast<const uint2*>(n_chains + e0); nl = nc2.x; nr = nc2.y; }
    lc = chains + lbase; rc = chains + rbase;
    if (nl > JB || nr > JB) later = true;
    else {
      // the chains are sorted by transcript on both ends, so visiting the matches in (a, b) order is the order of the general merge
      uint32_t lt[JB], rt[JB];
#pragma unroll
      for (int a = 0; a < JB; ++a) { lt[a] = (uint32_t)a < nl ? lc[a].tid : 0xFFFFFFFFu; rt[a] = (uint32_t)a < nr ? rc[a].tid : 0xFFFFFFFEu; }
      unsigned long long M = 0;
#pragma unroll
      for (int a = 0; a < JB; ++a)
#pragma unroll
        for (int b = 0; b < JB; ++b) if (lt[a] == rt[b]) M |= 1ULL << (a * JB + b);
      while (M) {
        const int bit = __ffsll((long long)M) - 1; M &= M - 1;
        const uint32_t a = (uint32_t)bit / JB, b = (uint32_t)bit % JB;
        int32_t fl; if (!pair_ok(P, lc[a], rc[b], &fl, &dove)) continue;
        const double cov = lc[a].score + rc[b].score; if (cov > best) best = cov;
        const uint32_t ti = lc[a].tid;
#pragma unroll
        for (int q = 0; q < JP; ++q) if ((uint32_t)q == np) { pc[q] = cov; pt[q] = ti; pa[q] = a; pb[q] = b; pf[q] = (uint32_t)fl; }
        ++np;
      }
      if (np > JP) later = true;
      else if (np > 0) {
        mode = 1;
        const double thr = P.consensus_frac * best;
#pragma unroll
        for (int q = 0; q < JP; ++q) {
          bool keep = (uint32_t)q < np && pc[q] >= thr;
          if (keep) { double bt = 0.0;
#pragma unroll
            for (int z = 0; z < JP; ++z) if ((uint32_t)z < np && pt[z] == pt[q] && pc[z] >= thr && pc[z] > bt) bt = pc[z];
            keep = pc[q] >= P.post_thr * bt; }
          if (keep) ++cnt; else if ((uint32_t)q < np) pf[q] = 0xFFFFFFFFu;   // dropped
        }
      } else if (P.allow_orphans && (nl || nr)) {   // no concordant pair: the orphan rule of join_fragment<>, scores requested together
        mode = 3;
        double ob = 0.0;
#pragma unroll
        for (int a = 0; a < JB; ++a) { const double sl = (uint32_t)a < nl ? lc[a].score : 0.0, sr = (uint32_t)a < nr ? rc[a].score : 0.0; if (sl > ob) ob = sl; if (sr > ob) ob = sr; }
        othr = P.orphan_thr * ob;
#pragma unroll
        for (int a = 0; a < JB; ++a) { if ((uint32_t)a < nl && lc[a].score >= othr) ++cnt; if ((uint32_t)a < nr && rc[a].score >= othr) ++cnt; }
      }
    }
  }
  // the rest list: one cursor atomic per block
  { __shared__ uint32_t s_rw[16]; __shared__ uint32_t s_rbase;
    const unsigned long long lm = __ballot(later); const int wv = (int)(threadIdx.x >> 6), nwv = (int)((blockDim.x + 63) >> 6);
    if (lane == 0) s_rw[wv] = (uint32_t)__popcll(lm);
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t tot = 0; for (int i = 0; i < nwv; ++i) { const uint32_t v = s_rw[i]; s_rw[i] = tot; tot += v; } s_rbase = tot ? atomicAdd(nrest, tot) : 0u; }
    __syncthreads();
    if (later) rest[s_rbase + s_rw[wv] + (uint32_t)__popcll(lm & ((1ULL << lane) - 1))] = f; }
  const uint64_t start = join_alloc(cnt, cursor);
  if (!act || later) return;
  n_cand[f] = cnt; cand_start[f] = start; frag_flags[f] = (uint8_t)(dove ? 1 : 0);
  if (cnt == 0 || start + cnt > cand_cap) return;   // overflow: the host re-runs with a larger array
  sq_cand_dev* out = cands + start;
  uint32_t w = 0;
  if (mode == 1) {
#pragma unroll
    for (int q = 0; q < JP; ++q) if ((uint32_t)q < np && pf[q] != 0xFFFFFFFFu) {
      cand_init(out[w], pc[q], pt[q], lbase + pa[q], rbase + pb[q], pf[q], SQ_MS_PAIRED_END_PAIRED);
      ++w;
    }
  } else {   // mode 3; pad[0] remembers which end anchors an orphan candidate (see join_fragment<>)
#pragma unroll
    for (int a = 0; a < JB; ++a) if ((uint32_t)a < nl) { const double sc = lc[a].score; if (sc >= othr) {
      cand_init(out[w], sc, lc[a].tid, lbase + (uint32_t)a, 0xFFFFFFFFu, 0, SQ_MS_PAIRED_END_LEFT); out[w].pad[0] = 1; ++w; } }
#pragma unroll
    for (int b = 0; b < JB; ++b) if ((uint32_t)b < nr) { const double sc = rc[b].score; if (sc >= othr) {
      cand_init(out[w], sc, rc[b].tid, 0xFFFFFFFFu, rbase + (uint32_t)b, 0, SQ_MS_PAIRED_END_RIGHT); out[w].pad[0] = 2; ++w; } }
  }
  for (uint32_t i = 0; i < cnt; ++i) cand_frag[start + i] = f;
}

// the fragments k_join2 left on the list: the general multi-pass enumeration, a lane per fragment
__global__ void k_join2_rest(sq_map_params P, uint32_t paired, const uint64_t* __restrict__ chain_off, const sq_chain_dev* __restrict__ chains,
                             const uint32_t* __restrict__ n_chains, uint32_t* __restrict__ n_cand, uint64_t* __restrict__ cand_start,
                             sq_cand_dev* __restrict__ cands, uint32_t* __restrict__ cand_frag, uint64_t cand_cap, uint8_t* __restrict__ frag_flags,
                             unsigned long long* __restrict__ cursor, const uint32_t* __restrict__ rest, const uint32_t* __restrict__ nrest) {
  const uint32_t nlist = *nrest;
  if (blockIdx.x * blockDim.x >= nlist) return;   // the grid is sized for the worst case: whole blocks leave here
  const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = li < nlist;
  const uint32_t f = act ? rest[li] : 0;
  uint32_t cnt = 0; bool dove = false;
  const sq_chain_dev* lc = nullptr; const sq_chain_dev* rc = nullptr; uint32_t nl = 0, nr = 0, lbase = 0, rbase = 0;
  if (act && paired) {
    const uint32_t e0 = 2 * f, e1 = 2 * f + 1;
    lbase = (uint32_t)chain_off[e0]; rbase = (uint32_t)chain_off[e1]; nl = n_chains[e0]; nr = n_chains[e1];
    lc = chains + lbase; rc = chains + rbase;
    cnt = join_fragment<false>(P, lc, nl, lbase, rc, nr, rbase, nullptr, &dove);
  } else if (act) { lbase = (uint32_t)chain_off[f]; nl = n_chains[f]; lc = chains + lbase; cnt = nl; }
  const uint64_t start = join_alloc(cnt, cursor);
  if (!act) return;
  n_cand[f] = cnt; cand_start[f] = start; frag_flags[f] = (uint8_t)(dove ? 1 : 0);
  if (cnt == 0 || start + cnt > cand_cap) return;
  sq_cand_dev* out = cands + start;
  if (paired) { bool d2; join_fragment<true>(P, lc, nl, lbase, rc, nr, rbase, out, &d2); }
  else for (uint32_t a = 0; a < nl; ++a) cand_init(out[a], lc[a].score, lc[a].tid, lbase + a, 0xFFFFFFFFu, 0, SQ_MS_SINGLE_END);
  for (uint32_t i = 0; i < cnt; ++i) cand_frag[start + i] = f;
}

// The fragments k_join2 left on the list, paired libraries: a group of 16 lanes per fragment.  The group copies what the join reads
// of the fragment's chains (transcript, position, strand, read length, score: 20 bytes per chain) into LDS once, coalesced, and every
// pass of join_fragment<> — best coverage, per-transcript best, count, fill — runs from there, a lane per left chain; the candidates come
// out in the same (left chain, right chain) order.  Fragments with more than JG_CAP chains on an end take join_fragment<> itself on lane 0.
#define JG 16
#define JG_CAP 48
__global__ void __launch_bounds__(256) k_join2_group(sq_map_params P, const uint64_t* __restrict__ chain_off, const sq_chain_dev* __restrict__ chains,
                             const uint32_t* __restrict__ n_chains, uint32_t* __restrict__ n_cand, uint64_t* __restrict__ cand_start,
                             sq_cand_dev* __restrict__ cands, uint32_t* __restrict__ cand_frag, uint64_t cand_cap, uint8_t* __restrict__ frag_flags,
                             unsigned long long* __restrict__ cursor, const uint32_t* __restrict__ rest, const uint32_t* __restrict__ nrest) {
  constexpr int GPB = 256 / JG, CP = JG_CAP + 1;
  const uint32_t nlist = *nrest;   // the list length is only known on the device: a fixed grid walks the list
  __shared__ uint32_t s_tid[GPB][2][CP]; __shared__ int32_t s_pos[GPB][2][CP]; __shared__ uint32_t s_meta[GPB][2][CP];   // meta: fw | read_len << 1
  __shared__ double s_sc[GPB][2][CP]; __shared__ double s_bt[GPB][CP]; __shared__ uint32_t s_cnt[GPB][CP];
  const int gi = (int)(threadIdx.x / JG), gl = (int)(threadIdx.x % JG);
  for (uint32_t b0 = blockIdx.x * GPB; b0 < nlist; b0 += gridDim.x * GPB) {
  const uint32_t li = b0 + (uint32_t)gi;
  const bool act = li < nlist;
  const uint32_t f = act ? rest[li] : 0;
  uint32_t nl = 0, nr = 0, lbase = 0, rbase = 0;
  if (act) { const uint32_t e0 = 2 * f; const sq_u64x2 co = *reinterpret_cast<const sq_u64x2*>(chain_off + e0); lbase = (uint32_t)co.x; rbase = (uint32_t)co.y;
    const uint2 nc2 = *reinterpret_cast<const uint2*>(n_chains + e0); nl = nc2.x; nr = nc2.y; }
  const sq_chain_dev* lc = chains + lbase; const sq_chain_dev* rc = chains + rbase;
  const bool big = nl > JG_CAP || nr > JG_CAP;
  uint32_t cnt = 0; bool dove = false;
  double best = -1.0, thr = 0.0, othr = 0.0;
  if (act && big) {
    if (gl == 0) cnt = join_fragment<false>(P, lc, nl, lbase, rc, nr, rbase, nullptr, &dove);
  } else if (act) {
    for (uint32_t i = (uint32_t)gl; i < nl + nr; i += JG) {
      const int side = i < nl ? 0 : 1; const uint32_t x = side ? i - nl : i; const sq_chain_dev* ch = (side ? rc : lc) + x;
      const sq_u32x4 h = reinterpret_cast<const sq_u32x4_a8*>(reinterpret_cast<const char*>(ch) + 16)->v;   // pos, first, pad2, n_mems | fw << 16 | ..
      s_tid[gi][side][x] = ch->tid; s_pos[gi][side][x] = (int32_t)h.x; s_meta[gi][side][x] = ((h.w >> 16) & 1u) | ((uint32_t)ch->read_len << 1); s_sc[gi][side][x] = ch->score;
    }
  }
  __syncthreads();
  // pair test of join_fragment<> / pair_ok from the LDS copy
  auto pair = [&](uint32_t a, uint32_t b, int32_t* fl, bool* dv) -> bool {
    const uint32_t ma = s_meta[gi][0][a], mb = s_meta[gi][1][b];
    if ((ma & 1u) == (mb & 1u)) return false;
    const bool afw = (ma & 1u) != 0;
    const int32_t fpos = afw ? s_pos[gi][0][a] : s_pos[gi][1][b], rpos = afw ? s_pos[gi][1][b] : s_pos[gi][0][a];
    const int32_t rlen = (int32_t)((afw ? mb : ma) >> 1);
    if (rpos < fpos) { *dv = true; if (!P.allow_dovetail) return false; }
    const int32_t fr = rpos + rlen - fpos;
    if (fr <= 0 || fr > (int32_t)P.frag_len_max) return false;
    *fl = fr; return true;
  };
  // right-chain range of left chain a: the chains of an end are sorted by transcript
  auto rrange = [&](uint32_t a, uint32_t* j0, uint32_t* j1) {
    const uint32_t t = s_tid[gi][0][a]; uint32_t lo = 0, hi = nr;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (s_tid[gi][1][m] < t) lo = m + 1; else hi = m; }
    uint32_t e = lo; while (e < nr && s_tid[gi][1][e] == t) ++e;
    *j0 = lo; *j1 = e;
  };
  const bool grp = act && !big;
  if (grp) {
    for (uint32_t a = (uint32_t)gl; a < nl; a += JG) {
      uint32_t j0, j1; rrange(a, &j0, &j1);
      for (uint32_t b = j0; b < j1; ++b) { int32_t fl; if (!pair(a, b, &fl, &dove)) continue; const double cov = s_sc[gi][0][a] + s_sc[gi][1][b]; if (cov > best) best = cov; }
    }
  }
  // group-wide best coverage and dovetail flag (all lanes take part: inactive groups carry the neutral values)
#pragma unroll
  for (int sft = 1; sft < JG; sft <<= 1) { const double o = __shfl_xor(best, sft, 64); if (o > best) best = o; const int d = __shfl_xor((int)dove, sft, 64); dove = dove || d; }
  const bool pairs = grp && best >= 0.0;
  if (pairs) {
    thr = P.consensus_frac * best;
    for (uint32_t a = (uint32_t)gl; a < nl; a += JG) {   // best coverage of left chain a among its pairs that pass the consensus threshold
      uint32_t j0, j1; rrange(a, &j0, &j1); double bt = 0.0; bool d2;
      for (uint32_t b = j0; b < j1; ++b) { int32_t fl; if (!pair(a, b, &fl, &d2)) continue; const double cov = s_sc[gi][0][a] + s_sc[gi][1][b]; if (cov < thr) continue; if (cov > bt) bt = cov; }
      s_bt[gi][a] = bt;
    }
  }
  __syncthreads();
  if (pairs) {
    for (uint32_t a = (uint32_t)gl; a < nl; a += JG) {   // per-transcript best: the left chains of one transcript are neighbours
      const uint32_t t = s_tid[gi][0][a]; double bt = s_bt[gi][a];
      for (uint32_t x = a; x-- > 0 && s_tid[gi][0][x] == t;) if (s_bt[gi][x] > bt) bt = s_bt[gi][x];
      for (uint32_t x = a + 1; x < nl && s_tid[gi][0][x] == t; ++x) if (s_bt[gi][x] > bt) bt = s_bt[gi][x];
      const double pthr = P.post_thr * bt;
      uint32_t j0, j1; rrange(a, &j0, &j1); uint32_t c = 0; bool d2;
      for (uint32_t b = j0; b < j1; ++b) { int32_t fl; if (!pair(a, b, &fl, &d2)) continue; const double cov = s_sc[gi][0][a] + s_sc[gi][1][b]; if (cov < thr || cov < pthr) continue; ++c; }
      s_cnt[gi][a] = c;
    }
  } else if (grp && P.allow_orphans) {
    double ob = 0.0;
    for (uint32_t i = (uint32_t)gl; i < nl + nr; i += JG) { const double sc = i < nl ? s_sc[gi][0][i] : s_sc[gi][1][i - nl]; if (sc > ob) ob = sc; }
    othr = ob;
  }
  // the orphan threshold needs the group maximum (neutral 0 elsewhere)
#pragma unroll
  for (int sft = 1; sft < JG; sft <<= 1) { const double o = __shfl_xor(othr, sft, 64); if (o > othr) othr = o; }
  othr = P.orphan_thr * othr;
  __syncthreads();
  // counts -> offsets (lane 0 of the group walks them: at most JG_CAP entries), total per fragment
  uint32_t tot = 0;
  if (pairs) {
    if (gl == 0) { for (uint32_t a = 0; a < nl; ++a) { const uint32_t c = s_cnt[gi][a]; s_cnt[gi][a] = tot; tot += c; } cnt = tot; }
  } else if (grp && P.allow_orphans && gl == 0) {
    for (uint32_t a = 0; a < nl; ++a) if (s_sc[gi][0][a] >= othr) ++tot;
    for (uint32_t b = 0; b < nr; ++b) if (s_sc[gi][1][b] >= othr) ++tot;
    cnt = tot;
  }
  const uint64_t start0 = join_alloc(gl == 0 ? cnt : 0u, cursor);   // ends with a barrier: the offsets in s_cnt are visible to the group
  const uint32_t lead = (uint32_t)((threadIdx.x & 63) & ~(JG - 1));
  const uint64_t start = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(start0 >> 32), (int)lead, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)start0, (int)lead, 64);
  cnt = (uint32_t)__shfl((int)cnt, (int)lead, 64);
  if (act && gl == 0) { n_cand[f] = cnt; cand_start[f] = start; frag_flags[f] = (uint8_t)(dove ? 1 : 0); }
  if (act && cnt != 0 && start + cnt <= cand_cap) {
  sq_cand_dev* out = cands + start;
  if (big) { if (gl == 0) { bool d2; join_fragment<true>(P, lc, nl, lbase, rc, nr, rbase, out, &d2); } }
  else if (pairs) {
    for (uint32_t a = (uint32_t)gl; a < nl; a += JG) {
      const uint32_t t = s_tid[gi][0][a]; double bt = s_bt[gi][a];
      for (uint32_t x = a; x-- > 0 && s_tid[gi][0][x] == t;) if (s_bt[gi][x] > bt) bt = s_bt[gi][x];
      for (uint32_t x = a + 1; x < nl && s_tid[gi][0][x] == t; ++x) if (s_bt[gi][x] > bt) bt = s_bt[gi][x];
      const double pthr = P.post_thr * bt;
      uint32_t j0, j1; rrange(a, &j0, &j1); uint32_t w = s_cnt[gi][a]; bool d2;
      for (uint32_t b = j0; b < j1; ++b) {
        int32_t fl; if (!pair(a, b, &fl, &d2)) continue; const double cov = s_sc[gi][0][a] + s_sc[gi][1][b]; if (cov < thr || cov < pthr) continue;
        cand_init(out[w], cov, t, lbase + a, rbase + b, (uint32_t)fl, SQ_MS_PAIRED_END_PAIRED); ++w;
      }
    }
  } else if (gl == 0) {   // orphans: pad[0] remembers which end anchors the candidate (see join_fragment<>)
    uint32_t w = 0;
    for (uint32_t a = 0; a < nl; ++a) if (s_sc[gi][0][a] >= othr) { cand_init(out[w], s_sc[gi][0][a], s_tid[gi][0][a], lbase + a, 0xFFFFFFFFu, 0, SQ_MS_PAIRED_END_LEFT); out[w].pad[0] = 1; ++w; }
    for (uint32_t b = 0; b < nr; ++b) if (s_sc[gi][1][b] >= othr) { cand_init(out[w], s_sc[gi][1][b], s_tid[gi][1][b], 0xFFFFFFFFu, rbase + b, 0, SQ_MS_PAIRED_END_RIGHT); out[w].pad[0] = 2; ++w; }
  }
  for (uint32_t i = (uint32_t)gl; i < cnt; i += JG) cand_frag[start + i] = f;
  }
  __syncthreads();   // the next fragment of the group reuses the LDS rows
  }
}

// ---- a4 scoring --------------------------------------------------------------------------------
struct ScoreCtx {
  const uint64_t* refseq; const uint64_t* ref_accum; const uint32_t* ref_len;
  const uint64_t* rpack; const uint64_t* rnmask; const uint16_t* rlen;
  const uint64_t* mkey; const uint64_t* mval; const uint32_t* mnext;
  sq_dp_item* dpq; uint32_t* counters; uint32_t dpq_cap;
  uint32_t rw;   // packing stride of rpack (words per read end)
};

// ---- a5 — recoverOrphans (SalmonQuantify.cpp:1356-1364) with the in-tree edlib infix aligner (src/edlib.cpp:290-372) ----
// Bit-parallel edit distance (Myers 1999 / Hyyro 2001): a column of the DP matrix is held as vertical +1/-1 delta
// words; the query is at most 256 bases = NW <= 4 words, all in registers.  `hin` is the horizontal delta entering the
// block from the row above it (0 on row 0 in infix mode, +1 in prefix mode); returns the delta leaving bit `obit`.
__device__ inline int myers_step(uint64_t& Pv, uint64_t& Mv, uint64_t Eq, int hin, int obit) {
  const uint64_t hneg = hin < 0 ? 1ull : 0ull, hpos = hin > 0 ? 1ull : 0ull;
  const uint64_t Xv = Eq | Mv;
  Eq |= hneg;
  const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
  uint64_t Ph = Mv | ~(Xh | Pv), Mh = Pv & Xh;
  const int hout = (int)((Ph >> obit) & 1) - (int)((Mh >> obit) & 1);
  Ph = (Ph << 1) | hpos;
  Mh = (Mh << 1) | hneg;
  Pv = Mh | ~(Xv | Ph);
  Mv = Ph & Xv;
  return hout;
}

// infix alignment of the strand-normalised mate (fw: as read, else reverse complement) inside the window
// [wstart, wstart + wl) of the 2-bit reference text; SPEC §a5: minimum edit distance <= k, FIRST end among the minima,
// and for that end the start that the backward prefix pass reports LAST (the longest span).  True if found.
template <int NW>
__device__ inline bool infix_align(const ReadView& r, bool fw, const uint64_t* __restrict__ refseq, uint64_t wstart, int wl, int k,
                                   int* start, int* end, int* dist = nullptr) {
  const int n = r.L;
  const int lastw = (n - 1) >> 6, lastb = (n - 1) & 63;
  uint64_t E0[NW], E1[NW], E2[NW], E3[NW], Pv[NW], Mv[NW];   // match masks of the four bases; vertical +1 / -1 deltas
#pragma unroll
  for (int w = 0; w < NW; ++w) { E0[w] = E1[w] = E2[w] = E3[w] = 0; Pv[w] = ~0ull; Mv[w] = 0; }
  for (int i = 0; i < n; ++i) {
    const uint32_t b = norm_base(r, fw, i);
    const uint64_t bit = 1ull << (i & 63);
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (w == (i >> 6)) {
        if (b == 0) E0[w] |= bit;
        else if (b == 1) E1[w] |= bit;
        else if (b == 2) E2[w] |= bit;
        else if (b == 3) E3[w] |= bit;          // b == 4 (N) stays out of every mask: it matches nothing
      }
  }
  // forward pass, infix mode: score = D[n][j], the best alignment of the whole query ending at window column j
  int score = n, best = -1, e0 = -1;
  for (int j = 0; j < wl; ++j) {
    const uint32_t tc = sq_fetch_base(refseq, wstart + (uint64_t)j);
    int h = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (w <= lastw) {
        const uint64_t eq = tc == 0 ? E0[w] : tc == 1 ? E1[w] : tc == 2 ? E2[w] : E3[w];
        h = myers_step(Pv[w], Mv[w], eq, h, w == lastw ? lastb : 63);
      }
    score += h;
    if (score <= k && (best < 0 || score < best)) { best = score; e0 = j; }
  }
  if (best < 0) return false;
  // backward prefix pass from e0: the reversed query against the reversed window prefix
#pragma unroll
  for (int w = 0; w < NW; ++w) { E0[w] = E1[w] = E2[w] = E3[w] = 0; Pv[w] = ~0ull; Mv[w] = 0; }
  for (int i = 0; i < n; ++i) {
    const uint32_t b = norm_base(r, fw, n - 1 - i);
    const uint64_t bit = 1ull << (i & 63);
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (w == (i >> 6)) {
        if (b == 0) E0[w] |= bit;
        else if (b == 1) E1[w] |= bit;
        else if (b == 2) E2[w] |= bit;
        else if (b == 3) E3[w] |= bit;
      }
  }
  const int m2 = min(e0 + 1, n + best);     // D'[n][j] >= j - n: columns past n + best cannot reach `best`
  score = n;
  int jlast = -1;
  for (int j = 1; j <= m2; ++j) {
    const uint32_t tc = sq_fetch_base(refseq, wstart + (uint64_t)(e0 - (j - 1)));
    int h = 1;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (w <= lastw) {
        const uint64_t eq = tc == 0 ? E0[w] : tc == 1 ? E1[w] : tc == 2 ? E2[w] : E3[w];
        h = myers_step(Pv[w], Mv[w], eq, h, w == lastw ? lastb : 63);
      }
    score += h;
    if (score == best) jlast = j;
  }
  if (jlast < 0) return false;
  *end = e0;
  *start = e0 - (jlast - 1);
  if (dist) *dist = best;
  return true;
}

// parity tap for the aligner alone (sq_debug_infix_align): case i = packed query i against window [toff[i], toff[i+1]) of `text`
__global__ void k_infix_cases(uint32_t ncases, const uint64_t* __restrict__ rpack, const uint64_t* __restrict__ rnmask,
    const uint16_t* __restrict__ rlen,
                              const uint64_t* __restrict__ text, const uint64_t* __restrict__ toff, const int32_t* __restrict__ kmax,
                                  int32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncases) return;
  const ReadView r = read_view(rpack, rnmask, rlen, i, SQ_READ_WORDS_MIN);   // the tap's queries are packed with the default stride (map.hip)
  int st = -1, en = -1, ed = -1;
  bool ok = false;
  const int wl = (int)(toff[i + 1] - toff[i]);
  if (r.L > 0 && wl > 0) {
    if (r.L <= 64) ok = infix_align<1>(r, true, text, toff[i], wl, kmax[i], &st, &en, &ed);
    else if (r.L <= 128) ok = infix_align<2>(r, true, text, toff[i], wl, kmax[i], &st, &en, &ed);
    else ok = infix_align<4>(r, true, text, toff[i], wl, kmax[i], &st, &en, &ed);
  }
  out[4 * i] = ok ? 1 : 0;
  out[4 * i + 1] = ok ? ed : -1;
  out[4 * i + 2] = ok ? st : -1;
  out[4 * i + 3] = ok ? en : -1;
}

// thread per fragment: orphan-only fragments with at most maxReadOccs candidates look for the missing mate of every
// anchor.  A recovered mate is written as a chain without MEMs at slab index rec_base + (anchor's chain index) — every
// chain anchors at most one orphan candidate — and the candidate becomes a proper pair.
#define SQ_RECOVER_WINDOW 1000
__global__ void k_recover(sq_map_params P, ScoreCtx S, uint32_t nfrag, const uint64_t* __restrict__ cand_off,
    const uint32_t* __restrict__ n_cand,
                          sq_cand_dev* __restrict__ cands, sq_chain_dev* __restrict__ chains, uint32_t rec_base,
                              unsigned long long* __restrict__ stats) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t rescued = 0;
  if (f < nfrag) {
    const uint32_t nc = n_cand[f];
    sq_cand_dev* C = cands + cand_off[f];
    if (nc && nc <= P.max_read_occs && C[0].pad[0] != 0) {     // orphan-only fragment, not "too many hits" (SalmonQuantify.cpp:1343-1356)
      for (uint32_t i = 0; i < nc; ++i) {
        const bool left = C[i].pad[0] == 1;                    // the anchor is the left end: look for the right mate
        const uint32_t ai = left ? C[i].lc : C[i].rc;
        const sq_chain_dev an = chains[ai];
        const uint32_t me = 2 * f + (left ? 1u : 0u);
        const ReadView r = read_view(S.rpack, S.rnmask, S.rlen, me, S.rw);
        const int ML = r.L;
        if (ML == 0) continue;
        const int Tlen = (int)S.ref_len[an.tid];
        const uint64_t g = S.ref_accum[an.tid];
        int ws, wl;                                            // window on the transcript: downstream of a forward anchor, upstream of a reverse one
        if (an.fw) {
          ws = max(0, an.pos);
          wl = min(SQ_RECOVER_WINDOW, Tlen - ws);
        } else {
          const int endPos = min(Tlen, an.pos + (int)an.read_len);
          ws = max(0, endPos - SQ_RECOVER_WINDOW);
          wl = endPos - ws;
        }
        if (wl <= 0) continue;
        const bool mfw = an.fw == 0;                           // the mate lies on the other strand
        int st = 0, en = 0;
        bool ok;
        if (ML <= 64) ok = infix_align<1>(r, mfw, S.refseq, g + (uint64_t)ws, wl, ML / 4, &st, &en);
        else if (ML <= 128) ok = infix_align<2>(r, mfw, S.refseq, g + (uint64_t)ws, wl, ML / 4, &st, &en);
        else ok = infix_align<4>(r, mfw, S.refseq, g + (uint64_t)ws, wl, ML / 4, &st, &en);
        if (!ok) continue;
        const int mpos = ws + st;
        const int fl = an.fw ? (mpos + ML - an.pos) : (an.pos + (int)an.read_len - mpos);
        if (fl <= 0 || fl > (int)P.frag_len_max) continue;     // same bound as a joined pair (SPEC §a3)
        sq_chain_dev m;
        m.score = 0.0; m.tid = an.tid; m.pos = mpos; m.last_end = ws + en + 1; m.first = 0; m.n_mems = 0;
        m.read_len = (uint16_t)ML; m.fw = mfw ? 1 : 0; m.pad[0] = m.pad[1] = m.pad[2] = 0; m.pad2 = 0;
        chains[rec_base + ai] = m;
        if (left) C[i].rc = rec_base + ai; else C[i].lc = rec_base + ai;
        C[i].mate_status = SQ_MS_PAIRED_END_PAIRED;
        C[i].frag_len = (uint32_t)fl;
        rescued = 1;
      }
    }
  }
  wave_stat_add(&stats[ST_RESCUED], rescued);
}


// ---- scoring a chain: written so that the lanes of a wave walk the same instruction stream ---------------------------------------
// [r2] The SQ counters of the first version (tools/profile_sq.sh) showed k_score parked in s_waitcnt 91 % of its wave cycles with 23 %
// of the lanes active: head, gap and tail regions were three inlined copies of the comparison, forward and reverse-complement windows
// two more, every second pool word sat behind its own branch, and a second walk queued the DP regions — each a separately serialised
// round of dependent loads.  Here every step of the walk produces at most ONE region descriptor per lane and a single copy of the
// comparison evaluates it; pool reads always load both words; the strand is a select; the next MEM is in flight while the current
// region is compared; regions that need the DP are remembered (two per end) and queued after the walk with their budgets.

// `n` (<= 32) bases from nt position p of a packed pool: both words come with one 16-byte load (no data-dependent branch around a load)
__device__ inline uint64_t fetch_bases_u(const uint64_t* __restrict__ pool, uint64_t p, uint32_t n) {
  const uint64_t w = p >> 5; const uint32_t sh = (uint32_t)(p & 31) * 2;
  const sq_u64x2 ab = reinterpret_cast<const sq_u64x2_a8*>(pool + w)->v;   // words w and w + 1 (every pool is padded by a word)
  const uint64_t lo = (ab.x >> sh) | (sh ? (ab.y << (64 - sh)) : 0ULL);   // bits of the second word beyond 2n are masked off: same value as sq_fetch_bases
  return lo & sq_kmask(n);
}
__device__ inline uint64_t fetch_bits_u(const uint64_t* __restrict__ m, uint32_t p, uint32_t n) {   // n <= 32 one-bit flags from p
  const uint32_t w = p >> 6, sh = p & 63;
  const sq_u64x2 ab = reinterpret_cast<const sq_u64x2_a8*>(m + w)->v;
  const uint64_t lo = (ab.x >> sh) | (sh ? (ab.y << (64 - sh)) : 0ULL);
  return lo & ((1ULL << n) - 1);
}
// mismatch count of the strand-normalised read window R[qlo, qlo+n) against the reference text [tlo, tlo+n).  A mismatch count does
// not depend on the order in which the aligned pairs are visited, so leftward regions are compared as forward windows.
__device__ inline int count_mm_u(const ReadView& r, bool fw, int qlo, const uint64_t* __restrict__ refseq, int64_t tlo, int n) {
  int mm = 0;
  for (int j = 0; j < n; j += 32) {
    const int c = (n - j) < 32 ? (n - j) : 32;
    const int p = fw ? (qlo + j) : (r.L - (qlo + j + c));   // rc: the read window whose reverse complement is R[qlo+j .. qlo+j+c)
    const uint64_t raw = fetch_bases_u(r.w, (uint64_t)p, (uint32_t)c);
    const uint64_t nb = fetch_bits_u(r.nm, (uint32_t)p, (uint32_t)c);
    const uint64_t t = fetch_bases_u(refseq, (uint64_t)(tlo + j), (uint32_t)c);
    const uint64_t q = fw ? raw : sq_revcomp(raw, (uint32_t)c);
    const uint64_t nn = fw ? nb : (__brevll(nb) >> (64 - c));
    const uint64_t x = q ^ t; const uint64_t d = (x | (x >> 1)) & 0x5555555555555555ULL;
    uint64_t ns = nn;   // spread the N flags to even bit positions and OR them in
    ns = (ns | (ns << 16)) & 0x0000FFFF0000FFFFULL;
    ns = (ns | (ns << 8)) & 0x00FF00FF00FF00FFULL;
    ns = (ns | (ns << 4)) & 0x0F0F0F0F0F0F0F0FULL;
    ns = (ns | (ns << 2)) & 0x3333333333333333ULL;
    ns = (ns | (ns << 1)) & 0x5555555555555555ULL;
    mm += __popcll(d | ns);
  }
  return mm;
}

struct ChainHead { uint32_t tid, first, pad2; int32_t pos; uint16_t n_mems; uint8_t fw, by_mask; };   // what scoring reads of a chain
// A record or window is requested with ONE load instruction: a second load that touches a cache line whose fill is still in flight
// parks the CU's in-order L1 until the fill arrives (TCP_PENDING_STALL_CYCLES was 72 % of k_score's cycles with field-by-field loads).
__device__ inline ChainHead chain_head(const sq_chain_dev* __restrict__ ch, uint32_t tid) {
  const sq_u32x4 v = reinterpret_cast<const sq_u32x4_a8*>(reinterpret_cast<const char*>(ch) + 16)->v;   // pos, first, pad2, n_mems | fw << 16 | pad[0] << 24
  ChainHead h; h.tid = tid; h.pos = (int32_t)v.x; h.first = v.y; h.pad2 = v.z; h.n_mems = (uint16_t)(v.w & 0xFFFFu); h.fw = (uint8_t)((v.w >> 16) & 0xFFu);
  h.by_mask = (uint8_t)(v.w >> 24);
  return h;
}
#define SC_QCAP 192   // DP regions a block stages in LDS (a block of 256 candidates queues ~60); the overflow goes straight to the global queue
struct DpStage { sq_dp_item* q; uint32_t* n; };   // LDS
__device__ inline void dp_enqueue(const ScoreCtx& S, const DpStage& Q, uint32_t cand, uint8_t end, bool fw, int mode, int qstart, int dir, int n,
    int64_t tstart, int tl, int32_t budget) {
  sq_dp_item it;
  it.cand = cand; it.end = end; it.mode = (uint8_t)mode; it.rc = fw ? 0 : 1; it.pad = 0;
  it.qstart = qstart; it.qdir = dir; it.n = n; it.tstart = tstart; it.tdir = dir; it.tl = tl; it.budget = budget;
  const uint32_t slot = wave_alloc(Q.n);   // LDS cursor
  if (slot < SC_QCAP) { Q.q[slot] = it; return; }
  const uint32_t g = wave_alloc(&S.counters[8]);   // (queues per DP height were tried: k_dp gained 0.15 ms, the extra same-address atomics cost k_score 1 ms)
  if (g < S.dpq_cap) S.dpq[g] = it;
}

// Score one chain against its read end (SPEC §a4).  The walk resolves every region the mismatch-count fast path can decide and sums
// an upper bound (ma * n) for the rest; if even that bound misses minScoreFraction the end is invalid and no DP is queued.
// Otherwise the remembered regions are queued, each with the lowest region score that could still make the end valid (k_dp stops
// early below it).  num_dp_alignments counts every region the fast path cannot decide, once.
#define SC_PEND 2
__device__ inline int32_t score_chain(const sq_map_params& P, const ScoreCtx& S, const DpStage& Q, const ChainHead& ch, int Tlen, int64_t g, uint64_t mem_base,
    uint32_t end_id, uint32_t cand, uint8_t end, uint32_t* ndp, uint8_t* fail) {
  const ReadView r = read_view(S.rpack, S.rnmask, S.rlen, end_id, S.rw);
  const int L = r.L; const bool fw = ch.fw != 0; const uint32_t nm = ch.n_mems;
  const int32_t minacc = (int32_t)(P.min_score_fraction * (double)(P.ma * L));
  const bool by_mask = ch.by_mask != 0;
  int64_t score = 0, ub = 0;
  // regions waiting for the DP: qstart | n << 9 | mode << 18 | leftward << 19; tstart - g; tl
  uint32_t pq[SC_PEND]; int32_t pt[SC_PEND], pl[SC_PEND]; uint32_t npend = 0;
#pragma unroll
  for (int i = 0; i < SC_PEND; ++i) { pq[i] = 0; pt[i] = 0; pl[i] = 0; }
  for (int pass = 0; pass < 2; ++pass) {
    const bool requeue = pass == 1;                           // only when an end has more than SC_PEND DP regions (rare)
    const int64_t fast_total = score, ub_total = ub;
    int64_t sc_fast = 0; ub = 0;
    int prevQ = 0, prevR = (nm == 0) ? ch.pos : 0;            // n_mems == 0: recovered mate (SPEC §a5), one extension alignment from its start
    uint32_t mbits = ch.pad2;
    uint32_t mi = by_mask ? ch.first + (uint32_t)(__ffs((int)mbits) - 1) : ch.first;
    uint64_t key = 0, val = 0;
    if (nm) { key = S.mkey[mem_base + mi]; val = S.mval[mem_base + mi]; }
    for (uint32_t s = 0; s <= nm; ++s) {
      bool have = false; int mode = 1, qstart = 0, dir = 1, n = 0, tl = 0; int32_t trel = 0;
      if (s < nm) {
        int qs = (int32_t)((val >> 10) & 1023), ln = (int32_t)(val & 1023);
        int rs = (int32_t)((int64_t)(key & ((1ULL << 40) - 1)) - g);        // the chain's transcript is known: no ref_accum look-up per MEM
        // the next MEM is requested before this step's region is compared
        if (s + 1 < nm) {
          if (by_mask) { mbits &= mbits - 1; mi = ch.first + (uint32_t)(__ffs((int)mbits) - 1); } else mi = S.mnext[mem_base + mi];
          key = S.mkey[mem_base + mi]; val = S.mval[mem_base + mi];
        }
        bool use = true;
        if (s == 0) {
          if (qs > 0) { const int ws = max(0, rs - qs - SQ_REF_EXTEND); have = true; mode = 1; qstart = qs - 1; dir = -1; n = qs; trel = rs - 1; tl = max(0, rs - ws); }
        } else {
          const int ov = max(0, max(prevQ - qs, prevR - rs));
          if (ov > 0) { qs += ov; rs += ov; ln -= ov; if (ln <= 0) use = false; }
          if (use) { const int gq = qs - prevQ, gr = rs - prevR; if (gq > 0 || gr > 0) { have = true; mode = 0; qstart = prevQ; dir = 1; n = gq; trel = prevR; tl = gr; } }
        }
        if (use) { sc_fast += (int64_t)P.ma * ln; prevQ = qs + ln; prevR = rs + ln; }
      } else if (prevQ < L) {
        const int tail = L - prevQ; const int we = min(Tlen, prevR + tail + SQ_REF_EXTEND);
        have = true; mode = 1; qstart = prevQ; dir = 1; n = tail; trel = prevR; tl = max(0, we - prevR);
      }
      // the one copy of the comparison: lanes without a comparable region run it with n = 0
      const bool cmp = have && n > 0 && ((mode == 0 && tl == n) || (mode == 1 && tl >= n));
      const int qlo = dir > 0 ? qstart : qstart - n + 1;
      const int64_t tlo = g + (dir > 0 ? (int64_t)trel : (int64_t)trel - n + 1);
      const int mm = count_mm_u(r, fw, qlo, S.refseq, tlo, cmp ? n : 0);
      if (have) {
        const int lim = (mode == 0) ? (2 * (P.go + P.ge) + P.ma) : (P.go + P.ge);
        if (n == 0 && mode == 1) { /* nothing to align: 0 */ }
        else if (cmp && mm * (P.ma - P.mp) <= lim) sc_fast += P.ma * (n - mm) + P.mp * mm;
        else {
          if (!requeue) ++*ndp;
          // trivial DP outcomes that need no matrix (mirrors dp_align's early returns)
          if (n == 0) sc_fast += (tl == 0) ? 0 : (tl <= P.bw ? -(P.go + P.ge * tl) : SQ_NEG_INF);
          else if (tl == 0) sc_fast += (n <= P.bw) ? -(P.go + P.ge * n) : SQ_NEG_INF;
          else {
            ub += (int64_t)P.ma * n;
            if (requeue) {
              const int32_t budget = (int32_t)max((int64_t)SQ_NEG_INF, (int64_t)minacc - (fast_total + ub_total - (int64_t)P.ma * n));
              dp_enqueue(S, Q, cand, end, fw, mode, qstart, dir, n, g + trel, tl, budget);
            } else {
              const uint32_t q = (uint32_t)qstart | ((uint32_t)n << 9) | ((uint32_t)mode << 18) | (dir < 0 ? (1u << 19) : 0u);
#pragma unroll
              for (int i = 0; i < SC_PEND; ++i) if ((uint32_t)i == npend) { pq[i] = q; pt[i] = trel; pl[i] = tl; }
              ++npend;
            }
          }
        }
      }
    }
    score = sc_fast;
    if (requeue) break;
    if (ub == 0) break;                                     // nothing needs the DP
    if (score + ub < (int64_t)minacc || score < -(1 << 29)) { *fail = 1; break; }   // cannot reach minScoreFraction: invalid without any DP
    if (npend <= SC_PEND) {
#pragma unroll
      for (int i = 0; i < SC_PEND; ++i) if ((uint32_t)i < npend) {
        const int n = (int)((pq[i] >> 9) & 511), mode = (int)((pq[i] >> 18) & 1), dir = (pq[i] >> 19) & 1 ? -1 : 1;
        const int32_t budget = (int32_t)max((int64_t)SQ_NEG_INF, (int64_t)minacc - (score + ub - (int64_t)P.ma * n));
        dp_enqueue(S, Q, cand, end, fw, mode, (int)(pq[i] & 511), dir, n, g + pt[i], pl[i], budget);
      }
      break;
    }
  }
  if (score < -(1 << 30)) score = -(1 << 30);
  return (int32_t)score;
}

__device__ inline bool joint_compat(const sq_map_params& P, bool orphan, bool isLeft, bool lfw, bool rfw) {  // SalmonQuantify.cpp:1467-1516
  const uint8_t s = P.lib_strand;
  bool c = (s == 4) ? (orphan ? true : (lfw != rfw)) : false;
  if (c) return true;
  if (orphan) {
    if (s == 0) return (isLeft && lfw) || (!isLeft && !rfw);
    if (s == 1) return (isLeft && !lfw) || (!isLeft && rfw);
    return false;
  }
  if (s == 0) return lfw && !rfw;
  if (s == 1) return !lfw && rfw;
  return false;
}
__device__ inline bool compat_se(const sq_map_params& P, bool fwd, uint8_t ms) {  // SalmonUtils.cpp:195-268
  const uint8_t s = P.lib_strand, o = P.lib_orient;
  switch (ms) {
    case SQ_MS_SINGLE_END: return fwd ? (s == 4 || s == 2) : (s == 4 || s == 3);
    case SQ_MS_PAIRED_END_LEFT: if (o == 0) return s == 4 || (s == 2 && fwd) || (s == 3 && !fwd);
    return fwd ? (s == 4 || s == 0) : (s == 4 || s == 1);
    case SQ_MS_PAIRED_END_RIGHT: if (o == 0) return s == 4 || (s == 2 && fwd) || (s == 3 && !fwd);
    return fwd ? (s == 4 || s == 1) : (s == 4 || s == 0);
    default: return false;
  }
}

// (126 VGPRs: four waves per SIMD.  Capping it at five waves (amdgpu_waves_per_eu) spills 108 bytes per lane: 4.0 instead of 3.0 ms per 4x10^6 pairs.)
__global__ void k_score(sq_map_params P, ScoreCtx S, uint64_t ncand, uint32_t paired, const uint64_t* __restrict__ mem_off,   // launched with blocks of 256 (the LDS stages below are sized for them); __launch_bounds__(256) doubled its time (6.0 vs 3.0 ms)
    const uint64_t* __restrict__ cand_off,
    uint32_t nfrag,
                        const sq_chain_dev* __restrict__ chains, sq_cand_dev* __restrict__ cands, const uint32_t* __restrict__ cand_frag,
                            unsigned long long* __restrict__ stats) {
  uint64_t ci = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t ndp = 0;
  // the block's 256 candidate records (12 KB, contiguous) come in through LDS with coalesced 16-byte loads, every cache line requested
  // once; each thread then takes its own 48 bytes.  They leave the same way.
  __shared__ sq_u32x4 s_c[256 * 3];
  __shared__ sq_dp_item s_q[SC_QCAP]; __shared__ uint32_t s_qn, s_qbase; __shared__ unsigned long long s_ndp;
  static_assert(sizeof(sq_cand_dev) == 48, "sq_cand_dev is moved as three 16-byte pieces");
  static_assert(sizeof(sq_dp_item) == 40, "sq_dp_item is copied out as five 8-byte words");
  if (threadIdx.x == 0) { s_qn = 0; s_ndp = 0; }
  const DpStage Q{s_q, &s_qn};
  const uint64_t blk0 = (uint64_t)blockIdx.x * blockDim.x;                       // first candidate of the block
  const uint64_t npiece = (ncand - blk0 < blockDim.x ? ncand - blk0 : (uint64_t)blockDim.x) * 3;
  sq_u32x4* gsrc = reinterpret_cast<sq_u32x4*>(cands + blk0);
#pragma unroll
  for (int k = 0; k < 3; ++k) { const uint32_t i = (uint32_t)k * blockDim.x + threadIdx.x; if (i < npiece) s_c[i] = gsrc[i]; }
  __syncthreads();
  sq_cand_dev c;
  { sq_u32x4* cp = reinterpret_cast<sq_u32x4*>(&c); cp[0] = s_c[threadIdx.x * 3]; cp[1] = s_c[threadIdx.x * 3 + 1]; cp[2] = s_c[threadIdx.x * 3 + 2]; }
  if (ci < ncand) {
    const uint32_t f = cand_frag[ci];
    const bool orphan = c.mate_status != SQ_MS_PAIRED_END_PAIRED;
    const bool hasL = c.lc != 0xFFFFFFFFu, hasR = c.rc != 0xFFFFFFFFu;
    // both chains and their transcripts' bounds are requested up front: the right end's loads are in flight while the left end is scored
    ChainHead hl{}, hr{};
    if (hasL) hl = chain_head(chains + c.lc, c.tid);
    if (hasR) hr = chain_head(chains + c.rc, c.tid);
    int Tl = 0, Tr = 0; int64_t gl = 0, gr = 0;
    if (hasL) { Tl = (int)S.ref_len[hl.tid]; gl = (int64_t)S.ref_accum[hl.tid]; }
    if (hasR) { Tr = (int)S.ref_len[hr.tid]; gr = (int64_t)S.ref_accum[hr.tid]; }
    const bool lfw = hasL ? hl.fw != 0 : false, rfw = hasR ? hr.fw != 0 : false;
    const int32_t lpos_ = hasL ? hl.pos : 0, rpos_ = hasR ? hr.pos : 0;
    bool isc = paired ? joint_compat(P, orphan, hasL, lfw, rfw) : compat_se(P, lfw, SQ_MS_SINGLE_END);
    c.compat = isc;
    // the joining coverage has done its work (k_join2): its 8 bytes now carry the chains' implied positions and pad[1] their strands,
    // so the per-fragment selection (k_select) builds the alignment records without touching the chain slabs again
    c.cov = __longlong_as_double((long long)(((unsigned long long)(uint32_t)rpos_ << 32) | (unsigned long long)(uint32_t)lpos_));
    c.pad[1] = (uint8_t)((lfw ? 1 : 0) | (rfw ? 2 : 0));
    if (!isc && P.ignore_incompat) { c.valid = 0; c.lfail = c.rfail = 2; }   // 2 = skipped (not scored)
    else {
      uint32_t e0 = paired ? 2 * f : f, e1 = 2 * f + 1;
      c.lfail = c.rfail = 0;
      uint64_t mo0, mo1 = 0;
      if (paired) { const sq_u64x2 mo = *reinterpret_cast<const sq_u64x2*>(mem_off + e0); mo0 = mo.x; mo1 = mo.y; } else mo0 = mem_off[e0];
      if (hasL) c.lscore = score_chain(P, S, Q, hl, Tl, gl, mo0, e0, (uint32_t)ci, 0, &ndp, &c.lfail);
      // an end that already failed makes the pair invalid: the mate's DP regions are not needed
      // (SPEC §a4: the pair is dropped either way; only num_dp_alignments would differ, so the mate is still scored)
      if (hasR) c.rscore = score_chain(P, S, Q, hr, Tr, gr, mo1, e1, (uint32_t)ci, 1, &ndp, &c.rfail);
    }
  }
  { const sq_u32x4* cp = reinterpret_cast<const sq_u32x4*>(&c); s_c[threadIdx.x * 3] = cp[0]; s_c[threadIdx.x * 3 + 1] = cp[1]; s_c[threadIdx.x * 3 + 2] = cp[2]; }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; ++k) { const uint32_t i = (uint32_t)k * blockDim.x + threadIdx.x; if (i < npiece) gsrc[i] = s_c[i]; }
  // the block's DP regions take one range of the global queue (one atomic), copied out coalesced; the counter as well
  block_stat_add(&s_ndp, ndp);
  const uint32_t nq = s_qn < SC_QCAP ? s_qn : SC_QCAP;   // stable: every enqueue happened before the barrier above
  if (threadIdx.x == 0) s_qbase = nq ? atomicAdd(&S.counters[8], nq) : 0u;
  __syncthreads();
  if (threadIdx.x == 0 && s_ndp) atomicAdd(&stats[ST_DP], s_ndp);
  { const uint64_t* src = reinterpret_cast<const uint64_t*>(s_q); uint64_t* dst = reinterpret_cast<uint64_t*>(S.dpq + s_qbase);
    for (uint32_t i = threadIdx.x; i < nq * 5; i += blockDim.x) if (s_qbase + i / 5 < S.dpq_cap) dst[i] = src[i]; }
}

// banded Gotoh in registers: band index b = j - i + W, W = SQ_MAX_BAND (runtime bw <= W). SPEC §a4.
// One thread per queued region; both DP rows (H, F) live in VGPRs (launch bound 64 -> no spills) and
// the 31 target bases under the band ride in one 64-bit window that shifts by one base per row, so a
// row costs one query base + one target base fetch instead of 31 gathers.
// [r2] The rows of a DP region (its query length) decide how long its lane runs; in queue order a wave waited for its longest region with
// 44 % of the lanes active.  A counting sort of the queue by length class (longest first) makes the waves uniform: per-block LDS
// histograms, one scan (scan_kernels.h), a scatter of region indices; k_dp reads its region through the permutation.
#define DP_CLASSES 32
__device__ inline uint32_t dp_class(int n) { const uint32_t c = (uint32_t)n >> 3; return (DP_CLASSES - 1) - (c < DP_CLASSES - 1 ? c : DP_CLASSES - 1); }
__global__ void __launch_bounds__(256) k_dp_hist(const sq_dp_item* __restrict__ q, uint32_t n, uint32_t chunk, uint32_t nb, uint32_t* __restrict__ bh) {
  __shared__ uint32_t h[DP_CLASSES];
  if (threadIdx.x < DP_CLASSES) h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t b0 = blockIdx.x * chunk, b1 = b0 + chunk < n ? b0 + chunk : n;
  for (uint32_t i = b0 + threadIdx.x; i < b1; i += blockDim.x) atomicAdd(&h[dp_class(q[i].n)], 1u);
  __syncthreads();
  if (threadIdx.x < DP_CLASSES) bh[threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];   // class-major: the scan yields (class, block) offsets
}
__global__ void __launch_bounds__(256) k_dp_scatter(const sq_dp_item* __restrict__ q, uint32_t n, uint32_t chunk, uint32_t nb, const uint64_t* __restrict__ off,
    uint32_t* __restrict__ perm) {
  __shared__ uint32_t h[DP_CLASSES];
  if (threadIdx.x < DP_CLASSES) h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t b0 = blockIdx.x * chunk, b1 = b0 + chunk < n ? b0 + chunk : n;
  for (uint32_t i = b0 + threadIdx.x; i < b1; i += blockDim.x) {
    const uint32_t c = dp_class(q[i].n); const uint32_t r = atomicAdd(&h[c], 1u);
    perm[off[c * nb + blockIdx.x] + r] = i;
  }
}

__device__ inline uint32_t dp_tbase(const uint64_t* refseq, const sq_dp_item& it, int x) {   // target base x of the region (0 when outside)
  if (x < 0 || x >= it.tl) return 0u;
  return sq_fetch_base(refseq, (uint64_t)(it.tstart + (int64_t)it.tdir * x));
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) k_dp_general(sq_map_params P, ScoreCtx S, uint32_t nitems, sq_cand_dev* __restrict__ cands,
    const uint32_t* __restrict__ cand_frag,
    uint32_t paired, const uint32_t* __restrict__ perm) {
  uint32_t ii = blockIdx.x * blockDim.x + threadIdx.x;
  if (ii >= nitems) return;
  const sq_dp_item it = S.dpq[perm[ii]];   // regions in the order of k_dp_scatter: the 64 lanes of a wave run about the same number of rows
  const uint32_t f = cand_frag[it.cand];
  const uint32_t end_id = paired ? 2 * f + it.end : f;
  ReadView r = read_view(S.rpack, S.rnmask, S.rlen, end_id, S.rw);
  const bool fw = it.rc == 0;
  const int n = it.n, tl = it.tl, w = P.bw, go = P.go, ge = P.ge;
  constexpr int W = SQ_MAX_BAND, BW = 2 * SQ_MAX_BAND + 1;
  int32_t Hp[BW + 1], Fp[BW + 1];
#pragma unroll
  for (int b = 0; b <= BW; ++b) { Hp[b] = SQ_NEG_INF; Fp[b] = SQ_NEG_INF; }
#pragma unroll
  for (int b = 0; b < BW; ++b) { int j = b - W; if (j >= 0 && j <= tl && j <= w) Hp[b] = (j == 0) ? 0 : -(go + ge * j); }
  // window for row i holds target bases x = i - W - 1 + b (b = 0..30) at bits [2b, 2b+1]; row 1 -> x = b - W
  uint64_t twin = 0;
#pragma unroll
  for (int b = 0; b < BW; ++b) twin |= (uint64_t)dp_tbase(S.refseq, it, b - W) << (2 * b);
  bool hopeless = false;
  // the query and the target advance one base per row: their packed words are fetched once per 32 rows, not once per row
  int q_wi = -1; uint64_t q_w = 0, q_n = 0; int64_t t_wi = -1; uint64_t t_w = 0;
  for (int i = 1; i <= n; ++i) {
    uint32_t qb;
    { const int x = it.qstart + it.qdir * (i - 1); const int idx = fw ? x : r.L - 1 - x; const int wi = idx >> 5;
      if (wi != q_wi) { q_wi = wi; q_w = r.w[wi]; q_n = r.nm[idx >> 6]; }
      const uint32_t b = ((q_n >> (idx & 63)) & 1) ? 4u : ((uint32_t)(q_w >> ((idx & 31) * 2)) & 3u);
      qb = fw ? b : (b > 3 ? 4u : 3u - b); }
    int32_t left_h = SQ_NEG_INF, left_e = SQ_NEG_INF, rowmax = SQ_NEG_INF;
#pragma unroll
    for (int b = 0; b < BW; ++b) {
      const int j = i + b - W;
      int32_t hv = SQ_NEG_INF, ev = SQ_NEG_INF, fv = SQ_NEG_INF;
      const bool inband = (j >= 0 && j <= tl && (b - W) >= -w && (b - W) <= w);
      if (inband) {
        if (j == 0) { hv = (i <= w) ? -(go + ge * i) : SQ_NEG_INF; fv = hv; }
        else {
          ev = max(left_e, left_h - go) - ge;
          fv = max(Fp[b + 1], Hp[b + 1] - go) - ge;
          const uint32_t tb = (uint32_t)(twin >> (2 * b)) & 3u;
          const int32_t sc = (qb == tb) ? P.ma : P.mp;   // qb == 4 (N) never equals a target base
          hv = max(Hp[b] + sc, max(ev, fv));
          ev = max(ev, SQ_NEG_INF); fv = max(fv, SQ_NEG_INF); hv = max(hv, SQ_NEG_INF);
        }
      }
      Hp[b] = hv; Fp[b] = fv; left_h = hv; left_e = ev; rowmax = max(rowmax, max(hv, max(ev, fv)));
    }
    // every remaining query base adds at most `ma`: once even that cannot reach the budget the end is invalid
    if ((int64_t)rowmax + (int64_t)P.ma * (n - i) < (int64_t)it.budget) { hopeless = true; break; }
    { const int x = i + W; uint32_t tb = 0;
      if (x < it.tl) { const int64_t gp = it.tstart + (int64_t)it.tdir * x; const int64_t wi = gp >> 5; if (wi != t_wi) { t_wi = wi; t_w = S.refseq[wi]; }
        tb = (uint32_t)(t_w >> ((gp & 31) * 2)) & 3u; }
      twin = (twin >> 2) | ((uint64_t)tb << (2 * (BW - 1))); }
  }
  sq_cand_dev* c = &cands[it.cand];
  if (hopeless) { if (it.end == 0) c->lfail = 1; else c->rfail = 1; return; }
  int32_t res = SQ_NEG_INF;
  if (it.mode == 0) {
    if (abs(n - tl) <= w) {
#pragma unroll
      for (int b = 0; b < BW; ++b) if (b == tl - n + W) res = Hp[b];
    }
  } else {
#pragma unroll
    for (int b = 0; b < BW; ++b) { int j = n + b - W; if (j >= max(0, n - w) && j <= min(tl, n + w) && Hp[b] > res) res = Hp[b]; }
  }
  if (res <= SQ_NEG_INF / 2 || res < it.budget) { if (it.end == 0) c->lfail = 1; else c->rfail = 1; }
  else atomicAdd(it.end == 0 ? &c->lscore : &c->rscore, res);
}

// [r3] The same DP with the full band (bw == SQ_MAX_BAND, the default) and NO per-cell conditions: 12 VALU operations per cell instead of 25.
//  * cells left of the matrix (j < 0) simply hold SQ_NEG_INF-ish values: column 0 then comes out of the recurrence by itself
//    (F(i,0) = max(F(i-1,0), H(i-1,0) - go) - ge = -(go + ge i); the diagonal and E parents are "minus infinity", so H(i,0) = F(i,0));
//  * cells right of the target (j > tl) are computed like any other (the window holds base 0 there): a cell's parents are (i-1,j-1),
//    (i-1,j), (i,j-1), so a cell inside the target never reads one outside, and the result picks below only look inside.  They can only
//    delay the `hopeless` exit, which is a shortcut, not a result (the final score is compared with the budget again);
//  * no clamping at SQ_NEG_INF: values drift by at most (go + ge + mp) per step, 2^29 away from real scores, and every "is this cell
//    reachable" test is `<= SQ_NEG_INF / 2`.
// Every reachable cell holds exactly the value k_dp_general computes; the tests compare both with the checker.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) k_dp(sq_map_params P, ScoreCtx S, uint32_t nitems, sq_cand_dev* __restrict__ cands,
    const uint32_t* __restrict__ cand_frag,
    uint32_t paired, const uint32_t* __restrict__ perm) {
  uint32_t ii = blockIdx.x * blockDim.x + threadIdx.x;
  if (ii >= nitems) return;
  const sq_dp_item it = S.dpq[perm[ii]];
  const uint32_t f = cand_frag[it.cand];
  const uint32_t end_id = paired ? 2 * f + it.end : f;
  ReadView r = read_view(S.rpack, S.rnmask, S.rlen, end_id, S.rw);
  const bool fw = it.rc == 0;
  const int n = it.n, tl = it.tl, go = P.go, ge = P.ge;
  constexpr int W = SQ_MAX_BAND, BW = 2 * SQ_MAX_BAND + 1;
  int32_t Hp[BW + 1], Fp[BW + 1];
#pragma unroll
  for (int b = 0; b <= BW; ++b) { Hp[b] = SQ_NEG_INF; Fp[b] = SQ_NEG_INF; }
#pragma unroll
  for (int b = 0; b < BW; ++b) { int j = b - W; if (j >= 0 && j <= tl) Hp[b] = (j == 0) ? 0 : -(go + ge * j); }
  uint64_t twin = 0;
#pragma unroll
  for (int b = 0; b < BW; ++b) twin |= (uint64_t)dp_tbase(S.refseq, it, b - W) << (2 * b);
  bool hopeless = false;
  int q_wi = -1; uint64_t q_w = 0, q_n = 0; int64_t t_wi = -1; uint64_t t_w = 0;
  for (int i = 1; i <= n; ++i) {
    uint32_t qb;
    { const int x = it.qstart + it.qdir * (i - 1); const int idx = fw ? x : r.L - 1 - x; const int wi = idx >> 5;
      if (wi != q_wi) { q_wi = wi; q_w = r.w[wi]; q_n = r.nm[idx >> 6]; }
      const uint32_t b = ((q_n >> (idx & 63)) & 1) ? 4u : ((uint32_t)(q_w >> ((idx & 31) * 2)) & 3u);
      qb = fw ? b : (b > 3 ? 4u : 3u - b); }
    const uint32_t tlo = (uint32_t)twin, thi = (uint32_t)(twin >> 32);
    int32_t left_g = SQ_NEG_INF, left_e = SQ_NEG_INF, rowmax = SQ_NEG_INF;   // left_g = H of the cell to the left, minus go
#pragma unroll
    for (int b = 0; b < BW; ++b) {
      const int32_t ev = max(left_e, left_g) - ge;
      const int32_t fv = max(Fp[b + 1], Hp[b + 1] - go) - ge;
      const uint32_t tb = b < 16 ? ((tlo >> (2 * b)) & 3u) : ((thi >> (2 * b - 32)) & 3u);
      const int32_t sc = (qb == tb) ? P.ma : P.mp;   // qb == 4 (N) never equals a target base
      const int32_t hv = max(Hp[b] + sc, max(ev, fv));
      Hp[b] = hv; Fp[b] = fv; left_g = hv - go; left_e = ev; rowmax = max(rowmax, hv);
    }
    if ((int64_t)rowmax + (int64_t)P.ma * (n - i) < (int64_t)it.budget) { hopeless = true; break; }
    { const int x = i + W; uint32_t tb = 0;
      if (x < it.tl) { const int64_t gp = it.tstart + (int64_t)it.tdir * x; const int64_t wi = gp >> 5; if (wi != t_wi) { t_wi = wi; t_w = S.refseq[wi]; }
        tb = (uint32_t)(t_w >> ((gp & 31) * 2)) & 3u; }
      twin = (twin >> 2) | ((uint64_t)tb << (2 * (BW - 1))); }
  }
  sq_cand_dev* c = &cands[it.cand];
  if (hopeless) { if (it.end == 0) c->lfail = 1; else c->rfail = 1; return; }
  int32_t res = SQ_NEG_INF;
  if (it.mode == 0) {
    if (abs(n - tl) <= W) {
#pragma unroll
      for (int b = 0; b < BW; ++b) if (b == tl - n + W) res = Hp[b];
    }
  } else {
#pragma unroll
    for (int b = 0; b < BW; ++b) { int j = n + b - W; if (j >= max(0, n - W) && j <= min(tl, n + W) && Hp[b] > res) res = Hp[b]; }
  }
  if (res <= SQ_NEG_INF / 2 || res < it.budget) { if (it.end == 0) c->lfail = 1; else c->rfail = 1; }
  else atomicAdd(it.end == 0 ? &c->lscore : &c->rscore, res);
}

// ---- a7/a8/a9 — updateRefMappings + filterAndCollectAlignments (SalmonMappingUtils.hpp:225-405) ----
__device__ inline uint8_t fmt_id(uint8_t t, uint8_t o, uint8_t s) { return (uint8_t)(t | (o << 1) | (s << 3)); }
__device__ inline uint8_t hit_type_pe(int32_t e1, bool f1, uint32_t l1, int32_t e2, bool f2, uint32_t l2) {  // SalmonUtils.cpp:577-631, canDovetail=false
  (void)l1; (void)l2;
  if (f1 != f2) { if (f1) return (e1 <= e2) ? fmt_id(1, 2, 0) : fmt_id(1, 1, 0); return (e2 <= e1) ? fmt_id(1, 2, 1) : fmt_id(1, 1, 1); }
  return f1 ? fmt_id(1, 0, 2) : fmt_id(1, 0, 3);
}

// a candidate's scores against minScoreFraction (SalmonMappingUtils.hpp:225-281): the two ends' scores as the selection sees them, whether the
// candidate stands, and its hit score (SQ_INVALID_SCORE: incompatible, skipped before alignment and not counted as filtered;
// SQ_INVALID_SCORE + 1: scored but invalid — a filtered mapping; else the score).  Shared by k_finalize and the host's candidate tap.
struct CandFinal { int32_t ls, rs, hs; bool ok; };
__host__ __device__ inline CandFinal cand_final(const sq_map_params& P, const sq_cand_dev& c, uint32_t n1, uint32_t n2) {
  CandFinal r; r.ls = SQ_INVALID_SCORE; r.rs = SQ_INVALID_SCORE; r.ok = false; r.hs = SQ_INVALID_SCORE;
  if (c.lfail == 2) return r;
  const bool hasL = c.lc != 0xFFFFFFFFu, hasR = c.rc != 0xFFFFFFFFu;
  if (hasL) {
    const int32_t minacc = (int32_t)(P.min_score_fraction * (double)(P.ma * (int32_t)n1));
    r.ls = (c.lfail || c.lscore < SQ_NEG_INF / 2 || c.lscore < minacc) ? SQ_INVALID_SCORE : c.lscore;
  }
  if (hasR) {
    const int32_t minacc = (int32_t)(P.min_score_fraction * (double)(P.ma * (int32_t)n2));
    r.rs = (c.rfail || c.rscore < SQ_NEG_INF / 2 || c.rscore < minacc) ? SQ_INVALID_SCORE : c.rscore;
  }
  r.ok = (hasL && hasR) ? (r.ls != SQ_INVALID_SCORE && r.rs != SQ_INVALID_SCORE) : ((hasL ? r.ls : r.rs) != SQ_INVALID_SCORE);
  r.hs = r.ok ? ((hasL && hasR) ? r.ls + r.rs : (hasL ? r.ls : r.rs)) : (SQ_INVALID_SCORE + 1);
  return r;
}
// thread per candidate: the hit score and the transcript as two compact arrays (SoA), so that the per-fragment selection below streams
// 8 bytes per candidate instead of 48.  [r5] The candidate records are no longer written back (0.75 GB per 5 x 10^6 pairs that only the
// debug tap read: a standing candidate's scores ARE its raw scores, and the tap applies cand_final itself).
__global__ void k_finalize(sq_map_params P, uint64_t ncand, uint32_t paired, const sq_cand_dev* __restrict__ cands,
    const uint32_t* __restrict__ cand_frag,
    const uint16_t* __restrict__ rlen,
                           int32_t* __restrict__ hs_out, uint32_t* __restrict__ tid_out) {
  uint64_t ci = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= ncand) return;
  const sq_cand_dev c = cands[ci];
  tid_out[ci] = c.tid;
  const uint32_t f = cand_frag[ci]; const uint32_t e0 = paired ? 2 * f : f;
  hs_out[ci] = cand_final(P, c, rlen[e0], paired ? rlen[e0 + 1] : 0).hs;
}

// [r5] Per-fragment selection AND the compact alignment array in one launch.  Round 4 wrote the winners into per-candidate slots, scanned the
// counts in three more launches and copied the slots together in a fifth (k_compact_alns: every alignment written twice and read once in
// between).  Here a block takes a ticket (so that blocks start in fragment order), selects, scans its 256 counts in LDS and learns where its
// alignments begin from the blocks before it by a decoupled look-back over one 64-bit descriptor per block (top two bits: 1 = this block's own
// total, 2 = the total of everything up to and including this block; a block only ever waits for blocks that hold earlier tickets, which are
// running).  The winners are remembered as (candidate, score) pairs — the first SEL_KW of a fragment in LDS, further ones as finished records
// in the old slots — and materialised once, at their final place.
#define SEL_TB 256
#define SEL_KW 8
__global__ void __launch_bounds__(SEL_TB) k_select(sq_map_params P, uint32_t nfrag, uint32_t paired, const uint64_t* __restrict__ cand_off,
    const uint32_t* __restrict__ n_cand,
    const sq_cand_dev* __restrict__ cands, const int32_t* __restrict__ hs_arr,
                         const uint32_t* __restrict__ tid_arr,
                         const uint16_t* __restrict__ rlen, const uint8_t* __restrict__ frag_flags, sq_aln* __restrict__ aln_slots,
                             uint32_t* __restrict__ n_aln,
                             uint8_t* __restrict__ map_type,
                         unsigned long long* __restrict__ stats, sq_aln* __restrict__ aln_out, uint64_t* __restrict__ aln_off,
                         unsigned long long* __restrict__ desc, uint32_t* __restrict__ ticket) {
  __shared__ uint32_t s_bid; __shared__ uint32_t s_wsum[SEL_TB / 64]; __shared__ unsigned long long s_base;
  __shared__ uint32_t s_widx[SEL_KW][SEL_TB]; __shared__ int32_t s_wsc[SEL_KW][SEL_TB];
  if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u);
  __syncthreads();
  const uint32_t bid = s_bid, tx = threadIdx.x;
  const uint32_t gid = bid * SEL_TB + tx;
  const bool act = gid < nfrag;
  const uint32_t f = act ? gid : 0;
  const uint64_t c0 = act ? cand_off[f] : 0; const uint32_t nc = act ? n_cand[f] : 0;
  const sq_cand_dev* C = cands + c0; const int32_t* HS = hs_arr + c0; const uint32_t* TID = tid_arr + c0;
  const uint32_t e0 = act ? (paired ? 2 * f : f) : 0;
  const uint32_t n1 = rlen[e0], n2 = paired ? rlen[e0 + 1] : 0;
  uint32_t nfilt = 0;
  int32_t bestDecoy = SQ_INVALID_SCORE, bestScore = SQ_INVALID_SCORE;
  auto decoy_cut = [&](int32_t bd) -> int32_t { return (int32_t)(P.decoy_threshold * (double)bd); };
  // pass 1 (updateRefMappings, SalmonMappingUtils.hpp:225-281): decoys raise the running cut-off; a
  // non-decoy hit is "recorded" only if it reaches the best decoy score seen so far (SPEC §a7)
  {
    int32_t runDecoy = SQ_INVALID_SCORE;
    for (uint32_t i = 0; i < nc; ++i) {
      const int32_t hs = HS[i];
      if (hs == SQ_INVALID_SCORE) continue;
      if (hs == SQ_INVALID_SCORE + 1) { ++nfilt; continue; }
      if (TID[i] >= P.first_decoy) { if (hs > runDecoy) runDecoy = hs; if (hs > bestDecoy) bestDecoy = hs; }
      else if (hs >= decoy_cut(runDecoy)) { if (hs > bestScore) bestScore = hs; }
    }
  }
  const bool onlyDecoy = (bestScore < decoy_cut(bestDecoy)) && (bestDecoy > SQ_INVALID_SCORE);
  uint32_t na = 0; uint8_t mt = SQ_MT_UNMAPPED; uint32_t ffilt = 0, fdecoy = 0; uint8_t first_ms = 0;
  // the alignment record of a winner: candidate `ci` of the fragment with hit score `hs`
  auto make_aln = [&](uint32_t ci, int32_t hs) -> sq_aln {
    const sq_cand_dev c = C[ci];
    const double v = (double)bestScore - (double)hs;
    sq_aln a;
    a.tid = c.tid;
    a.est_aln_prob = P.hard_filter ? -1.0 : sq_exp(-P.score_exp * v);
    a.mate_status = paired ? c.mate_status : (uint8_t)SQ_MS_SINGLE_END;
    a.frag_len = c.frag_len;
    const unsigned long long posbits = (unsigned long long)__double_as_longlong(c.cov);   // written by k_score: implied positions of the two chains
    const int32_t lpos = (int32_t)(uint32_t)posbits, rpos = (int32_t)(uint32_t)(posbits >> 32); const uint8_t lfwb = c.pad[1] & 1, rfwb = (c.pad[1] >> 1) & 1;
    if (c.mate_status == SQ_MS_PAIRED_END_PAIRED) {
      a.pos = lpos;
      a.fwd = lfwb;
      a.read_len = (uint16_t)n1;
      a.mate_pos = rpos;
      a.mate_fwd = rfwb;
      a.mate_len = (uint16_t)n2;
      a.score = c.lscore;
      a.mate_score = c.rscore;
      int32_t e1 = a.fwd ? a.pos : a.pos + (int32_t)a.read_len, e2 = a.mate_fwd ? a.mate_pos : a.mate_pos + (int32_t)a.mate_len;
      a.format_id = hit_type_pe(e1, a.fwd, a.read_len, e2, a.mate_fwd, a.mate_len);
    } else {
      const bool left = c.lc != 0xFFFFFFFFu;
      a.pos = left ? lpos : rpos; a.fwd = left ? lfwb : rfwb; a.read_len = (uint16_t)(left ? n1 : n2); a.score = left ? c.lscore : c.rscore;
      a.mate_pos = 0; a.mate_fwd = 1; a.mate_len = paired ? 0 : a.read_len; a.mate_score = 0;
      a.format_id = a.fwd ? fmt_id(0, 3, 2) : fmt_id(0, 3, 3);
    }
    return a;
  };
  if (bestScore > SQ_INVALID_SCORE && !onlyDecoy) {
    const int32_t bd = (bestDecoy == SQ_INVALID_SCORE) ? SQ_INVALID_SCORE + 1 : bestDecoy;
    const int32_t thr = P.hard_filter ? bestScore : decoy_cut(bd);
    // winners: per transcript the best recorded hit (ties -> the later compatible hit).  Candidates come
    // as one tid-sorted run (pairs / single-end) or two tid-sorted runs (left orphans, then right
    // orphans); a two-pointer merge visits every transcript once, in ascending tid, in index order.
    // "recorded" is re-derived on the fly: replaying the running decoy cut-off needs index order, so
    // the prefix maximum of decoy scores is recomputed per run position.
    uint32_t split = nc;
    // orphan-only fragment: left-anchored run, then right-anchored run
    if (nc && C[0].pad[0] != 0 && paired) {
      split = 0;
      while (split < nc && C[split].pad[0] == 1) ++split;
    }
    // running decoy maxima at the start of the second run (the first run starts from INVALID)
    int32_t runA = SQ_INVALID_SCORE, runB = SQ_INVALID_SCORE;
    for (uint32_t i = 0; i < split && split < nc; ++i) {
      const int32_t hs = HS[i];
      if (hs > SQ_INVALID_SCORE + 1 && TID[i] >= P.first_decoy && hs > runB) runB = hs;
    }
    uint32_t ia = 0, ib = split;
    while (ia < split || ib < nc) {
      const uint32_t ta = ia < split ? TID[ia] : 0xFFFFFFFFu, tb = ib < nc ? TID[ib] : 0xFFFFFFFFu;
      const uint32_t t = ta < tb ? ta : tb;
      int32_t cur = SQ_INVALID_SCORE; int curi = -1;
      while (ia < split && TID[ia] == t) { const int32_t ds = HS[ia];
        if (ds > SQ_INVALID_SCORE + 1) {
          if (t >= P.first_decoy) {
            if (ds > runA) runA = ds;
          } else if (ds >= decoy_cut(runA)) {
            if (curi < 0 || ds > cur || (ds == cur && C[ia].compat)) {
              cur = ds;
              curi = (int)ia;
            }
          }
        }
        ++ia; }
      while (ib < nc && TID[ib] == t) { const int32_t ds = HS[ib];
        if (ds > SQ_INVALID_SCORE + 1) {
          if (t >= P.first_decoy) {
            if (ds > runB) runB = ds;
          } else if (ds >= decoy_cut(runB)) {
            if (curi < 0 || ds > cur || (ds == cur && C[ib].compat)) {
              cur = ds;
              curi = (int)ib;
            }
          }
        }
        ++ib; }
      if (curi < 0 || cur < thr) continue;
      if (!P.hard_filter && sq_exp(-P.score_exp * ((double)bestScore - (double)cur)) < P.min_aln_prob) continue;
      if (na == 0) first_ms = paired ? C[curi].mate_status : (uint8_t)SQ_MS_SINGLE_END;
      if (na < SEL_KW) { s_widx[na][tx] = (uint32_t)curi; s_wsc[na][tx] = cur; }
      else aln_slots[c0 + na] = make_aln((uint32_t)curi, cur);   // a fragment with more winners than the LDS columns hold: the rest waits in its slots
      ++na;
    }
    if (na) {
      switch (first_ms) {
        case SQ_MS_PAIRED_END_PAIRED: mt = SQ_MT_PAIRED_MAPPED;
        break;
        case SQ_MS_PAIRED_END_LEFT: mt = SQ_MT_LEFT_ORPHAN;
        break;
        case SQ_MS_PAIRED_END_RIGHT: mt = SQ_MT_RIGHT_ORPHAN;
        break;
        default: mt = SQ_MT_SINGLE_MAPPED;
      }
    }
  } else if (nc) {
    mt = onlyDecoy ? SQ_MT_DECOY : SQ_MT_UNMAPPED;
    ffilt = 1; fdecoy = onlyDecoy ? 1 : 0;
  }
  if (act) { n_aln[f] = na; map_type[f] = mt; }
  // where the block's alignments begin: exclusive scan of the 256 counts, then the look-back
  const uint32_t lane = tx & 63, wave = tx >> 6;
  uint32_t incl = na;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= (uint32_t)d) incl += t; }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  uint32_t woff = 0, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < SEL_TB / 64; ++w) { if (w < wave) woff += s_wsum[w]; total += s_wsum[w]; }
  if (tx == 0) {
    const unsigned long long VM = (1ULL << 62) - 1;
    unsigned long long before = 0;
    if (bid == 0) __hip_atomic_store(&desc[0], (2ULL << 62) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else {
      __hip_atomic_store(&desc[bid], (1ULL << 62) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (uint32_t j = bid - 1;;) {
        const unsigned long long dsc = __hip_atomic_load(&desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long fl = dsc >> 62;
        if (fl == 0) { __builtin_amdgcn_s_sleep(1); continue; }
        before += dsc & VM;
        if (fl == 2) break;
        --j;
      }
      __hip_atomic_store(&desc[bid], (2ULL << 62) | (before + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_base = before;
    if ((uint64_t)(bid + 1) * SEL_TB >= nfrag) aln_off[nfrag] = before + total;   // the block of the last fragments closes the offsets
  }
  __syncthreads();
  if (act) {
    const uint64_t at = s_base + woff + (incl - na);
    aln_off[f] = at;
    sq_aln* out = aln_out + at;
    for (uint32_t j = 0; j < na; ++j) out[j] = j < SEL_KW ? make_aln(s_widx[j][tx], s_wsc[j][tx]) : aln_slots[c0 + j];
  }
  // seven counters: summed per block in LDS, one atomic per counter and block (see block_stat_add)
  __shared__ unsigned long long s_st[7];
  if (threadIdx.x < 7) s_st[threadIdx.x] = 0;
  __syncthreads();
  block_stat_add(&s_st[0], ffilt); block_stat_add(&s_st[1], fdecoy);
  block_stat_add(&s_st[2], nfilt);
  block_stat_add(&s_st[3], na);
  block_stat_add(&s_st[4], na ? 1 : 0);
  block_stat_add(&s_st[5], nc ? 1 : 0);
  block_stat_add(&s_st[6], (act && !nc && (frag_flags[f] & 1)) ? 1 : 0);
  __syncthreads();
  if (threadIdx.x < 7 && s_st[threadIdx.x]) {
    const int which = threadIdx.x == 0 ? ST_FRAGFILT : threadIdx.x == 1 ? ST_DECOY : threadIdx.x == 2 ? ST_MAPFILT : threadIdx.x == 3 ? ST_ALNS
                      : threadIdx.x == 4 ? ST_MAPPED : threadIdx.x == 5 ? ST_JOINT : ST_DOVETAIL;
    atomicAdd(&stats[which], s_st[threadIdx.x]);
  }
}

// [r6] a grid of at most 1 024 blocks: with a block per 256 fragments the kernel was its own 4 x 10^4 same-line atomics (0.24 ms per 5 x 10^6 pairs)
__global__ void k_count_kmer_frags(uint32_t nfrag, uint32_t paired, const uint32_t* __restrict__ n_chains,
    unsigned long long* __restrict__ stats) {
  unsigned long long any = 0, nch = 0;
  for (uint32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < nfrag; f += gridDim.x * blockDim.x) {
    const uint32_t v = paired ? (n_chains[2 * f] + n_chains[2 * f + 1]) : n_chains[f];
    nch += v; any += v != 0;
  }
  __shared__ unsigned long long s_st[2];
  if (threadIdx.x < 2) s_st[threadIdx.x] = 0;
  __syncthreads();
  block_stat_add(&s_st[0], any); block_stat_add(&s_st[1], nch);
  __syncthreads();
  if (threadIdx.x < 2 && s_st[threadIdx.x]) atomicAdd(&stats[threadIdx.x == 0 ? ST_KMER : ST_CHAINS], s_st[threadIdx.x]);
}

}  // namespace sqk
