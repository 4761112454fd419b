// hip/mem_kernels.h — row a2 in ONE kernel per size class: projection of a read end's uni-MEMs through the contig
// table (fillMemCollection), the per-end sort by (transcript, reference position) and the chaining DP (findOptChain)
// all happen in LDS; HBM sees the uni-MEM records and contig-table runs once on the way in, and the sorted MEM
// records + chains once on the way out.  (Round 1 ran k_project -> a global 61-bit radix sort -> k_chain: the MEM
// records crossed HBM about ten times, and the thread-per-end projection stored them 8 bytes at a time.)
//
// A read end with n projected MEMs is handled by a group of G lanes:
//   n <= 16    G = 16  (four ends per wave; the common case: ~13 MEMs on ~7 transcripts)        k_mems<16, 16, 256>   16 KB of LDS per block
//   n <= 32    G = 16                                                                           k_mems<16, 32, 256>   18 KB
//   n <= 64    G = 16                                                                           k_mems<16, 64, 256>   32 KB
//   n <= 256   G = 64  (one wave per end, four MEMs per lane) [r3: the rank sort of a wave costs n x CAP / 64 compares whatever n is]   k_mems<64, 256, 128>
//   larger     the flat passes over a compacted list (k_project_list -> radix sort -> k_lg_*).  [r6] A class of 257 .. 1024 MEMs (a wave per end, sixteen MEMs per lane,
//              k_mems<64, 1024, 128>) existed until round 6: its chaining is a lane per transcript group, and an end of that size on a decoy chromosome is ONE group — on
//              configs[3] it took 7.4 ms per 4 x 10^6 pairs where the flat passes take 5.0 for the same ends (profiles/r06_large_end_threshold.txt)
// Lanes expand one occurrence each (coalesced contig-table loads, all gathers in flight), rank-sort the keys held
// in LDS (stable: ties keep emission order, SPEC §a2), then every lane runs the chaining DP of whole transcripts.
// Same arithmetic, same order of operations as the checker: results are bit-identical whatever the class.
#pragma once
#include "map_kernels.h"

namespace sqk {

#define MK_X_CAP 16
#define MK_T_CAP 32
#define MK_S_CAP 64
#define MK_L_CAP 256

__device__ inline void mk_wsync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }

// ends by size class; an end without MEMs has no chains.  Classes (MEMs per end): 0: <= 8 (eight lanes per end: k_mems<8, 8, 256>), 1: <= 16, 2: <= 32, 3: <= 64, 4: <= 256, 5: unused
// since round 6 (its list stays empty), 6: larger.  ctr[c] = ends in class c, ctr[7] = MEMs of the large ends.  Classes 0 and 1 share a kernel instantiation (as 2, 3 and 4
// have theirs) but are launched from their own lists: the four ends of a wave then cost about the same, and a wave waits for its
// slowest end.  Blocks of 1024: the per-class counts of the 16 waves are summed in LDS, so a block costs at most seven cursor atomics.
#define MK_NCLS 7
__device__ inline int mk_class(uint32_t n) { return n <= 8 ? 0 : (n <= MK_X_CAP ? 1 : (n <= MK_T_CAP ? 2 : (n <= MK_S_CAP ? 3 : (n <= MK_L_CAP ? 4 : 6)))); }
// A list entry carries what k_mems needs to start on the end — slab offset, MEM and uni-MEM counts, read length — so that its first load
// is its only load before the uni-MEM records: x = end, y = MEMs | uni-MEMs << 16 | length << 22, (z, w) = slab offset.
__global__ void __launch_bounds__(1024) k_mem_classes(uint32_t nends, const uint32_t* __restrict__ n_proj, const uint64_t* __restrict__ mem_off,
                              const uint32_t* __restrict__ n_uni, const uint16_t* __restrict__ rlen, uint32_t* __restrict__ lists /* [MK_NCLS][nends] */,
                              uint4* __restrict__ linfo /* [MK_NCLS - 1][nends] */, uint32_t* __restrict__ lbase, uint32_t* __restrict__ n_chains,
                              uint32_t* __restrict__ ctr) {
  __shared__ uint32_t s_cnt[MK_NCLS][16]; __shared__ uint32_t s_base[MK_NCLS];
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
  const uint32_t n = e < nends ? n_proj[e] : 0;
  // [r4] an end with more uni-MEMs than a size class keeps headers for (rare: a read across many short unitigs) takes the large-end path, whatever its MEM count
  const int cls = (e >= nends || n == 0) ? -1 : (n_uni[e] > SQ_MAX_UNIMEMS ? MK_NCLS - 1 : mk_class(n));
  if (e < nends && n == 0) n_chains[e] = 0;
  unsigned long long m[MK_NCLS];
#pragma unroll
  for (int c = 0; c < MK_NCLS; ++c) { m[c] = __ballot(cls == c); if (lane == 0) s_cnt[c][wv] = (uint32_t)__popcll(m[c]); }
  __syncthreads();
  if (threadIdx.x < MK_NCLS) {
    uint32_t tot = 0;
    for (int w = 0; w < 16; ++w) { const uint32_t v = s_cnt[threadIdx.x][w]; s_cnt[threadIdx.x][w] = tot; tot += v; }
    s_base[threadIdx.x] = tot ? atomicAdd(&ctr[threadIdx.x], tot) : 0u;
  }
  __syncthreads();
  if (cls >= 0) {
    unsigned long long mine = 0;
#pragma unroll
    for (int c = 0; c < MK_NCLS; ++c) if (c == cls) mine = m[c];
    const uint32_t pos = s_base[cls] + s_cnt[cls][wv] + (uint32_t)__popcll(mine & ((1ULL << lane) - 1));
    lists[(size_t)cls * nends + pos] = e;
    if (cls == MK_NCLS - 1) lbase[pos] = atomicAdd(&ctr[MK_NCLS], n);
    else { const uint64_t mo = mem_off[e]; linfo[(size_t)cls * nends + pos] = make_uint4(e, n | (n_uni[e] << 16) | ((uint32_t)rlen[e] << 22), (uint32_t)mo, (uint32_t)(mo >> 32)); }
  }
}

struct MkHdr { uint64_t a; uint32_t cnt, ulen, ustart; uint16_t qpos, lenfw; };   // one uni-MEM of the end: contig-table run + what the projection needs

template <int G, int CAP, int TBK>
__global__ void __launch_bounds__(TBK) k_mems(const uint64_t* __restrict__ uoff, const uint64_t* __restrict__ ctab_off, const uint64_t* __restrict__ ctab,
                                              const uint64_t* __restrict__ ref_accum, sq_map_params P, const double* __restrict__ gapcost,
                                              const uint4* __restrict__ list, uint32_t nlist,
                                              const sq_unimem_dev* __restrict__ um, uint32_t us /* uni-MEM slots per end in `um` */,
                                              uint64_t* __restrict__ mkey, uint64_t* __restrict__ mval, uint32_t* __restrict__ mnext,
                                              sq_chain_dev* __restrict__ chains, uint32_t* __restrict__ n_chains) {
  constexpr int GPB = TBK / G, E = CAP / G;
  static_assert(G == 8 || G == 16 || G == 32 || G == 64, "group = a power-of-two slice of a wave");
  constexpr int CP8 = CAP + 1, CP4 = CAP + 1, CP2 = CAP + 2, CP1 = CAP + 4;   // padded rows: the groups of a wave must not sit on the same banks
  __shared__ uint64_t s_key[GPB][CP8];                 // ranking keys, then (same bytes) the DP scores f[] as doubles
  // The uni-MEM headers are dead once the projection is done: the sorted MEM columns and the DP's arrays take their bytes (LDS per
  // block decides how many blocks a CU holds: 29 -> 17 KB for the tiny class).
  constexpr int O_R = 0, O_TID = O_R + 4 * CP4, O_Q = O_TID + 4 * CP4, O_LF = O_Q + 2 * CP2, O_P = O_LF + 2 * CP2, O_ACC = O_P + 2 * CP2,
                O_GS = O_ACC + 2 * CP2, O_GC = O_GS + 2 * (CP2 + 2), O_FL = O_GC + 2 * CP2, O_END = O_FL + CP1;
  // [r3] only uni-MEMs with occurrences are kept (those over max_occ project nothing), so an end of n MEMs has at most n headers
  constexpr int HB = (CAP < (int)SQ_MAX_UNIMEMS ? CAP : (int)SQ_MAX_UNIMEMS) * (int)sizeof(MkHdr);
  constexpr int UW = ((O_END > HB ? O_END : HB) + 7) / 8 + 1;   // 8-byte words per group; + 1: the groups of a wave must not sit on the same banks
  __shared__ uint64_t s_u[GPB][UW];
  __shared__ double s_gap[SQ_MAX_CHAIN_GAP + 1];
  char* const ub = reinterpret_cast<char*>(s_u[threadIdx.x / G]);
  MkHdr* const g_h = reinterpret_cast<MkHdr*>(ub);
  int32_t* const g_r = reinterpret_cast<int32_t*>(ub + O_R); uint32_t* const g_tid = reinterpret_cast<uint32_t*>(ub + O_TID);
  int16_t* const g_q = reinterpret_cast<int16_t*>(ub + O_Q); uint16_t* const g_lf = reinterpret_cast<uint16_t*>(ub + O_LF);      // len | fw << 15
  int16_t* const g_p = reinterpret_cast<int16_t*>(ub + O_P); uint16_t* const g_acc = reinterpret_cast<uint16_t*>(ub + O_ACC);
  uint16_t* const g_gs = reinterpret_cast<uint16_t*>(ub + O_GS); uint16_t* const g_gc = reinterpret_cast<uint16_t*>(ub + O_GC);
  uint8_t* const g_fl = reinterpret_cast<uint8_t*>(ub + O_FL);
  const int tx = (int)threadIdx.x, gl = tx % G, gi = tx / G, lane = tx & 63, gsh = lane & ~(G - 1);
  for (int i = tx; i <= SQ_MAX_CHAIN_GAP; i += TBK) s_gap[i] = gapcost[i];
  __syncthreads();
  const uint32_t li = blockIdx.x * GPB + (uint32_t)gi;
  const bool act = li < nlist;
  uint4 li4 = make_uint4(0, 0, 0, 0); if (act) li4 = list[li];
  const uint32_t e = li4.x;
  const uint64_t base = ((uint64_t)li4.w << 32) | li4.z;
  const uint32_t n = li4.y & 0xFFFFu;
  const uint32_t nu = (li4.y >> 16) & 0x3Fu;
  const int L = (int)(li4.y >> 22);
  // ---- uni-MEM headers: lane i fetches uni-MEM i; its contig-table run and unitig length came with it from k_seed ----
  { uint32_t kept = 0;
    for (uint32_t i0 = 0; i0 < nu; i0 += G) {   // the lanes of a group run this loop together: the ballot below sees all of them
      const uint32_t i = i0 + (uint32_t)gl; const bool in = i < nu;
      sq_unimem_dev m{}; if (in) m = um[(size_t)e * us + i];
      const bool has = in && m.cnt != 0;
      const unsigned long long bm = __ballot(has);
      const uint32_t gm = (G == 64) ? 0u : (uint32_t)((bm >> gsh) & ((1ull << (G & 63)) - 1));
      const uint32_t below = (G == 64) ? (uint32_t)__popcll(bm & ((1ull << lane) - 1)) : (uint32_t)__popc(gm & ((1u << gl) - 1));
      if (has) { MkHdr h; h.a = m.ctab_a; h.cnt = m.cnt; h.ulen = m.ulen; h.ustart = m.ustart; h.qpos = m.qpos; h.lenfw = (uint16_t)(m.len | (m.fw ? 0x8000u : 0u)); g_h[kept + below] = h; }
      kept += (G == 64) ? (uint32_t)__popcll(bm) : (uint32_t)__popc(gm);
    } }
  mk_wsync();
  // ---- projection: output slot p = occurrence (uni-MEM i, j - a_i) in emission order; one lane per occurrence ----
  uint64_t K[E]; int32_t R[E]; uint32_t T[E]; int16_t Q[E]; uint16_t LF[E]; uint32_t rk[E];
  {
    uint32_t hi = 0, hacc = 0;
#pragma unroll
    for (int t = 0; t < E; ++t) {
      const uint32_t p = (uint32_t)gl + (uint32_t)(G * t);
      K[t] = 0; R[t] = 0; T[t] = 0; Q[t] = 0; LF[t] = 0; rk[t] = 0;
      if (p < n) {
        while (p >= hacc + g_h[hi].cnt) { hacc += g_h[hi].cnt; ++hi; }
        const MkHdr h = g_h[hi];
        const uint64_t o = ctab[h.a + (p - hacc)];
        const uint32_t tid = (uint32_t)(o >> 32); const bool ufw = (o >> 31) & 1; const int upos = (int)(o & 0x7FFFFFFF);
        const int mlen = (int)(h.lenfw & 0x7FFFu); const bool mfw = (h.lenfw & 0x8000u) != 0;
        const int rpos = ufw ? upos + (int)h.ustart : upos + ((int)h.ulen - ((int)h.ustart + mlen));
        const bool fw = (ufw == mfw);
        const uint32_t q = fw ? (uint32_t)h.qpos : (uint32_t)(L - ((int)h.qpos + mlen));
        K[t] = ref_accum[tid] + (uint64_t)rpos; R[t] = rpos; T[t] = tid; Q[t] = (int16_t)q; LF[t] = (uint16_t)(mlen | (fw ? 0x8000 : 0));
        s_key[gi][p] = K[t];
      }
    }
  }
  mk_wsync();
  // ---- stable sort: rank = number of records that come before mine in (key, emission index) order ----
  if constexpr (G == 64) {
    // [r3] a wave per end (65 .. 1024 MEMs: repeat families, a decoy genome): counting the records before mine costs n x CAP / 64 compares per
    // lane — 13 of the 18.7 ms per 4 x 10^6 pairs on the configs[3] index went there.  A bitonic network over (key << 10 | emission index) in LDS
    // sorts the same total order in log^2 steps; the ranks are read back through the index bits.
    uint32_t N = 64; while (N < n) N <<= 1;
    uint64_t* const sk = s_key[gi];
    for (uint32_t p = (uint32_t)gl; p < N; p += 64) sk[p] = p < n ? ((sk[p] << 10) | (uint64_t)p) : ~0ull;   // every lane rewrites the keys it stored
    mk_wsync();
    for (uint32_t k2 = 2; k2 <= N; k2 <<= 1)
      for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
        for (uint32_t x = (uint32_t)gl; x < N / 2; x += 64) {
          const uint32_t i = ((x & ~(j - 1)) << 1) | (x & (j - 1)), l = i | j;
          const uint64_t a = sk[i], b = sk[l];
          if ((a > b) == ((i & k2) == 0)) { sk[i] = b; sk[l] = a; }
        }
        mk_wsync();
      }
    uint16_t* const rank_of = reinterpret_cast<uint16_t*>(g_p);   // the DP's predecessor column is not in use yet
    for (uint32_t r = (uint32_t)gl; r < n; r += 64) rank_of[(uint32_t)(sk[r] & 1023u)] = (uint16_t)r;
    mk_wsync();
#pragma unroll
    for (int t = 0; t < E; ++t) { const uint32_t p = (uint32_t)gl + (uint32_t)(G * t); if (p < n) rk[t] = rank_of[p]; }
  } else {
    for (uint32_t q2 = 0; q2 < n; ++q2) {
      const uint64_t kq = s_key[gi][q2];
#pragma unroll
      for (int t = 0; t < E; ++t) {
        const uint32_t p = (uint32_t)gl + (uint32_t)(G * t);
        rk[t] += ((kq < K[t]) | ((kq == K[t]) & (q2 < p))) ? 1u : 0u;
      }
    }
  }
  mk_wsync();
#pragma unroll
  for (int t = 0; t < E; ++t) {
    const uint32_t p = (uint32_t)gl + (uint32_t)(G * t);
    if (p < n) {
      const uint32_t r = rk[t];
      g_r[r] = R[t]; g_tid[r] = T[t]; g_q[r] = Q[t]; g_lf[r] = LF[t]; g_fl[r] = 0;
      mkey[base + r] = ((uint64_t)e << 40) | K[t];
      mval[base + r] = mem_pack_val(T[t], (uint32_t)(uint16_t)Q[t], (uint32_t)(LF[t] & 0x7FFFu), (LF[t] >> 15) & 1u);
    }
  }
  mk_wsync();
  // ---- transcript groups: starts of the runs of equal transcript id ----
  uint32_t ng = 0;
#pragma unroll
  for (int t = 0; t < E; ++t) {
    const uint32_t p = (uint32_t)gl + (uint32_t)(G * t);
    const bool st = p < n && (p == 0 || g_tid[p] != g_tid[p - 1]);
    const unsigned long long bm = __ballot(st);
    const uint32_t gm = (G == 64) ? 0u : (uint32_t)((bm >> gsh) & ((1ull << (G & 63)) - 1));
    const uint32_t below = (G == 64) ? (uint32_t)__popcll(bm & ((1ull << lane) - 1)) : (uint32_t)__popc(gm & ((1u << gl) - 1));
    const uint32_t tot = (G == 64) ? (uint32_t)__popcll(bm) : (uint32_t)__popc(gm);
    if (st) g_gs[ng + below] = (uint16_t)p;
    ng += tot;
  }
  if (gl == 0) g_gs[ng] = (uint16_t)n;
  mk_wsync();
  // ---- chaining DP, one lane per transcript (SPEC §a2; same operations in the same order as the checker) ----
  double* f = (double*)s_key[gi];
  double lbest = 0.0;
  for (uint32_t k = (uint32_t)gl; k < ng; k += G) {
    const int g0 = (int)g_gs[k], g1 = (int)g_gs[k + 1];
    double best = 0.0;
    for (int i = g0; i < g1; ++i) {
      const int qi = g_q[i], ri = g_r[i], len_i = (int)(g_lf[i] & 0x7FFFu); const uint32_t fwi = g_lf[i] >> 15;
      double fi = (double)len_i; int pi = -1; int rounds = 2;
      for (int j = i - 1; j >= g0; --j) {
        const int rd = ri - g_r[j];
        if (rd > SQ_MAX_CHAIN_GAP) break;   // [r3] the MEMs of a transcript are sorted by reference position: every earlier one is farther still.  Exactly what
                                            // skipping them one by one gives, without the quadratic walk over the copies of a repeat family on a chromosome
        if ((uint32_t)(g_lf[j] >> 15) != fwi) continue;
        const int qd = qi - (int)g_q[j];
        if (qd < 0 || max(qd, rd) > SQ_MAX_CHAIN_GAP) continue;
        const int l = abs(qd - rd);
        const double a = (double)min(len_i, min(qd, rd));
        const double sc = f[j] + a - s_gap[l];
        if (sc > fi) { fi = sc; pi = j; }
        if (!P.no_heuristic && pi >= 0) { if (--rounds <= 0) break; }
      }
      f[i] = fi; g_p[i] = (int16_t)pi;
      if (fi > best) best = fi;
    }
    const double thr = P.pre_thr * best;
    uint32_t nacc = 0;
    for (;;) {   // accept chain ends by (score desc, index asc); s_fl: bit 0 used by an accepted chain, bit 1 tried and dropped
      int bi = -1; double bf = 0.0;
      for (int i = g0; i < g1; ++i) {
        if (g_fl[i]) continue;
        const double fv = f[i];
        if (fv >= thr && (bi < 0 || fv > bf)) { bi = i; bf = fv; }
      }
      if (bi < 0) break;
      bool clash = false;
      for (int x = bi; x >= 0; x = g_p[x]) if (g_fl[x] & 1) { clash = true; break; }
      if (clash) { g_fl[bi] |= 2; continue; }
      for (int x = bi; x >= 0; x = g_p[x]) g_fl[x] |= 1;
      g_acc[g0 + (int)nacc] = (uint16_t)bi; ++nacc;
      if (bf > lbest) lbest = bf;
    }
    g_gc[k] = (uint16_t)nacc;
  }
  // ---- hitFilterPolicy AFTER + consensus fraction over the end's chains; chains go out in (transcript, acceptance) order ----
  double bestAll = lbest;
  uint32_t ngmax = ng;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    if (s < G) { const double o = __shfl_xor(bestAll, s, 64); if (o > bestAll) bestAll = o; }
    const uint32_t og = (uint32_t)__shfl_xor((int)ngmax, s, 64); if (og > ngmax) ngmax = og;
  }
  const double cthr = P.consensus_frac * bestAll;
  uint32_t carry = 0;
  for (uint32_t t0 = 0; t0 < ngmax; t0 += G) {
    const uint32_t k = t0 + (uint32_t)gl;
    uint32_t kept = 0; int g0 = 0, g1 = 0; uint32_t nacc = 0;
    if (k < ng) {
      g0 = (int)g_gs[k]; g1 = (int)g_gs[k + 1]; nacc = g_gc[k];
      for (uint32_t c = 0; c < nacc; ++c) if (f[g_acc[g0 + (int)c]] >= cthr) ++kept;
    }
    uint32_t incl = kept;
#pragma unroll
    for (int s = 1; s < G; s <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, s, G); if (gl >= s) incl += o; }
    const uint32_t tot = (uint32_t)__shfl((int)incl, G - 1, G);
    uint32_t w = carry + incl - kept;
    if (k < ng && kept) {
      const int gn = g1 - g0; const bool by_mask = gn <= 32;
      for (uint32_t c = 0; c < nacc; ++c) {
        const int bi = (int)g_acc[g0 + (int)c];
        const double bf = f[bi];
        if (bf < cthr) continue;
        uint32_t mask = 0, cnt = 0; int first = bi;
        for (int x = bi; x >= 0; x = g_p[x]) {
          ++cnt; first = x;
          if (by_mask) mask |= 1u << (x - g0);
          else { const int pr = g_p[x]; if (pr >= 0) mnext[base + (uint32_t)pr] = (uint32_t)x; }
        }
        sq_chain_dev ch;
        ch.score = bf; ch.tid = g_tid[bi]; ch.pos = g_r[first] - (int32_t)g_q[first];
        ch.last_end = g_r[bi] + (int32_t)(g_lf[bi] & 0x7FFFu);
        ch.first = by_mask ? (uint32_t)g0 : (uint32_t)first; ch.n_mems = (uint16_t)cnt; ch.read_len = (uint16_t)L;
        ch.fw = (uint8_t)(g_lf[bi] >> 15); ch.pad[0] = by_mask ? 1 : 0; ch.pad[1] = ch.pad[2] = 0; ch.pad2 = mask;
        chains[base + w] = ch; ++w;
      }
    }
    carry += tot;
  }
  if (act && gl == 0) n_chains[e] = carry;
  // no counters here: four ends per wave would mean a same-address atomic per wave (~10 ns each, 5 x 10^5 waves per 10^6 pairs);
  // the MEM total is the scan's last element and the chains are counted per fragment in k_count_kmer_frags
}

// ---- the large ends (more than MK_L_CAP MEMs): projection into a compact buffer, library radix sort, scatter back ----
__global__ void k_project_list(sq_dict_view d, const uint64_t* __restrict__ ctab_off, const uint64_t* __restrict__ ctab, const uint64_t* __restrict__ ref_accum,
                               sq_map_params P, const uint32_t* __restrict__ list, const uint32_t* __restrict__ lbase, uint32_t nlist,
                               const uint16_t* __restrict__ rlen, const sq_unimem_dev* __restrict__ um, uint32_t us, const uint32_t* __restrict__ n_uni,
                               uint64_t* __restrict__ ckey, uint64_t* __restrict__ cval) {
  // one wave per end, a lane per occurrence of the current uni-MEM (runs of up to maxOccsPerHit = 1000 entries)
  const uint32_t wi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; const uint32_t lane = threadIdx.x & 63;
  if (wi >= nlist) return;
  const uint32_t e = list[wi]; uint64_t w = lbase[wi]; const int L = rlen[e];
  const sq_unimem_dev* in = um + (size_t)e * us;
  for (uint32_t i = 0; i < n_uni[e]; ++i) {
    const sq_unimem_dev m = in[i];
    const uint64_t a = ctab_off[m.unitig], b = ctab_off[m.unitig + 1];
    if (b - a > P.max_occ) continue;
    const int ulen = (int)(d.uoff[m.unitig + 1] - d.uoff[m.unitig]);
    for (uint64_t j = a + lane; j < b; j += 64) {
      const uint64_t o = ctab[j]; const uint32_t tid = (uint32_t)(o >> 32); const bool ufw = (o >> 31) & 1; const int upos = (int)(o & 0x7FFFFFFF);
      const int rpos = ufw ? upos + (int)m.ustart : upos + (ulen - ((int)m.ustart + (int)m.len));
      const bool fw = (ufw == (m.fw != 0));
      const uint32_t q = fw ? m.qpos : (uint32_t)(L - ((int)m.qpos + (int)m.len));
      ckey[w + (j - a)] = ((uint64_t)e << 40) | (ref_accum[tid] + (uint64_t)rpos);
      cval[w + (j - a)] = mem_pack_val(tid, q, m.len, fw);
    }
    w += b - a;
  }
}
// sorted compact records -> the ends' slabs: record i of end e lands at mem_off[e] + (i - first record of e).  [r6] The index of the end's first record is the high half of
// the flat passes' running maximum (`se`, below) — the scatter runs behind that scan instead of finding it by a binary search over the sorted keys per record (25 dependent
// loads each: 2.1 ms per 4 x 10^6 pairs on configs[3])
__global__ void k_scatter_sorted(uint64_t total, const uint64_t* __restrict__ skey, const uint64_t* __restrict__ sval, const uint64_t* __restrict__ se, const uint64_t* __restrict__ mem_off,
                                 uint64_t* __restrict__ mkey, uint64_t* __restrict__ mval) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint64_t key = skey[i]; const uint64_t e = key >> 40;
  const uint64_t dst = mem_off[e] + (i - (se[i] >> 32));
  mkey[dst] = key; mval[dst] = sval[i];
}

// ---- [r4] chaining of the large ends as flat passes over the sorted compact records (all large ends of a batch at once) ----
// The thread-per-end walk (k_chain) took 40 ms per 4 x 10^6 pairs on the 3.1 Gnt decoy index: an end in a genomic repeat family has
// thousands of MEMs on one decoy chromosome, and one lane went through them one by one (and through the accepted-chain search once per chain).
// What makes it parallel: the MEMs of a (end, transcript) group are sorted by reference position, and a MEM can only chain onto MEMs at most
// SQ_MAX_CHAIN_GAP before it — so a gap of more than that between neighbours splits the group into CLUSTERS that no chain crosses.  The DP and
// the clash rule of the acceptance run per cluster (a handful of MEMs); the group's threshold and the end's best score are maxima (atomics);
// the order the reference's loop accepts chains in — by score, ties by index — is restored by two stable radix sorts at the end.
// Same operations on the same values as k_chain / SPEC §a2, so the chains are bit for bit the same.
#define LG_CL 1u   // first MEM of a cluster
#define LG_GR 2u   // first MEM of an (end, transcript) group
#define LG_EN 4u   // first MEM of an end
struct LgMaxPair { __host__ __device__ __forceinline__ uint64_t operator()(uint64_t a, uint64_t b) const {
  const uint32_t ah = (uint32_t)(a >> 32), bh = (uint32_t)(b >> 32), al = (uint32_t)a, bl = (uint32_t)b;
  return ((uint64_t)(ah > bh ? ah : bh) << 32) | (uint64_t)(al > bl ? al : bl); } };
// start / end of large end i's run in the compact projection (the runs tile the buffer in the order the cursor handed them out)
struct LgSegBound { const uint32_t* lbase; const uint32_t* list; const uint32_t* n_proj; uint32_t end;
  __host__ __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return lbase[i] + (end ? n_proj[list[i]] : 0u); } };
// flags per record + the input of the running-maximum scan (hi: index of the end's first record, lo: of the group's); group / end maxima start at 0
__global__ void k_lg_flags(uint32_t total, const uint64_t* __restrict__ skey, const uint64_t* __restrict__ sval, uint8_t* __restrict__ flags,
                           uint64_t* __restrict__ se_in, uint64_t* __restrict__ gbest, uint64_t* __restrict__ ebest, uint32_t* __restrict__ n_chains) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= total) return;
  const uint64_t k = skey[i]; const uint32_t tid = (uint32_t)(sval[i] >> 32);
  uint32_t f = 0;
  if (i == 0) f = LG_CL | LG_GR | LG_EN;
  else {
    const uint64_t kp = skey[i - 1]; const uint32_t tp = (uint32_t)(sval[i - 1] >> 32);
    if ((kp >> 40) != (k >> 40)) f = LG_CL | LG_GR | LG_EN;
    else if (tp != tid) f = LG_CL | LG_GR;
    else if ((k & ((1ull << 40) - 1)) - (kp & ((1ull << 40) - 1)) > (uint64_t)SQ_MAX_CHAIN_GAP) f = LG_CL;
  }
  flags[i] = (uint8_t)f;
  se_in[i] = ((uint64_t)((f & LG_EN) ? i : 0u) << 32) | (uint64_t)((f & LG_GR) ? i : 0u);
  if (f & LG_GR) gbest[i] = 0;
  if (f & LG_EN) { ebest[i] = 0; n_chains[(uint32_t)(k >> 40)] = 0; }
}
struct LgMem { int32_t r, q, len; uint32_t fw; };
__device__ __forceinline__ LgMem lg_mem(const uint64_t* __restrict__ skey, const uint64_t* __restrict__ sval, uint32_t i, uint64_t r0) {
  const uint64_t v = sval[i]; LgMem m; m.r = (int32_t)((skey[i] & ((1ull << 40) - 1)) - r0); m.q = (int32_t)((v >> 10) & 1023); m.len = (int32_t)(v & 1023); m.fw = (uint32_t)((v >> 20) & 1); return m;
}
// ---- [r6] the DP and the acceptance over TILES of the sorted records, in LDS ----
// A thread per cluster walking global memory (round 4's k_lg_dp / k_lg_accept) pays a trip to memory per MEM it looks at, and the 64 clusters of a wave lie ~70 bytes apart: every
// load is 64 lines.  On configs[3] a batch has 77 M such records in 8.9 M clusters (8.6 MEMs on average, 120 at most; 1.7 looks back per MEM): 6.2 + 2.2 ms.  Here a block owns the
// clusters that START in its tile of LG_T records: the tile and LG_O records behind it are loaded once, coalesced, the cluster starts are compacted from the flags, a thread takes a
// cluster and works in LDS (a record = one 16-byte word: position and packed MEM in 61 bits, f beside them), results leave coalesced.  A cluster that runs past the loaded records
// (none does on configs[3]) continues in global memory: same operations in the same order on the same values either way.
#define LG_T 1024u
#define LG_O 128u
#define LG_N (LG_T + LG_O)
#define LG_TB 128u
#define LG_POS_MASK ((1ull << 40) - 1)
struct LgRec { uint64_t pv; double f; };   // pv = position << 21 | fw << 20 | q << 10 | len
__device__ __forceinline__ uint64_t lg_pv(uint64_t key, uint64_t val) { return ((key & LG_POS_MASK) << 21) | (val & 0x1FFFFFull); }
// local start indices of the clusters that start in [0, nown) into s_cl (ascending), *s_end = where the last one ends (local; may lie beyond the loaded records); returns their count.
// s_flag: the loaded records' flags (the low byte of a 16-bit word: the array is the predecessor distances' afterwards)
__device__ inline uint32_t lg_tile_clusters(const uint8_t* __restrict__ flags, uint32_t total, uint32_t T0, uint32_t nown, uint32_t nload, const uint16_t* s_flag, uint16_t* s_cl, uint32_t* s_cnt, uint32_t* s_end) {
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr uint32_t R = LG_T / LG_TB, W = LG_TB / 64;
  unsigned long long b[R];
#pragma unroll
  for (uint32_t r = 0; r < R; ++r) { const uint32_t i = r * LG_TB + threadIdx.x; b[r] = __ballot(i < nown && (s_flag[i] & LG_CL)); if (lane == 0) s_cnt[r * W + wv] = (uint32_t)__popcll(b[r]); }
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t t = 0; for (uint32_t k = 0; k < LG_T / 64; ++k) { const uint32_t v = s_cnt[k]; s_cnt[k] = t; t += v; } s_cnt[LG_T / 64] = t; }
  __syncthreads();
  const uint32_t ncl = s_cnt[LG_T / 64];
#pragma unroll
  for (uint32_t r = 0; r < R; ++r) { const uint32_t i = r * LG_TB + threadIdx.x; if ((b[r] >> lane) & 1) s_cl[s_cnt[r * W + wv] + (uint32_t)__popcll(b[r] & ((1ull << lane) - 1))] = (uint16_t)i; }
  if (threadIdx.x == 0 && ncl) {
    uint32_t i = nown;
    while (T0 + i < total) { const uint32_t f = i < nload ? (uint32_t)s_flag[i] : (uint32_t)flags[T0 + i]; if (f & LG_CL) break; ++i; }
    *s_end = i;
  }
  __syncthreads();
  return ncl;
}
__global__ void __launch_bounds__(LG_TB) k_lg_dp2(uint32_t total, uint32_t ncap /* records a block loads: LG_T .. LG_N (tests lower it to send clusters across the edge) */, const uint64_t* __restrict__ skey,
                                                  const uint64_t* __restrict__ sval, const uint8_t* __restrict__ flags, const uint64_t* __restrict__ se, sq_map_params P,
                                                  const double* __restrict__ gapcost, double* __restrict__ cf, int32_t* __restrict__ cp, uint8_t* __restrict__ mused, uint64_t* __restrict__ gbest) {
  __shared__ LgRec s_rec[LG_N]; __shared__ double s_gap[SQ_MAX_CHAIN_GAP + 1]; __shared__ uint32_t s_cnt[LG_T / 64 + 1]; __shared__ uint32_t s_end; __shared__ uint16_t s_cp[LG_N]; __shared__ uint16_t s_cl[LG_T];
  const uint32_t T0 = blockIdx.x * LG_T; const uint32_t nown = min(LG_T, total - T0), nload = min(ncap, total - T0);
  for (uint32_t i = threadIdx.x; i < nload; i += LG_TB) { s_rec[i].pv = lg_pv(skey[T0 + i], sval[T0 + i]); s_cp[i] = flags[T0 + i]; }
  for (uint32_t i = threadIdx.x; i <= SQ_MAX_CHAIN_GAP; i += LG_TB) s_gap[i] = gapcost[i];
  __syncthreads();
  const uint32_t ncl = lg_tile_clusters(flags, total, T0, nown, nload, s_cp, s_cl, s_cnt, &s_end);
  if (ncl == 0) return;
  for (uint32_t c = threadIdx.x; c < ncl; c += LG_TB) {
    const uint32_t s0 = s_cl[c], s1 = c + 1 < ncl ? (uint32_t)s_cl[c + 1] : s_end;
    double best = 0.0;
    for (uint32_t i = s0; i < s1; ++i) {
      const uint64_t ipv = i < nload ? s_rec[i].pv : lg_pv(skey[T0 + i], sval[T0 + i]);
      const int iq = (int)((ipv >> 10) & 1023), il = (int)(ipv & 1023); const uint32_t ifw = (uint32_t)(ipv >> 20) & 1u; const int64_t ip = (int64_t)(ipv >> 21);
      double fi = (double)il; int pi = -1; int rounds = 2;
      for (int j = (int)i - 1; j >= (int)s0; --j) {
        uint64_t jpv; double jf;
        if ((uint32_t)j < nload) { const LgRec r = s_rec[j]; jpv = r.pv; jf = r.f; } else { jpv = lg_pv(skey[T0 + j], sval[T0 + j]); jf = cf[T0 + j]; }
        const int64_t rd64 = ip - (int64_t)(jpv >> 21);
        if (rd64 > SQ_MAX_CHAIN_GAP) break;
        if (((uint32_t)(jpv >> 20) & 1u) != ifw) continue;
        const int rd = (int)rd64, qd = iq - (int)((jpv >> 10) & 1023);
        if (qd < 0 || max(qd, rd) > SQ_MAX_CHAIN_GAP) continue;
        const int l = abs(qd - rd);
        const double a = (double)min(il, min(qd, rd));
        const double sc = jf + a - s_gap[l];
        if (sc > fi) { fi = sc; pi = j; }
        if (!P.no_heuristic && pi >= 0) { if (--rounds <= 0) break; }
      }
      if (i < nload) { s_rec[i].f = fi; s_cp[i] = pi >= 0 ? (uint16_t)(i - (uint32_t)pi) : (uint16_t)0; }
      else { cf[T0 + i] = fi; cp[T0 + i] = pi >= 0 ? (int32_t)(T0 + (uint32_t)pi) : -1; mused[T0 + i] = 0; }
      if (fi > best) best = fi;
    }
    atomicMax((unsigned long long*)&gbest[(uint32_t)se[T0 + s0]], (unsigned long long)__double_as_longlong(best));   // f > 0: the bit pattern orders like the value
  }
  __syncthreads();
  const uint32_t lo = s_cl[0], hi = min(s_end, nload);
  for (uint32_t i = lo + threadIdx.x; i < hi; i += LG_TB) { cf[T0 + i] = s_rec[i].f; const uint32_t d = s_cp[i]; cp[T0 + i] = d ? (int32_t)(T0 + i - d) : -1; mused[T0 + i] = 0; }
}
// chain ends by (score desc, index asc) among the cluster's MEMs over the group's threshold; a chain that runs into an accepted one is dropped (SPEC §a2).
// mused: 1 member of an accepted chain, 2 tried and dropped, 4 the last MEM of an accepted chain
__global__ void __launch_bounds__(LG_TB) k_lg_accept2(uint32_t total, uint32_t ncap, const uint8_t* __restrict__ flags, const uint64_t* __restrict__ se, sq_map_params P, const double* __restrict__ cf,
                                                      const int32_t* __restrict__ cp, uint8_t* __restrict__ mused, const uint64_t* __restrict__ gbest, uint64_t* __restrict__ ebest) {
  __shared__ double s_cf[LG_N]; __shared__ uint32_t s_cnt[LG_T / 64 + 1]; __shared__ uint32_t s_end; __shared__ uint16_t s_cp[LG_N]; __shared__ uint16_t s_fm[LG_N]; __shared__ uint16_t s_cl[LG_T];
  const uint32_t T0 = blockIdx.x * LG_T; const uint32_t nown = min(LG_T, total - T0), nload = min(ncap, total - T0);
  for (uint32_t i = threadIdx.x; i < nload; i += LG_TB) { s_cf[i] = cf[T0 + i]; const int32_t p = cp[T0 + i]; s_cp[i] = p >= 0 ? (uint16_t)(T0 + i - (uint32_t)p) : (uint16_t)0; s_fm[i] = flags[T0 + i]; }
  __syncthreads();
  const uint32_t ncl = lg_tile_clusters(flags, total, T0, nown, nload, s_fm, s_cl, s_cnt, &s_end);
  if (ncl == 0) return;
  for (uint32_t i = threadIdx.x; i < nload; i += LG_TB) s_fm[i] = 0;    // the flags are done with: the array holds `mused` from here on
  __syncthreads();
  // (a predecessor's distance fits 16 bits for every record in LDS: it lies in the same cluster, which starts in this tile)
#define LG_CF(i) ((i) < nload ? s_cf[i] : cf[T0 + (i)])
#define LG_MU(i) ((i) < nload ? (uint32_t)s_fm[i] : (uint32_t)mused[T0 + (i)])
#define LG_PREV(x) ((uint32_t)(x) < nload ? (s_cp[x] ? (int)(x) - (int)s_cp[x] : -1) : (cp[T0 + (x)] >= 0 ? (int)((uint32_t)cp[T0 + (x)] - T0) : -1))
  for (uint32_t c = threadIdx.x; c < ncl; c += LG_TB) {
    const uint32_t s0 = s_cl[c], s1 = c + 1 < ncl ? (uint32_t)s_cl[c + 1] : s_end;
    const uint64_t sev = se[T0 + s0];
    const double thr = P.pre_thr * __longlong_as_double((long long)gbest[(uint32_t)sev]);
    double top = 0.0;
    for (;;) {
      int bi = -1; double bf = 0.0;
      for (uint32_t i = s0; i < s1; ++i) {
        if (LG_MU(i)) continue;
        const double fv = LG_CF(i);
        if (fv >= thr && (bi < 0 || fv > bf)) { bi = (int)i; bf = fv; }
      }
      if (bi < 0) break;
      bool clash = false;
      for (int x = bi; x >= 0; x = LG_PREV(x)) if (LG_MU((uint32_t)x) & 1) { clash = true; break; }
      if (clash) { if ((uint32_t)bi < nload) s_fm[bi] |= 2; else mused[T0 + bi] |= 2; continue; }
      for (int x = bi; x >= 0; x = LG_PREV(x)) { if ((uint32_t)x < nload) s_fm[x] |= 1; else mused[T0 + x] |= 1; }
      if ((uint32_t)bi < nload) s_fm[bi] |= 4; else mused[T0 + bi] |= 4;
      if (bf > top) top = bf;
    }
    if (top > 0.0) atomicMax((unsigned long long*)&ebest[(uint32_t)(sev >> 32)], (unsigned long long)__double_as_longlong(top));
  }
#undef LG_CF
#undef LG_MU
#undef LG_PREV
  __syncthreads();
  const uint32_t lo = s_cl[0], hi = min(s_end, nload);
  for (uint32_t i = lo + threadIdx.x; i < hi; i += LG_TB) mused[T0 + i] = (uint8_t)s_fm[i];
}
// hitFilterPolicy AFTER + consensus fraction over the end's chains (k_chain's last loop): which accepted chains stay
__global__ void k_lg_keep(uint32_t total, const uint64_t* __restrict__ se, sq_map_params P, const double* __restrict__ cf, const uint8_t* __restrict__ mused,
                          const uint64_t* __restrict__ ebest, uint8_t* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= total) return;
  uint8_t k = 0;
  if (mused[i] & 4) { const double cthr = P.consensus_frac * __longlong_as_double((long long)ebest[(uint32_t)(se[i] >> 32)]); k = cf[i] >= cthr ? 1 : 0; }
  keep[i] = k;
}
__global__ void k_lg_key_score(uint32_t n, const uint32_t* __restrict__ idx, const double* __restrict__ cf, uint64_t* __restrict__ key) {   // ascending in ~bits = descending in score
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; if (k < n) key[k] = ~(uint64_t)__double_as_longlong(cf[idx[k]]);
}
__global__ void k_lg_key_group(uint32_t n, const uint32_t* __restrict__ idx, const uint64_t* __restrict__ skey, const uint64_t* __restrict__ sval, int tbits, uint64_t* __restrict__ key) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; if (k < n) { const uint32_t i = idx[k]; key[k] = ((skey[i] >> 40) << tbits) | (sval[i] >> 32); }
}
__global__ void k_lg_first(uint32_t n, const uint64_t* __restrict__ gkey, int tbits, uint32_t* __restrict__ first) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; if (k >= n) return;
  const uint32_t e = (uint32_t)(gkey[k] >> tbits);
  if (k == 0 || (uint32_t)(gkey[k - 1] >> tbits) != e) first[e] = k;
}
// chain k of the sorted list: its record into the end's slab (members linked through mnext, as k_chain's groups of more than CH_SMALL MEMs)
__global__ void k_lg_write(uint32_t n, const uint64_t* __restrict__ gkey, int tbits, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ first,
                           const uint64_t* __restrict__ skey, const uint64_t* __restrict__ sval, const uint64_t* __restrict__ se, const uint64_t* __restrict__ ref_accum,
                           const uint16_t* __restrict__ rlen, const uint64_t* __restrict__ mem_off, const double* __restrict__ cf, const int32_t* __restrict__ cp,
                           uint32_t* __restrict__ mnext, sq_chain_dev* __restrict__ chains, uint32_t* __restrict__ n_chains) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; if (k >= n) return;
  const uint64_t gk = gkey[k]; const uint32_t e = (uint32_t)(gk >> tbits); const uint32_t tid = (uint32_t)(gk & ((1ull << tbits) - 1));
  const uint32_t bi = idx[k]; const uint32_t es = (uint32_t)(se[bi] >> 32); const uint64_t base = mem_off[e];
  uint32_t cnt = 0; int fi = (int)bi;
  mnext[base + (bi - es)] = 0xFFFFFFFFu;
  for (int x = (int)bi; x >= 0; x = cp[x]) { ++cnt; fi = x; const int pr = cp[x]; if (pr >= 0) mnext[base + ((uint32_t)pr - es)] = (uint32_t)x - es; }
  const uint64_t ra = ref_accum[tid];
  const LgMem m0 = lg_mem(skey, sval, (uint32_t)fi, ra), ml = lg_mem(skey, sval, bi, ra);
  sq_chain_dev c;
  c.score = cf[bi]; c.tid = tid; c.pos = m0.r - m0.q; c.last_end = ml.r + ml.len; c.first = (uint32_t)fi - es; c.n_mems = (uint16_t)cnt; c.read_len = rlen[e];
  c.fw = (uint8_t)ml.fw; c.pad[0] = c.pad[1] = c.pad[2] = 0; c.pad2 = 0; c.spare = 0;
  const uint32_t w = k - first[e];
  chains[base + w] = c;
  if (k + 1 == n || (uint32_t)(gkey[k + 1] >> tbits) != e) n_chains[e] = w + 1;
}

}  // namespace sqk
