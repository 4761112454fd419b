// oracle/oracle.cpp — CPU restatement of the `salmon quant` hot path.  TEST INFRASTRUCTURE ONLY.
//
// This file is the checker the HIP path is compared against (tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py are its only callers).  Nothing under salmon_amd/ links, imports or
// calls it; the product path fails loudly without the HIP extension.
//
// PARITY STATUS: **parity unpinned** for rows a1-a6 (k-mer lookup / MEM collection / chaining /
// pair joining / selective-alignment scoring): that arithmetic lives in COMBINE-lab/pufferfish @
// ace68c1c022816ba8c50a1a07c5d08f2abd597d6 (cmake/SalmonDependencies.cmake:11-16), which is absent
// from /root/reference and cannot be fetched; the reference's tests hold no golden vectors for it
// (SURVEY.md §4, §8c).  Those stages restate the published algorithms (SSHash, pufferfish uni-MEMs,
// minimap2 chaining, ksw2 affine DP) with the choices frozen in oracle/SPEC.md.  Rows a7-a18 follow
// the in-tree reference files line by line; each function cites them.  Library-format compatibility
// (a9) IS pinned by the reference's tests/LibraryTypeTests.cpp truth tables (tests/test_libtype.py).
// The infix aligner of orphan recovery (a5) IS pinned against the reference's own src/edlib.cpp, compiled from
// /root/reference into oracle/_ref/libedlib_ref.so (oracle/Makefile `ref`), and through 600 committed vectors made
// from it (tests/golden/edlib_infix_vectors.json.gz).
// Compiled from the reference tree the same way and held against this file (tests/test_*_pin.py): the option defaults and log-space helpers, the
// forgetting-mass schedule, SimplePosBias.cpp + spline.h (--posBias), SBModel.cpp and GCFragModel.hpp (--seqBias / --gcBias),
// [r4] FragmentLengthDistribution.cpp and DistributionUtils.cpp (the FLD, the effective lengths of burn-in, an orphan's fragment-length probability:
// rows a10-a13) and EMUtils.cpp (the EM update: rows a15-a16); [r5] CollapsedEMOptimizer.cpp itself — the serial VBEMUpdate_ and the whole of
// CollapsedEMOptimizer::optimize, the default optimiser, with TBB / Boost / spdlog / ReadExperiment stood in for (oracle/_stub/vbem; tests/test_vbem_pin.py: same
// iteration counts, alphas to 1e-9; the pin found that the reference's plain-EM first iteration adds into alphasPrime left at 1.0 — now followed) and
// TranscriptCluster.hpp + ClusterForest.hpp (projectToPolytope and the cluster forest of normalizeAlphas, row a14: tests/test_polytope_pin.py), AlignmentModel.cpp
// (the CIGAR error model of alignment-based input: tests/test_alnmodel_pin.py) and SalmonMappingUtils.hpp's updateRefMappings / filterAndCollectAlignments with stand-in
// pufferfish types (rows a7 / a8: tests/test_selection_pin.py); gatherBootstraps / doBootstrap and CollapsedGibbsSampler.cpp (rows a16 / a17, compared in distribution: tests/test_bootstrap_pin.py, tests/test_gibbs_pin.py).  What cannot be
// compiled here (SalmonQuantify.cpp, SalmonUtils.cpp: Boost, TBB, pufferfish) is followed line by line and cited.
//
// Deliberate, documented deviations from the (nondeterministic) reference: see oracle/SPEC.md §D.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../include/salmon_hip.h"
#include "../include/sq_math.h"
#include "../include/sq_rng.h"

namespace orc {

// [r4] no cap on the uni-MEMs of a read end (SPEC §a1: the reference keeps them all; the product widens its slab until they fit)
constexpr int MAX_CHAIN_GAP = 200;   // SPEC §a2
constexpr double AVG_SEED = 31.0;    // SPEC §a2 (pufferfish uses a constant average seed length)
constexpr int REF_EXTEND = 20;       // aconf.refExtendLength (SalmonMappingUtils.hpp:184)
constexpr int32_t INVALID_SCORE = INT32_MIN;
constexpr int32_t NEG_INF = -(1 << 29);

// ---- own small k-mer helpers (independent of salmon_amd/csrc/sq_internal.h) ----------------------
static inline uint64_t kmask(uint32_t k) { return k >= 32 ? ~0ULL : (1ULL << (2 * k)) - 1; }
static inline uint64_t revcomp(uint64_t x, uint32_t k) {
  uint64_t r = 0;
  for (uint32_t i = 0; i < k; ++i) { r = (r << 2) | (3 - (x & 3)); x >>= 2; }
  return r;
}
static inline uint32_t base_at(const uint64_t* pool, uint64_t p) { return (uint32_t)(pool[p >> 5] >> ((p & 31) * 2)) & 3u; }

struct Index {
  uint32_t k = 31, first_decoy = 0;
  std::vector<std::string> names;
  std::vector<uint32_t> ref_len, ref_clen;
  std::vector<uint64_t> ref_accum, refseq, useq, uoff, ctab_off, ctab;
  // brute-force dictionary: canonical k-mer -> (unitig<<31 | off<<1 | canonical_is_fw_in_unitig)
  std::vector<uint64_t> hkeys, hvals; uint64_t hmask = 0;
  uint64_t num_kmers = 0;

  void build_dict() {
    uint64_t U = uoff.size() - 1, nk = 0;
    for (uint64_t u = 0; u < U; ++u) nk += uoff[u + 1] - uoff[u] - (k - 1);
    num_kmers = nk;
    uint64_t cap = 16; while (cap < nk * 2) cap <<= 1;
    hkeys.assign(cap, ~0ULL); hvals.assign(cap, 0); hmask = cap - 1;
    const uint64_t km = kmask(k);
    // every canonical k-mer occurs once in the unitigs (cDBG), so threads insert disjoint keys: one CAS claims a slot
    unsigned nt = std::max(1u, std::min(64u, std::thread::hardware_concurrency())); if (nk < (1u << 20)) nt = 1;
    std::atomic<uint64_t> next(0); std::vector<std::thread> th;
    auto work = [&]() {
      for (;;) {
        const uint64_t u0 = next.fetch_add(1024); if (u0 >= U) break; const uint64_t u1 = std::min(U, u0 + 1024);
        for (uint64_t u = u0; u < u1; ++u) {
          uint64_t b = uoff[u]; uint32_t ulen = (uint32_t)(uoff[u + 1] - b);
          uint64_t fw = 0;
          for (uint32_t i = 0; i < ulen; ++i) {
            fw = (fw >> 2) | ((uint64_t)base_at(useq.data(), b + i) << (2 * (k - 1)));
            if (i + 1 < k) continue;
            fw &= km; uint64_t rc = revcomp(fw, k); bool cf = fw < rc; uint64_t c = cf ? fw : rc;
            uint64_t h = sq_mix64(c) & hmask;
            for (;;) {
              uint64_t cur = __atomic_load_n(&hkeys[h], __ATOMIC_RELAXED);
              if (cur == ~0ULL) { uint64_t exp = ~0ULL; if (__atomic_compare_exchange_n(&hkeys[h], &exp, c, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break; cur = exp; }
              if (cur == c) break;
              h = (h + 1) & hmask;
            }
            hvals[h] = (u << 31) | ((uint64_t)(i + 1 - k) << 1) | (cf ? 1 : 0);
          }
        }
      }
    };
    if (nt == 1) work(); else { for (unsigned t = 0; t < nt; ++t) th.emplace_back(work); for (auto& x : th) x.join(); }
  }
  // SPEC §a1: canonical k-mer -> (unitig, offset, orientation of the *query* w.r.t. the unitig)
  inline bool lookup(uint64_t kmer, uint64_t& u, uint32_t& off, bool& fw) const {
    uint64_t rc = revcomp(kmer, k); bool qcf = kmer < rc; uint64_t c = qcf ? kmer : rc;
    uint64_t h = sq_mix64(c) & hmask;
    for (;;) {
      uint64_t kk = hkeys[h];
      if (kk == ~0ULL) return false;
      if (kk == c) { uint64_t v = hvals[h]; u = v >> 31; off = (uint32_t)((v >> 1) & 0x3FFFFFFF); fw = ((v & 1) != 0) == qcf; return true; }
      h = (h + 1) & hmask;
    }
  }
};

// ---- data carried between stages ----------------------------------------------------------------
struct UniMem { uint16_t qpos, len; uint64_t unitig; uint32_t ustart; bool fw; };
struct Mem { uint32_t tid; int32_t rpos; uint16_t q, len; bool fw; };  // q = strand-normalised read pos
struct Chain { uint32_t tid; bool fw; double score; int32_t pos; int32_t last_end; uint16_t read_len; std::vector<uint32_t> mems; /* indices into the end's Mem array, ascending */ };
struct Cand { uint32_t tid; int lc, rc; uint32_t frag_len; uint8_t mate_status; double cov; int32_t lscore = INVALID_SCORE,
    rscore = INVALID_SCORE; bool valid = false; };

struct Opts { sq_quant_opts o; std::vector<double> gapcost; int32_t ma, mp, go, ge, bw; };

static void make_opts(const sq_quant_opts* o, Opts& op) {
  op.o = *o; op.ma = o->match_score; op.mp = o->mismatch_penalty; op.go = o->gap_open; op.ge = o->gap_extend; op.bw = o->bandwidth;
  op.gapcost.assign(MAX_CHAIN_GAP + 1, 0.0);
  const double inv_ln2 = 1.0 / 0.6931471805599453;
  for (int l = 1; l <= MAX_CHAIN_GAP; ++l) op.gapcost[l] = 0.01 * AVG_SEED * (double)l + 0.5 * (sq_log((double)l) * inv_ln2);
}

struct ReadCodes { std::vector<uint8_t> c; };  // 0..3, 4 = N
static inline void encode_read(const uint8_t* s, uint32_t n, std::vector<uint8_t>& c) {
  c.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    switch (s[i]) {
      case 'A': case 'a': c[i] = 0;
      break;
      case 'C': case 'c': c[i] = 1;
      break;
      case 'G': case 'g': c[i] = 2;
      break;
      case 'T': case 't': c[i] = 3;
      break;
      default: c[i] = 4;
    }
  }
}

// a1 — MemCollector::operator() [external pufferfish]; call site SalmonQuantify.cpp:1266-1275.
static void collect_unimems(const Index& ix, const Opts& op, const std::vector<uint8_t>& rd, std::vector<UniMem>& out, sq_map_stats* st) {
  out.clear();
  const uint32_t k = ix.k; const int L = (int)rd.size();
  if (L < (int)k) return;
  int pos = 0, skip_until = -1; const int alt = (int)op.o.mismatch_seed_skip;
  while (pos + (int)k <= L) {
    int lastN = -1; for (int i = pos; i < pos + (int)k; ++i) if (rd[i] > 3) lastN = i;
    if (lastN >= 0) { pos = lastN + 1; continue; }
    uint64_t km = 0; for (uint32_t i = 0; i < k; ++i) km |= (uint64_t)rd[pos + i] << (2 * i);
    uint64_t u; uint32_t off; bool fw;
    if (st) st->num_lookups++;
    if (!ix.lookup(km, u, off, fw)) {
      if (pos < skip_until) { int np = pos + alt; if (np > skip_until) np = skip_until; pos = np; } else pos += 1;
      continue;
    }
    const uint64_t ub = ix.uoff[u]; const int ulen = (int)(ix.uoff[u + 1] - ub);
    int len = k; bool uend = false;
    if (fw) {
      for (;;) {
        if (pos + len >= L) break;
        if ((int)off + len >= ulen) {
          uend = true;
          break;
        }
        if (rd[pos + len] != base_at(ix.useq.data(), ub + off + len)) break;
        ++len;
      }
    } else {
      for (;;) {
        if (pos + len >= L) break;
        int up = (int)off - 1 - (len - (int)k);
        if (up < 0) {
          uend = true;
          break;
        }
        if (rd[pos + len] != 3 - base_at(ix.useq.data(), ub + up)) break;
        ++len;
      }
    }
    UniMem m;
    m.qpos = (uint16_t)pos;
    m.len = (uint16_t)len;
    m.unitig = u;
    m.fw = fw;
    m.ustart = fw ? off : (uint32_t)((int)off - (len - (int)k));
    out.push_back(m);
    if (pos + len >= L) break;
    int e = pos + len;
    pos = pos + len - (int)k + 1;
    skip_until = uend ? -1 : e + 1;
  }
  if (st) st->num_seeds += out.size();
}

// a2a — projection of uni-MEMs through the contig table (MemClusterer::fillMemCollection [external]).
static void project_mems(const Index& ix, const Opts& op, const std::vector<UniMem>& um, int L, std::vector<Mem>& mems) {
  mems.clear();
  for (const UniMem& m : um) {
    uint64_t a = ix.ctab_off[m.unitig], b = ix.ctab_off[m.unitig + 1];
    if (b - a > op.o.max_occs_per_hit) continue;  // configureMemClusterer(maxOccsPerHit), SalmonMappingUtils.hpp:159
    int ulen = (int)(ix.uoff[m.unitig + 1] - ix.uoff[m.unitig]);
    for (uint64_t i = a; i < b; ++i) {
      uint64_t e = ix.ctab[i]; Mem x; x.tid = (uint32_t)(e >> 32); bool ufw = (e >> 31) & 1; int upos = (int)(e & 0x7FFFFFFF);
      x.rpos = ufw ? upos + (int)m.ustart : upos + (ulen - ((int)m.ustart + (int)m.len));
      x.fw = (ufw == m.fw); x.len = m.len; x.q = x.fw ? m.qpos : (uint16_t)(L - (m.qpos + m.len));
      mems.push_back(x);
    }
  }
  // SPEC §a2: order by (tid, refPos); ties keep (uni-MEM, occurrence) emission order
  std::stable_sort(mems.begin(), mems.end(), [&](const Mem& a, const Mem& b) {
    uint64_t ka = ix.ref_accum[a.tid] + (uint64_t)(a.rpos < 0 ? 0 : a.rpos), kb = ix.ref_accum[b.tid] + (uint64_t)(b.rpos < 0 ? 0 : b.rpos);
    return ka < kb; });
}

// a2b — MemCollector::findChains / findOptChain [external]; call site SalmonQuantify.cpp:1276-1288.
static void chain_end(const Index&, const Opts& op, const std::vector<Mem>& mems, int L, std::vector<Chain>& chains) {
  chains.clear();
  const size_t n = mems.size();
  std::vector<double> f; std::vector<int> p; std::vector<uint8_t> used;
  size_t g0 = 0;
  while (g0 < n) {
    size_t g1 = g0; while (g1 < n && mems[g1].tid == mems[g0].tid) ++g1;
    const size_t gn = g1 - g0;
    f.assign(gn, 0); p.assign(gn, -1); used.assign(gn, 0);
    double best = 0;
    for (size_t i = 0; i < gn; ++i) {
      const Mem& hi = mems[g0 + i];
      f[i] = (double)hi.len; p[i] = -1; int rounds = 2;
      for (int j = (int)i - 1; j >= 0; --j) {
        const Mem& hj = mems[g0 + j];
        if (hj.fw != hi.fw) continue;
        int qd = (int)hi.q - (int)hj.q, rd = hi.rpos - hj.rpos;
        if (qd < 0 || std::max(qd, rd) > MAX_CHAIN_GAP) continue;
        int l = std::abs(qd - rd);
        double a = (double)std::min((int)hi.len, std::min(qd, rd));
        double s = f[j] + a - op.gapcost[l];
        if (s > f[i]) { f[i] = s; p[i] = j; }
        // Li 2018 heuristic (ProgramOptionsGenerator.cpp:160-167)
        if (!op.o.disable_chaining_heuristic && p[i] >= 0) {
          if (--rounds <= 0) break;
        }
      }
      if (f[i] > best) best = f[i];
    }
    const double thr = op.o.pre_merge_chain_sub_thresh * best;  // ProgramOptionsGenerator.cpp:111-118
    // accept chain ends by (score desc, index asc); a chain touching an already used MEM is dropped
    std::vector<uint8_t> tried(gn, 0);
    for (;;) {
      int bi = -1;
      for (size_t i = 0; i < gn; ++i) if (!tried[i] && !used[i] && f[i] >= thr && (bi < 0 || f[i] > f[bi])) bi = (int)i;
      if (bi < 0) break;
      tried[bi] = 1;
      bool clash = false; for (int x = bi; x >= 0; x = p[x]) if (used[x]) { clash = true; break; }
      if (clash) continue;
      Chain c; c.tid = mems[g0].tid; c.fw = mems[g0 + bi].fw; c.score = f[bi]; c.read_len = (uint16_t)L;
      for (int x = bi; x >= 0; x = p[x]) { used[x] = 1; c.mems.push_back((uint32_t)(g0 + x)); }
      std::reverse(c.mems.begin(), c.mems.end());
      const Mem& m0 = mems[c.mems.front()]; const Mem& ml = mems[c.mems.back()];
      c.pos = m0.rpos - (int)m0.q; c.last_end = ml.rpos + (int)ml.len;
      chains.push_back(std::move(c));
    }
    g0 = g1;
  }
  // hitFilterPolicy AFTER + consensusSlack (ProgramOptionsGenerator.cpp:103-110): per read end
  double bestAll = 0; for (auto& c : chains) bestAll = std::max(bestAll, c.score);
  const double cf = (op.o.consensus_slack == 0.0) ? 1.0 : (1.0 - op.o.consensus_slack);  // SalmonMappingUtils.hpp:160-162
  const double cthr = cf * bestAll;
  chains.erase(std::remove_if(chains.begin(), chains.end(), [&](const Chain& c) { return c.score < cthr; }), chains.end());
}

// a3 — pufferfish::util::joinReadsAndFilter [external]; call site SalmonQuantify.cpp:1339-1341,
// policy from SalmonMappingUtils.hpp:208-220.
static int join_pair(const Opts& op, const std::vector<Chain>& lc, const std::vector<Chain>& rc, std::vector<Cand>& out,
    bool* had_dovetail) {
  out.clear(); *had_dovetail = false;
  size_t i = 0, j = 0;
  while (i < lc.size() && j < rc.size()) {
    if (lc[i].tid < rc[j].tid) { ++i; continue; }
    if (lc[i].tid > rc[j].tid) { ++j; continue; }
    uint32_t t = lc[i].tid; size_t i1 = i, j1 = j;
    while (i1 < lc.size() && lc[i1].tid == t) ++i1;
    while (j1 < rc.size() && rc[j1].tid == t) ++j1;
    for (size_t a = i; a < i1; ++a) for (size_t b = j; b < j1; ++b) {
      const Chain& x = lc[a]; const Chain& y = rc[b];
      if (x.fw == y.fw) continue;  // mpol.noDiscordant = true
      const Chain& fwc = x.fw ? x : y; const Chain& rcc = x.fw ? y : x;
      if (rcc.pos < fwc.pos) { *had_dovetail = true; if (!op.o.allow_dovetail) continue; }
      int32_t fragEnd = rcc.pos + (int)rcc.read_len, fragStart = fwc.pos;
      int32_t fl = fragEnd - fragStart;
      if (fl <= 0 || fl > (int32_t)op.o.frag_len_max) continue;
      Cand c;
      c.tid = t;
      c.lc = (int)a;
      c.rc = (int)b;
      c.frag_len = (uint32_t)fl;
      c.mate_status = SQ_MS_PAIRED_END_PAIRED;
      c.cov = x.score + y.score;
      out.push_back(c);
    }
    i = i1; j = j1;
  }
  const double cf = (op.o.consensus_slack == 0.0) ? 1.0 : (1.0 - op.o.consensus_slack);
  if (!out.empty()) {
    double best = 0; for (auto& c : out) best = std::max(best, c.cov);
    double thr = cf * best;
    out.erase(std::remove_if(out.begin(), out.end(), [&](const Cand& c) { return c.cov < thr; }), out.end());
    // post-merge sub-optimality per target (ProgramOptionsGenerator.cpp:119-129)
    std::vector<Cand> kept; size_t a = 0;
    while (a < out.size()) {
      size_t b = a; double bt = 0; while (b < out.size() && out[b].tid == out[a].tid) { bt = std::max(bt, out[b].cov); ++b; }
      for (size_t c = a; c < b; ++c) if (out[c].cov >= op.o.post_merge_chain_sub_thresh * bt) kept.push_back(out[c]);
      a = b;
    }
    out.swap(kept);
    return 1;  // HAD_CONCORDANT
  }
  if (!op.o.allow_orphans) return 0;
  double best = 0;
  for (auto& c : lc) best = std::max(best, c.score);
  for (auto& c : rc) best = std::max(best, c.score);
  double thr = op.o.orphan_chain_sub_thresh * best;  // global (ProgramOptionsGenerator.cpp:130-137)
  for (size_t a = 0; a < lc.size(); ++a) if (lc[a].score >= thr) {
    Cand c;
    c.tid = lc[a].tid;
    c.lc = (int)a;
    c.rc = -1;
    c.frag_len = 0;
    c.mate_status = SQ_MS_PAIRED_END_LEFT;
    c.cov = lc[a].score;
    out.push_back(c);
  }
  for (size_t b = 0; b < rc.size(); ++b) if (rc[b].score >= thr) {
    Cand c;
    c.tid = rc[b].tid;
    c.lc = -1;
    c.rc = (int)b;
    c.frag_len = 0;
    c.mate_status = SQ_MS_PAIRED_END_RIGHT;
    c.cov = rc[b].score;
    out.push_back(c);
  }
  return out.empty() ? 0 : 2;
}

// ---- a5 — selective_alignment::utils::recoverOrphans [external] + in-tree edlib ---------------------
// call site SalmonQuantify.cpp:1356-1364; the infix aligner is reference src/edlib.cpp:290-372 called with
// {k, EDLIB_MODE_HW, EDLIB_TASK_LOC}.  Plain dynamic-programming restatement of what that call returns
// (edit distance, endLocations[0], startLocations[0]); checked against the compiled reference file in
// tests/test_recover.py.  Codes: 0..3 bases, anything else never matches (edlib compares bytes; the
// reference text holds no N).  SPEC §a5.
static bool infix_align(const uint8_t* q, int n, const uint8_t* t, int m, int k, int* ed, int* start, int* end) {
  if (n <= 0 || m <= 0) return false;
  if (k > n) k = n;                                           // edlib.cpp:683-685
  std::vector<int> col(n + 1), lastrow(m + 1);
  for (int i = 0; i <= n; ++i) col[i] = i;                    // D[i][0] = i
  lastrow[0] = n;
  for (int j = 1; j <= m; ++j) {                              // D[0][j] = 0: the gap before the query is free
    int diag = col[0]; col[0] = 0;
    for (int i = 1; i <= n; ++i) {
      int up = col[i];                                        // D[i][j-1]
      int v = std::min(std::min(diag + ((q[i - 1] == t[j - 1] && q[i - 1] < 4) ? 0 : 1), up + 1), col[i - 1] + 1);
      diag = up; col[i] = v;
    }
    lastrow[j] = col[n];
  }
  int best = -1, e0 = -1;
  // first end among the minima
  for (int j = 1; j <= m; ++j) if (lastrow[j] <= k && (best < 0 || lastrow[j] < best)) {
    best = lastrow[j];
    e0 = j - 1;
  }
  if (best < 0) return false;
  // start location (edlib.cpp:353-369): reversed query against the reversed target prefix [0, e0], prefix mode
  // (D'[0][j] = j), score limit = best; the LAST column that reaches `best` gives the start.
  const int m2 = e0 + 1; int jlast = -1;
  for (int i = 0; i <= n; ++i) col[i] = i;
  for (int j = 1; j <= m2; ++j) {
    int diag = col[0]; col[0] = j;
    const uint8_t tc = t[e0 - (j - 1)];
    for (int i = 1; i <= n; ++i) {
      int up = col[i]; const uint8_t qc = q[n - i];
      int v = std::min(std::min(diag + ((qc == tc && qc < 4) ? 0 : 1), up + 1), col[i - 1] + 1);
      diag = up; col[i] = v;
    }
    if (col[n] == best) jlast = j;
  }
  if (jlast < 0) return false;                                // cannot happen: the forward optimum is reachable backwards
  *ed = best; *end = e0; *start = e0 - (jlast - 1);
  return true;
}

#define RECOVER_WINDOW 1000   // "within the maximum fragment length" (doc/source/salmon.rst:323-331); the look-up window of one anchor
// Orphan-only fragments: look for the missing mate next to every anchor.  An anchor on the forward strand expects its
// mate reverse-complemented downstream (window = [max(0,pos), +1000) clipped to the transcript); an anchor on the
// reverse strand expects the mate forward, upstream (window = the 1000 bases ending at the anchor's end).  The mate
// (strand-normalised) is placed by infix alignment with at most len/4 edits.  A recovered mate becomes a chain without
// MEMs at the recovered start; the candidate becomes a proper pair.  Returns true if any mate was recovered.
static bool recover_orphans(const Index& ix, const Opts& op, const std::vector<uint8_t> rd[2], std::vector<Chain> ch[2],
    std::vector<Cand>& cands) {
  bool any = false;
  std::vector<uint8_t> qn, win;
  for (auto& c : cands) {
    if (c.mate_status != SQ_MS_PAIRED_END_LEFT && c.mate_status != SQ_MS_PAIRED_END_RIGHT) continue;
    const int ae = c.mate_status == SQ_MS_PAIRED_END_LEFT ? 0 : 1, me = 1 - ae;
    const Chain anchor = ch[ae][ae == 0 ? c.lc : c.rc];
    const int ML = (int)rd[me].size(); if (ML == 0) continue;
    const int Tlen = (int)ix.ref_len[anchor.tid]; const uint64_t g = ix.ref_accum[anchor.tid];
    int ws, wl;
    if (anchor.fw) { ws = std::max(0, anchor.pos); wl = std::min(RECOVER_WINDOW, Tlen - ws); }
    else { int endPos = std::min(Tlen, anchor.pos + (int)anchor.read_len); ws = std::max(0, endPos - RECOVER_WINDOW); wl = endPos - ws; }
    if (wl <= 0) continue;
    const bool mate_fw = !anchor.fw;
    qn.resize(ML);
    if (mate_fw) qn = rd[me]; else for (int i = 0; i < ML; ++i) { uint8_t b = rd[me][ML - 1 - i]; qn[i] = b > 3 ? 4 : (uint8_t)(3 - b); }
    win.resize(wl); for (int i = 0; i < wl; ++i) win[i] = (uint8_t)base_at(ix.refseq.data(), g + (uint64_t)(ws + i));
    int ed, st, en;
    if (!infix_align(qn.data(), ML, win.data(), wl, ML / 4, &ed, &st, &en)) continue;
    const int mpos = ws + st;
    const int fl = anchor.fw ? (mpos + ML - anchor.pos) : (anchor.pos + (int)anchor.read_len - mpos);
    if (fl <= 0 || fl > (int)op.o.frag_len_max) continue;     // keeps the invariant of joined pairs (§a3)
    Chain m; m.tid = anchor.tid; m.fw = mate_fw; m.score = 0.0; m.pos = mpos; m.last_end = ws + en + 1; m.read_len = (uint16_t)ML;
    ch[me].push_back(m);
    if (me == 1) c.rc = (int)ch[1].size() - 1; else c.lc = (int)ch[0].size() - 1;
    c.mate_status = SQ_MS_PAIRED_END_PAIRED; c.frag_len = (uint32_t)fl; any = true;
  }
  return any;
}

// ---- a4 — PuffAligner::calculateAlignments [external] with ksw2 affine-gap DP --------------------
// call site SalmonQuantify.cpp:1523-1525; configuration SalmonMappingUtils.hpp:168-206.
// Banded Gotoh (band |i-j| <= bandwidth). mode 0: global (query and target both consumed);
// mode 1: extension (query consumed, target end free).  SPEC §a4.
static int32_t dp_align(const Opts& op, const uint8_t* q, int n, const uint8_t* t, int tl, int mode) {
  const int w = op.bw, go = op.go, ge = op.ge;
  if (n == 0) return mode == 1 ? 0 : (tl == 0 ? 0 : (tl <= w ? -(go + ge * tl) : NEG_INF));
  if (tl == 0) return n <= w ? -(go + ge * n) : NEG_INF;
  const int W = tl + 1;
  std::vector<int32_t> H((size_t)(n + 1) * W, NEG_INF), E((size_t)(n + 1) * W, NEG_INF), F((size_t)(n + 1) * W, NEG_INF);
  H[0] = 0;
  for (int j = 1; j <= tl && j <= w; ++j) { H[j] = -(go + ge * j); E[j] = H[j]; }
  for (int i = 1; i <= n; ++i) {
    if (i <= w) { H[(size_t)i * W] = -(go + ge * i); F[(size_t)i * W] = H[(size_t)i * W]; }
    int jlo = std::max(1, i - w), jhi = std::min(tl, i + w);
    for (int j = jlo; j <= jhi; ++j) {
      size_t c = (size_t)i * W + j;
      int32_t e = std::max(E[c - 1], H[c - 1] - go) - ge;
      int32_t f = std::max(F[c - W], H[c - W] - go) - ge;
      int32_t s = (q[i - 1] == t[j - 1] && q[i - 1] < 4) ? op.ma : op.mp;
      int32_t h = std::max(H[c - W - 1] + s, std::max(e, f));
      E[c] = std::max(e, NEG_INF); F[c] = std::max(f, NEG_INF); H[c] = std::max(h, NEG_INF);
    }
  }
  if (mode == 0) return (std::abs(n - tl) <= w) ? H[(size_t)n * W + tl] : NEG_INF;
  int32_t best = NEG_INF;
  for (int j = std::max(0, n - w); j <= std::min(tl, n + w); ++j) best = std::max(best, H[(size_t)n * W + j]);
  return best;
}

// score of a region; uses the mismatch-count fast path when it is provably optimal (SPEC §a4)
static int32_t region_score(const Opts& op, const uint8_t* q, int n, const uint8_t* t, int tl, int mode, sq_map_stats* st) {
  if (n == 0 && mode == 1) return 0;
  if (n > 0 && ((mode == 0 && tl == n) || (mode == 1 && tl >= n))) {
    int mm = 0; for (int i = 0; i < n; ++i) mm += !(q[i] == t[i] && q[i] < 4);
    int lim = (mode == 0) ? (2 * (op.go + op.ge) + op.ma) : (op.go + op.ge);
    if (mm * (op.ma - op.mp) <= lim) return op.ma * (n - mm) + op.mp * mm;
  }
  if (st) st->num_dp_alignments++;
  return dp_align(op, q, n, t, tl, mode);
}

static int32_t align_chain(const Index& ix, const Opts& op, const Chain& ch, const std::vector<Mem>& mems,
                           const std::vector<uint8_t>& read_fw, sq_map_stats* st) {
  const int L = (int)read_fw.size();
  std::vector<uint8_t> R(L);
  if (ch.fw) R = read_fw; else for (int i = 0; i < L; ++i) { uint8_t c = read_fw[L - 1 - i]; R[i] = c > 3 ? 4 : (uint8_t)(3 - c); }
  const uint32_t tid = ch.tid; const int Tlen = (int)ix.ref_len[tid]; const uint64_t g = ix.ref_accum[tid];
  auto refb = [&](int x) -> uint8_t { return (uint8_t)base_at(ix.refseq.data(), g + (uint64_t)x); };
  std::vector<uint8_t> qb, tb;
  int64_t score = 0; int prevQ = 0, prevR = 0; bool first = true;
  if (ch.mems.empty()) prevR = ch.pos;   // recovered mate (§a5): one extension alignment of the whole read from its start
  for (uint32_t mi : ch.mems) {
    const Mem& m = mems[mi];
    int qs = m.q, rs = m.rpos, ln = m.len;
    if (first) {
      if (qs > 0) {  // read prefix, extended leftwards over <= qs + REF_EXTEND reference bases
        int ws = std::max(0, rs - qs - REF_EXTEND); int tl = std::max(0, rs - ws);
        qb.resize(qs); for (int i = 0; i < qs; ++i) qb[i] = R[qs - 1 - i];
        tb.resize(tl); for (int i = 0; i < tl; ++i) tb[i] = refb(rs - 1 - i);
        score += region_score(op, qb.data(), qs, tb.data(), tl, 1, st);
      }
      first = false;
    } else {
      int ov = std::max(0, std::max(prevQ - qs, prevR - rs));
      if (ov > 0) { qs += ov; rs += ov; ln -= ov; if (ln <= 0) continue; }
      int gq = qs - prevQ, gr = rs - prevR;
      if (gq > 0 || gr > 0) {
        qb.assign(R.begin() + prevQ, R.begin() + qs);
        tb.resize(gr); for (int i = 0; i < gr; ++i) tb[i] = refb(prevR + i);
        score += region_score(op, qb.data(), gq, tb.data(), gr, 0, st);
      }
    }
    score += (int64_t)op.ma * ln; prevQ = qs + ln; prevR = rs + ln;
  }
  if (prevQ < L) {
    int tail = L - prevQ; int we = std::min(Tlen, prevR + tail + REF_EXTEND); int tl = std::max(0, we - prevR);
    qb.assign(R.begin() + prevQ, R.end());
    tb.resize(tl); for (int i = 0; i < tl; ++i) tb[i] = refb(prevR + i);
    score += region_score(op, qb.data(), tail, tb.data(), tl, 1, st);
  }
  if (score < NEG_INF / 2) return INVALID_SCORE;
  int32_t min_acc = (int32_t)(op.o.min_score_fraction * (double)(op.ma * L));
  return (score >= min_acc) ? (int32_t)score : INVALID_SCORE;
}

// ---- a9 — library-format compatibility (src/util/SalmonUtils.cpp:138-298, :531-652) ---------------
enum { T_SE = 0, T_PE = 1 };
enum { O_SAME = 0, O_AWAY = 1, O_TOWARD = 2, O_NONE = 3 };
enum { S_SA = 0, S_AS = 1, S_S = 2, S_A = 3, S_U = 4 };
struct LibFmt { uint8_t type, orient, strand; };
static inline uint8_t format_id(LibFmt f) { return (uint8_t)(f.type | (f.orient << 1) | (f.strand << 3)); }
static LibFmt hit_type_pe(int32_t e1, bool f1, uint32_t l1, int32_t e2, bool f2, uint32_t l2, bool canDovetail) {  // SalmonUtils.cpp:577-631
  if (f1 != f2) {
    if (f1) {
      int32_t stretch = canDovetail ? (int32_t)l2 : 0;
      return (e1 <= e2 + stretch) ? LibFmt{T_PE, O_TOWARD, S_SA} : LibFmt{T_PE, O_AWAY, S_SA};
    }
    int32_t stretch = canDovetail ? (int32_t)l1 : 0;
    return (e2 <= e1 + stretch) ? LibFmt{T_PE, O_TOWARD, S_AS} : LibFmt{T_PE, O_AWAY, S_AS};
  }
  return f1 ? LibFmt{T_PE, O_SAME, S_S} : LibFmt{T_PE, O_SAME, S_A};
}
static LibFmt hit_type_se(bool fwd) { return fwd ? LibFmt{T_SE, O_NONE, S_S} : LibFmt{T_SE, O_NONE, S_A}; }  // :633-648
static bool compatible_hit_se(LibFmt exp, bool isForward, uint8_t ms) {  // SalmonUtils.cpp:195-268
  switch (ms) {
    case SQ_MS_SINGLE_END: return isForward ? (exp.strand == S_U || exp.strand == S_S) : (exp.strand == S_U || exp.strand == S_A);
    case SQ_MS_PAIRED_END_LEFT:
      if (exp.orient == O_SAME) return exp.strand == S_U || (exp.strand == S_S && isForward) || (exp.strand == S_A && !isForward);
      return isForward ? (exp.strand == S_U || exp.strand == S_SA) : (exp.strand == S_U || exp.strand == S_AS);
    case SQ_MS_PAIRED_END_RIGHT:
      if (exp.orient == O_SAME) return exp.strand == S_U || (exp.strand == S_S && isForward) || (exp.strand == S_A && !isForward);
      return isForward ? (exp.strand == S_U || exp.strand == S_AS) : (exp.strand == S_U || exp.strand == S_SA);
    default: return false;
  }
}
static bool compatible_hit_pe(LibFmt exp, LibFmt obs) {  // SalmonUtils.cpp:271-297
  if (obs.type != T_PE) return false;
  if (exp.orient != obs.orient) return false;
  return exp.strand == S_U || exp.strand == obs.strand;
}
static bool is_compatible(LibFmt obs, LibFmt exp, bool isForward, uint8_t ms) {  // :138-148
  return (ms != SQ_MS_PAIRED_END_PAIRED) ? compatible_hit_se(exp, isForward, ms) : compatible_hit_pe(exp, obs);
}
// pre-alignment compatibility of a joint hit (SalmonQuantify.cpp:1467-1516)
static bool joint_compat(LibFmt exp, bool orphan, bool isLeft, bool lfw, bool rfw) {
  bool unstr = exp.strand == S_U;
  bool c = unstr ? (orphan ? true : (lfw != rfw)) : false;
  if (c) return true;
  if (orphan) {
    if (exp.strand == S_SA) return (isLeft && lfw) || (!isLeft && !rfw);
    if (exp.strand == S_AS) return (isLeft && !lfw) || (!isLeft && rfw);
    return false;
  }
  if (exp.strand == S_SA) return lfw && !rfw;
  if (exp.strand == S_AS) return !lfw && rfw;
  return false;
}

// ---- a7/a8 — updateRefMappings + filterAndCollectAlignments (SalmonMappingUtils.hpp:225-405) ------
// The selection itself, as a function of what the candidates' alignment left behind: pinned to the reference's own header compiled with stand-in
// pufferfish types (oracle/ref_mapping_utils_shim.cpp, tests/test_selection_pin.py).  scored[i] == 0: the candidate was skipped (incompatible under
// ignoreIncompat) or its alignment failed — updateRefMappings never sees it (SalmonQuantify.cpp:1521-1533).  The candidates are taken in index order:
// updateRefMappings compares a hit with the best decoy score seen SO FAR (:231-246), which is why the cut-off is replayed and not taken from the end.
// NOTE (a quirk of the reference that is NOT followed, SPEC section a7): at the call site the slot index `idx` is not advanced when an incompatible
// candidate is skipped (SalmonQuantify.cpp:1521-1523, 2148-2150), so the record emitted for a later candidate takes its positions from jointHits[idx],
// an EARLIER candidate.  Which candidate that is depends on pufferfish's order of jointHits, which is not in this tree; here a record is built from
// its own candidate.  It never happens for an unstranded library (every candidate pufferfish returns is compatible with IU).
struct Selection { std::vector<size_t> kept; std::vector<double> prob; int32_t bestScore = INVALID_SCORE, bestDecoy = INVALID_SCORE; bool onlyDecoy = false; };
static void select_hits(const uint32_t* tid, const int32_t* score, const uint8_t* compat, const uint8_t* scored, size_t n, uint32_t first_decoy, double decoy_threshold,
                        bool hard_filter, double score_exp, double min_aln_prob, Selection& S) {
  S.kept.clear(); S.prob.clear(); S.bestScore = INVALID_SCORE; S.bestDecoy = INVALID_SCORE; S.onlyDecoy = false;
  for (size_t i = 0; i < n; ++i) if (scored[i] && tid[i] >= first_decoy) S.bestDecoy = std::max(S.bestDecoy, score[i]);
  auto decoy_cut = [&](int32_t bd) -> int32_t { return (int32_t)(decoy_threshold * (double)bd); };
  std::vector<uint8_t> keep(n, 0);
  {
    int32_t runDecoy = INVALID_SCORE;
    std::unordered_map<uint32_t, size_t> bestPer;
    for (size_t i = 0; i < n; ++i) {
      if (!scored[i]) continue;
      if (tid[i] >= first_decoy) { runDecoy = std::max(runDecoy, score[i]); continue; }
      if (score[i] < decoy_cut(runDecoy)) continue;
      auto it = bestPer.find(tid[i]);
      if (it == bestPer.end()) { bestPer[tid[i]] = i; keep[i] = 1; }
      else if (score[i] > score[it->second] || (score[i] == score[it->second] && compat[i])) {
        keep[it->second] = 0;
        it->second = i;
        keep[i] = 1;
      }
      if (score[i] > S.bestScore) S.bestScore = score[i];
    }
  }
  S.onlyDecoy = (S.bestScore < decoy_cut(S.bestDecoy)) && (S.bestDecoy > INVALID_SCORE);  // MappingScoreInfo::haveOnlyDecoyMappings :115-122
  if (S.bestScore > INVALID_SCORE && !S.onlyDecoy) {
    int32_t bd = (S.bestDecoy == INVALID_SCORE) ? INVALID_SCORE + 1 : S.bestDecoy;  // :292-294
    int32_t thr = hard_filter ? S.bestScore : decoy_cut(bd);
    std::vector<size_t> kept; for (size_t i = 0; i < n; ++i) if (keep[i] && score[i] >= thr) kept.push_back(i);
    std::stable_sort(kept.begin(), kept.end(), [&](size_t a, size_t b) { return tid[a] < tid[b]; });
    for (size_t i : kept) {
      double v = (double)S.bestScore - (double)score[i];
      double p = hard_filter ? -1.0 : sq_exp(-score_exp * v);
      if (!hard_filter && p < min_aln_prob) continue;
      S.kept.push_back(i); S.prob.push_back(p);
    }
  }
}
struct FragResult { std::vector<sq_aln> alns; uint8_t map_type = SQ_MT_UNMAPPED; };

struct Taps { std::vector<sq_unimem> unimems; std::vector<sq_mem> mems; std::vector<sq_chain> chains; std::vector<sq_cand> cands; bool on = false; };

static void map_fragment(const Index& ix, const Opts& op, uint32_t frag, const uint8_t* s1, uint32_t n1, const uint8_t* s2, uint32_t n2,
    bool paired,
                         FragResult& out, sq_map_stats& st, Taps* taps) {
  out.alns.clear(); out.map_type = SQ_MT_UNMAPPED;
  st.num_reads++;
  // [r4] SPEC §I: read ends are mapped whole (the product takes up to 1000 bases and refuses longer reads with an error; nothing is cut)
  std::vector<uint8_t> rd[2]; std::vector<UniMem> um[2]; std::vector<Mem> mems[2]; std::vector<Chain> ch[2];
  const int nends = paired ? 2 : 1;
  for (int e = 0; e < nends; ++e) {
    encode_read(e ? s2 : s1, e ? n2 : n1, rd[e]);
    collect_unimems(ix, op, rd[e], um[e], &st);
    project_mems(ix, op, um[e], (int)rd[e].size(), mems[e]);
    st.num_mems += mems[e].size();
    chain_end(ix, op, mems[e], (int)rd[e].size(), ch[e]);
    st.num_chains += ch[e].size();
    if (taps && taps->on) {
      uint32_t eid = paired ? frag * 2 + e : frag;
      for (auto& u : um[e]) {
        sq_unimem x{};
        x.end = eid;
        x.qpos = u.qpos;
        x.len = u.len;
        x.unitig = u.unitig;
        x.uoff = u.ustart;
        x.fw = u.fw;
        taps->unimems.push_back(x);
      }
      for (auto& m : mems[e]) {
        sq_mem x{};
        x.end = eid;
        x.tid = m.tid;
        x.rpos = m.rpos;
        x.qpos = m.q;
        x.len = m.len;
        x.fw = m.fw;
        taps->mems.push_back(x);
      }
      for (auto& c : ch[e]) {
        sq_chain x{};
        x.end = eid;
        x.tid = c.tid;
        x.pos = c.pos;
        x.last_end = c.last_end;
        x.fw = c.fw;
        x.n_mems = (uint32_t)c.mems.size();
        x.score = c.score;
        taps->chains.push_back(x);
      }
    }
  }
  if (!ch[0].empty() || !ch[1].empty()) st.num_mapped_at_least_a_kmer++;
  std::vector<Cand> cands; bool dovetail = false;
  if (paired) {
    const int jr = join_pair(op, ch[0], ch[1], cands, &dovetail);
    // orphan recovery (SalmonQuantify.cpp:1343-1364): orphan-only fragments with at most maxReadOccs candidates
    if (jr == 2 && op.o.recover_orphans && cands.size() <= op.o.max_read_occs && recover_orphans(ix, op, rd, ch,
        cands)) st.num_orphans_rescued++;
  }
  else {  // joinReadsAndFilterSingle: every surviving chain is a candidate (SalmonQuantify.cpp:2108-2109)
    for (size_t a = 0; a < ch[0].size(); ++a) {
      Cand c;
      c.tid = ch[0][a].tid;
      c.lc = (int)a;
      c.rc = -1;
      c.frag_len = 0;
      c.mate_status = SQ_MS_SINGLE_END;
      c.cov = ch[0][a].score;
      cands.push_back(c);
    }
  }
  if (cands.empty() && dovetail) st.num_dovetails++;
  st.num_candidates += cands.size();
  if (!cands.empty()) st.num_with_joint_hits++;  // upperBoundHits (SalmonQuantify.cpp:1368-1370)
  const LibFmt expf{op.o.lib_type, op.o.lib_orientation, op.o.lib_strand};
  // scoring + updateRefMappings
  int32_t bestDecoy = INVALID_SCORE, bestScore = INVALID_SCORE;
  std::vector<uint8_t> compat(cands.size(), 0), scored(cands.size(), 0);
  std::vector<int32_t> score(cands.size(), INVALID_SCORE);
  for (size_t i = 0; i < cands.size(); ++i) {
    Cand& c = cands[i];
    bool orphan = c.mate_status != SQ_MS_PAIRED_END_PAIRED;
    bool lfw = c.lc >= 0 ? ch[0][c.lc].fw : false, rfw = c.rc >= 0 ? ch[1][c.rc].fw : false;
    bool isc;
    if (!paired) isc = compatible_hit_se(expf, lfw, SQ_MS_SINGLE_END);  // SalmonQuantify.cpp:2141-2150
    else isc = joint_compat(expf, orphan, c.lc >= 0, lfw, rfw);
    compat[i] = isc;
    if (!isc && op.o.ignore_incompat) continue;
    if (c.lc >= 0) c.lscore = align_chain(ix, op, ch[0][c.lc], mems[0], rd[0], &st);
    if (c.rc >= 0) c.rscore = align_chain(ix, op, ch[1][c.rc], mems[1], rd[1], &st);
    int32_t hs;
    if (!orphan) hs = (c.lscore == INVALID_SCORE || c.rscore == INVALID_SCORE) ? INVALID_SCORE : c.lscore + c.rscore;
    else hs = c.lc >= 0 ? c.lscore : c.rscore;
    if (hs == INVALID_SCORE) { st.num_mappings_filtered++; continue; }
    c.valid = true; score[i] = hs; scored[i] = 1;
  }
  if (taps &&
      taps->on) for (auto& c : cands) { sq_cand x{}; x.frag = frag; x.tid = c.tid; x.lpos = c.lc >= 0 ? ch[0][c.lc].pos : 0; x.rpos = c.rc >= 0 ? ch[1][c.rc].pos : 0;
      x.lfw = c.lc >= 0 ? ch[0][c.lc].fw : 0; x.rfw = c.rc >= 0 ? ch[1][c.rc].fw : 0; x.mate_status = c.mate_status; x.valid = c.valid; x.lscore = c.lscore; x.rscore = c.rscore; x.frag_len = c.frag_len; taps->cands.push_back(x); }
  // updateRefMappings + filterAndCollectAlignments (select_hits above)
  Selection sel;
  { std::vector<uint32_t> tids(cands.size()); for (size_t i = 0; i < cands.size(); ++i) tids[i] = cands[i].tid;
    select_hits(tids.data(), score.data(), compat.data(), scored.data(), cands.size(), ix.first_decoy, op.o.decoy_threshold, op.o.hard_filter != 0, op.o.score_exp, op.o.min_aln_prob, sel); }
  bestScore = sel.bestScore; bestDecoy = sel.bestDecoy; const bool onlyDecoy = sel.onlyDecoy;
  if (bestScore > INVALID_SCORE && !onlyDecoy) {
    for (size_t k = 0; k < sel.kept.size(); ++k) {
      const size_t i = sel.kept[k]; const Cand& c = cands[i]; const double p = sel.prob[k];
      sq_aln a{}; a.tid = c.tid; a.est_aln_prob = p; a.mate_status = paired ? c.mate_status : SQ_MS_SINGLE_END; a.frag_len = c.frag_len;
      if (c.mate_status == SQ_MS_PAIRED_END_PAIRED) {
        const Chain& l = ch[0][c.lc]; const Chain& r = ch[1][c.rc];
        a.pos = l.pos;
        a.fwd = l.fw;
        a.read_len = (uint16_t)n1;
        a.mate_pos = r.pos;
        a.mate_fwd = r.fw;
        a.mate_len = (uint16_t)n2;
        a.score = c.lscore;
        a.mate_score = c.rscore;
        int32_t e1 = a.fwd ? a.pos : a.pos + (int32_t)a.read_len, e2 = a.mate_fwd ? a.mate_pos : a.mate_pos + (int32_t)a.mate_len;
        a.format_id = format_id(hit_type_pe(e1, a.fwd, a.read_len, e2, a.mate_fwd, a.mate_len, false));  // SalmonQuantify.cpp:1770-1777
      } else {
        bool left = c.lc >= 0; const Chain& o = left ? ch[0][c.lc] : ch[1][c.rc];
        a.pos = o.pos; a.fwd = o.fw; a.read_len = (uint16_t)(left ? n1 : n2); a.score = left ? c.lscore : c.rscore;
        a.mate_pos = 0; a.mate_fwd = 1; a.mate_len = paired ? 0 : a.read_len; a.mate_score = 0;
        a.format_id = format_id(hit_type_se(a.fwd));  // :1760-1768
      }
      out.alns.push_back(a);
    }
    if (!out.alns.empty()) {
      switch (out.alns.front().mate_status) {
        case SQ_MS_PAIRED_END_PAIRED: out.map_type = SQ_MT_PAIRED_MAPPED; break;
        case SQ_MS_PAIRED_END_LEFT: out.map_type = SQ_MT_LEFT_ORPHAN; break;
        case SQ_MS_PAIRED_END_RIGHT: out.map_type = SQ_MT_RIGHT_ORPHAN; break;
        default: out.map_type = SQ_MT_SINGLE_MAPPED; break;
      }
    }
  } else if (!cands.empty()) {
    // only reached with candidates (SalmonQuantify.cpp:1631-1654): decoy or unmapped
    bool anyScored = false; for (auto s : scored) anyScored |= (s != 0);
    (void)anyScored;
    out.map_type = onlyDecoy ? SQ_MT_DECOY : SQ_MT_UNMAPPED;
    st.num_decoy_fragments += onlyDecoy ? 1 : 0; st.num_fragments_filtered++;
  }
  st.num_alignments += out.alns.size();
  if (!out.alns.empty()) st.num_mapped++;
}

// ================================================================================================
// Online model + eq-classes (processMiniBatch, SalmonQuantify.cpp:426-1023) — batch-synchronous
// restatement (SPEC §D1): every fragment of a mini-batch sees the model as of the batch start.
// ================================================================================================
struct FLD {  // FragmentLengthDistribution.cpp:23-186 (bin size 1, max 1000, kernel Binomial(4, 0.5))
  std::vector<double> hist;
  double totMass = SQ_LOG_0;
  double kernel[5];
  bool cached = false;
  std::vector<double> cpmf, ccmf;
  uint32_t minLen = 1000;
  static double phi(double x) { return 0.5 * std::erfc(-x * 0.70710678118654752440); }
  void init(double mu, double sd) {
    hist.assign(1001, SQ_LOG_0);
    for (int i = 0; i <= 1000; ++i) {
      // boost::math::cdf(normal(mu, sd), i+.5) - cdf(i-.5): the prior is a table of constants;
      // the product receives the same table from the host, so libm's erfc is fine here.
      double nm = phi((i + 0.5 - mu) / sd) - phi((i - 0.5 - mu) / sd);
      double mass = SQ_LOG_EPSILON; if (nm != 0) mass = 0.0 /*log(alpha=1)*/ + sq_log(nm);
      hist[i] = mass;
    }
    const double pk[5] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
    for (int i = 0; i < 5; ++i) kernel[i] = sq_log(pk[i]);
    totMass = tree_total();
  }
  double tree_total() const {  // SPEC §D3: strided-halving logAdd tree over 1024 leaves
    std::vector<double> v(1024, SQ_LOG_0); for (int i = 0; i <= 1000; ++i) v[i] = hist[i];
    for (int s = 512; s >= 1; s >>= 1) for (int i = 0; i < s; ++i) v[i] = sq_log_add(v[i], v[i + s]);
    return v[0];
  }
  double pmf(size_t len) const {
    if (cached) return len < cpmf.size() ? cpmf[len] : cpmf.back();
    if (len > 1000) len = 1000;
    return hist[len] - totMass;
  }
  double cmf(size_t len) const { return len < ccmf.size() ? ccmf[len] : ccmf.back(); }  // only used once cached
  void apply_counts(const std::vector<uint32_t>& cnt, double logFM) {  // batched addVal (:85-110)
    for (int b = 1; b <= 1000; ++b)
      for (int i = 4; i >= 0; --i) {
        int len = b + 2 - i;
        if (len < 0 || len > 1000 || cnt[len] == 0) continue;
        hist[b] = sq_log_add(hist[b], logFM + kernel[i] + sq_log((double)cnt[len]));
      }
    totMass = tree_total();
  }
  void add_val(uint32_t len, double logFM) {  // addVal (:85-110) for ONE fragment (reference-order mode, SPEC D1r): the kernel's five bins around len, bin 0 kept empty
    for (int i = 0; i < 5; ++i) { const int b = (int)len - 2 + i; if (b > 0 && b <= 1000) hist[b] = sq_log_add(hist[b], logFM + kernel[i]); }
    totMass = tree_total();
  }
  void cache() {  // cacheCMF (:174-186) + getLockedPMF (:159-172)
    cpmf.resize(1001); double tot = SQ_LOG_0;
    for (int i = 0; i <= 1000; ++i) { cpmf[i] = hist[i] - totMass; tot = sq_log_add(tot, cpmf[i]); }
    for (int i = 0; i <= 1000; ++i) cpmf[i] -= tot;
    ccmf.resize(1001); double cum = SQ_LOG_0; for (int i = 0; i <= 1000; ++i) { cum = sq_log_add(cum, cpmf[i]); ccmf[i] = cum; }
    cached = true;
  }
};

// ---- fragment-GC bias (row f-3, --gcBias): own restatement, G/C counted base by base ------------------------------------------
static inline bool is_gc(const Index& ix, uint32_t t, int32_t i) { uint32_t b = base_at(ix.refseq.data(), ix.ref_accum[t] + (uint64_t)i); return b == 1 || b == 2; }
static inline int64_t gc_upto(const Index& ix, uint32_t t, int32_t i) { int64_t c = 0; for (int32_t x = 0; x <= i; ++x) c += is_gc(ix, t, x) ? 1 : 0; return c; }   // GCCount_[i]
// Transcript::gcDesc (Transcript.hpp:294-341)
static bool gc_desc(const Index& ix, uint32_t t, int32_t s, int32_t e, int32_t* fragFrac, int32_t* ctxFrac) {
  const int32_t RefLength = (int32_t)ix.ref_len[t]; const int lastPos = RefLength - 1;
  auto G = [&](int32_t i) { return gc_upto(ix, t, i); };
  int64_t cs = (s > 0) ? G(s - 1) : 0, ce = G(e);
  int fs = s - 4, fe = s + 1, ts = e - 2, te = e + 3;
  bool fpLeftExists = fs >= 0, fpRightExists = fe <= lastPos, tpLeftExists = ts >= 0, tpRightExists = te <= lastPos;
  int64_t fps = fpLeftExists ? G(fs) : 0, fpe = fpRightExists ? G(fe) : ce, tps = tpLeftExists ? G(ts) : 0, tpe = tpRightExists ? G(te) : ce;
  fs = fs < 0 ? 0 : fs; fe = fe > lastPos ? lastPos : fe; ts = ts < 0 ? 0 : ts; te = te > lastPos ? lastPos : te;
  int fpContextSize = !fpLeftExists ? (fe + 1) : (fe - fs), tpContextSize = !tpLeftExists ? (te + 1) : (te - ts);
  double contextSize = (double)(fpContextSize + tpContextSize);
  if (contextSize == 0) return false;
  *fragFrac = (int32_t)std::lrint((100.0 * (double)(ce - cs)) / (double)(e - s + 1));
  *ctxFrac = (int32_t)std::lrint(100.0 * ((double)((fpe - fps) + (tpe - tps)) / contextSize));
  return true;
}
static inline int32_t gc_frag_bin(int32_t f) { double w = 100.0 / 25; return std::min(24, (int32_t)((double)f / w)); }   // GCDesc::fragBin(25)
static inline int32_t gc_ctx_bin(int32_t f) { double w = 100.0 / 3; return std::min(2, (int32_t)((double)f / w)); }       // GCDesc::contextBin(3)

static void length_classes(const Index& ix, std::vector<uint32_t>& quant, std::vector<uint8_t>& cls);
struct EqVal { uint64_t count = 0; std::vector<uint64_t> wq; };
// ---- the CIGAR-based alignment error model of alignment-based input (row f4; src/alignment/AlignmentModel.cpp, include/.../AlignmentModel.hpp,
// AlignmentCommon.cpp:39-79 setBasesFromCIGAROp_, AtomicMatrix.hpp) -------------------------------------------------------------------------
// A first-order Markov chain over alignment columns: state = refSymbol * 9 + readSymbol (A C G T, '-' 4, soft clip 5, hard clip 6, pad 7, reference skip 8),
// 82 x 82 transition weights per read-position bin, one set for the read scored "left" and one for the read scored "right", kept in log space with
// their row sums (AtomicMatrix: every cell starts at log(alpha = 1), every row sum at log(82)).  logLikelihood walks a record's CIGAR and adds the
// log transition probabilities (foreground) and, per column, the bin's (0, 0) entry (background); update adds `logForgettingMass + p` to the cells the
// walk visits.  The two walks are restated separately because they differ (:190-191 resets both "advance" flags at the start of a CIGAR operation,
// :354 only one; :197-212 returns what it has when a CIGAR runs past the read, :358-373 stops updating).
// SPEC D1 applies: a group of mini-batches reads the matrices as of the group's start; the increments of a mini-batch are collected per cell as
// sums of exp(p) in fixed point (p = 0 unless the aligner is bowtie2: plain counts) and applied at the group's end in mini-batch order:
// cell = logAdd(cell, logFM_b + log(sum)), and the row sum likewise with the row's sum.
struct ErrModel {
  static constexpr uint32_t NS = 82, START = 81;
  uint32_t bins = 0; std::vector<double> cell[2], row[2];
  void init(uint32_t b) { bins = b; for (int s = 0; s < 2; ++s) { cell[s].assign((size_t)b * NS * NS, sq_log(1.0)); row[s].assign((size_t)b * NS, sq_log((double)NS * 1.0)); } }
  double tp(int side, uint32_t bin, uint32_t prev, uint32_t cur) const { return cell[side][((size_t)bin * NS + prev) * NS + cur] - row[side][(size_t)bin * NS + prev]; }
};
static inline int cig_type(uint32_t op) { return op < 9 ? (int)((0x3C1A7u >> (op << 1)) & 3u) : 0; }   // htslib bam_cigar_type: bit 0 consumes the read, bit 1 the reference
static inline void cig_states(uint32_t op, uint32_t& refB, uint32_t& readB) {   // setBasesFromCIGAROp_ (AlignmentCommon.cpp:39-79)
  switch (op) { case 1: refB = 4; break; case 2: readB = 4; break; case 3: readB = 8; break; case 4: refB = 5; break; case 5: refB = 6; readB = 6; break; case 6: refB = 7; readB = 7; break; default: break; }
}
struct ErrRec { int32_t pos; const uint32_t* cig; uint32_t ncig; const uint8_t* seq; int32_t len; };
// AlignmentModel::logLikelihood(bam_seq_t*, ...) (:98-244): fg and bg of one record against transcript t
static void err_like_rec(const ErrModel& E, int side, const Index& ix, uint32_t t, const ErrRec& R, double* fg, double* bg) {
  size_t readIdx = 0; int64_t tIdx = R.pos; const size_t tLen = ix.ref_len[t];
  if (tIdx < 0) { readIdx = (size_t)(-tIdx); tIdx = 0; }
  size_t uT = (size_t)tIdx;
  if (uT >= tLen) { *fg = SQ_LOG_0; *bg = 0.0; return; }
  if (R.ncig == 0) { *fg = SQ_LOG_EPSILON; *bg = 0.0; return; }
  if (R.len <= 0) { *fg = 0.0; *bg = 0.0; return; }   // no sequence to be had for this record (SPEC: it says nothing)
  double ll = 0.0, bl = 0.0; uint32_t bin = 0, prev = ErrModel::START; const double invLen = (double)E.bins / (double)R.len;
  auto rb = [&](size_t i) -> uint32_t { return i < (size_t)R.len ? R.seq[i] : 0u; };
  auto tb = [&](size_t i) -> uint32_t { return i < tLen ? base_at(ix.refseq.data(), ix.ref_accum[t] + i) : 0u; };
  for (uint32_t ci = 0; ci < R.ncig; ++ci) {
    const uint32_t opLen = R.cig[ci] >> 4, op = R.cig[ci] & 15u; const int ty = cig_type(op);
    uint32_t curRead = (ty & 1) ? rb(readIdx) : 0u, curRef = (ty & 2) ? tb(uT) : 0u;
    bool advRead = false, advRef = false;
    for (uint32_t i = 0; i < opLen; ++i) {
      if (advRead) { if (readIdx >= (size_t)R.len) { *fg = ll; *bg = bl; return; } curRead = rb(readIdx); bin = (uint32_t)((double)readIdx * invLen); advRead = false; }
      if (advRef) { if (uT >= tLen) { *fg = ll; *bg = bl; return; } curRef = tb(uT); advRef = false; }
      cig_states(op, curRef, curRead);
      const uint32_t cur = curRef * 9 + curRead;
      ll += E.tp(side, bin, prev, cur); bl += E.tp(side, bin, 0, 0); prev = cur;
      if (ty & 1) { ++readIdx; advRead = true; }
      if (ty & 2) { ++uT; advRef = true; }
    }
  }
  *fg = ll; *bg = bl;
}
// AlignmentModel::update(bam_seq_t*, ...) (:308-424): the cells one record's walk visits, as (bin * 82 + prev) * 82 + cur
static void err_update_rec(uint32_t bins, const Index& ix, uint32_t t, const ErrRec& R, std::vector<uint32_t>& cells) {
  int32_t readIdx = 0; int64_t tIdx = R.pos; const size_t tLen = ix.ref_len[t];
  if (tIdx < 0) { readIdx = (int32_t)(-tIdx); tIdx = 0; }
  size_t uT = (size_t)tIdx;
  if (uT >= tLen || R.ncig == 0 || R.len <= 0) return;
  bool advRead = false, advRef = false; uint32_t bin = 0, prev = ErrModel::START; const double invLen = (double)bins / (double)R.len;
  auto rb = [&](int32_t i) -> uint32_t { return (i >= 0 && i < R.len) ? R.seq[i] : 0u; };
  auto tb = [&](size_t i) -> uint32_t { return i < tLen ? base_at(ix.refseq.data(), ix.ref_accum[t] + i) : 0u; };
  for (uint32_t ci = 0; ci < R.ncig; ++ci) {
    const uint32_t opLen = R.cig[ci] >> 4, op = R.cig[ci] & 15u; const int ty = cig_type(op);
    uint32_t curRead = (ty & 1) ? rb(readIdx) : 0u, curRef = (ty & 2) ? tb(uT) : 0u;
    advRef = false;                                   // (:354: the read's flag keeps its value across operations)
    for (uint32_t i = 0; i < opLen; ++i) {
      if (advRead) { if (readIdx >= R.len) return; curRead = rb(readIdx); bin = (uint32_t)((double)readIdx * invLen); advRead = false; }
      if (advRef) { if (uT >= tLen) return; curRef = tb(uT); advRef = false; }
      cig_states(op, curRef, curRead);
      const uint32_t cur = curRef * 9 + curRead;
      cells.push_back((bin * ErrModel::NS + prev) * ErrModel::NS + cur);
      prev = cur;
      if (ty & 1) { ++readIdx; advRead = true; }
      if (ty & 2) { ++uT; advRef = true; }
    }
  }
}

struct QuantState {
  const Index* ix; Opts op;
  FLD fld; std::vector<double> ambigCMF;  // LogCMFCache pre-burn-in table (DistributionUtils.cpp:104-118)
  std::vector<double> liveCMF;            // FLD::cmf(len) before cacheCMF (FragmentLengthDistribution.cpp:143-158); read by single-end libraries only, whose histogram stays the prior
  std::vector<double> mass, priorMass, logEffLen; std::vector<uint64_t> uniq, total, massAcc;
  std::vector<double> fm;  // forgetting masses per mini-batch
  uint64_t numObserved = 0, numAssigned = 0, numMappedUB = 0, batchNo = 0, numCompat = 0; bool burnedIn = false;
  std::map<std::vector<uint32_t>, EqVal> eq;  // label (tids + bins) -> value
  std::vector<uint64_t> libCounts;
  uint64_t gcObs[75] = {0};   // observedGCMass (SalmonQuantify.cpp:938-972): sums of the normalised alignment probabilities, fixed point 2^-32 (order-free)
  uint64_t readCounter = 0;
  // SPEC D1r — reference-order mode (pin of row a10 only, tests/test_minibatch_pin.py): one mini-batch in flight AND every fragment's increments applied before the next
  // fragment reads the model (masses by logAdd in alignment order, FLD one fragment at a time), exactly the order of SalmonQuantify.cpp:547-1003 on one thread; the
  // uniform draws come from the caller (the reference's engine), one per kept alignment whether burned in or not (:974)
  bool refOrder = false; const double* draws = nullptr; uint64_t drawPos = 0, drawCap = 0;
  std::vector<int32_t> condMeans;   // conditional fragment-length means of the PRIOR distribution (ReadExperiment.inl:25-43), used by single-end --gcBias
  uint64_t posObs[2][100] = {{0}};   // --posBias (SPEC §P): observed read-start masses [5' model, 3' model][length class x 20 bins], fixed point 2^-32
  std::vector<uint8_t> lenClass;     // Transcript::lengthClassIndex (ReadExperiment.inl:352-388)
  uint64_t seqObs[2][576] = {{0}}; uint64_t seqSamples = 0;   // observed read-start context counts (SBModel cells [position][context]; FW, RC) and fragments sampled so far (SPEC §B2)
  // SPEC §D1: up to W = mini_batches_in_flight consecutive mini-batches read one model snapshot (the reference's numThreads workers
  // read a shared, slightly stale model: SalmonQuantify.cpp:2390-2403); their increments wait here and are applied in order
  struct PendingMB { double logFM; std::vector<std::pair<uint32_t, uint64_t>> massInc; std::vector<uint32_t> fldCnt; bool anyFld; uint32_t minLen;
    std::vector<std::pair<uint32_t, uint64_t>> errInc[2]; };   // [r5] (cell, fixed-point sum of exp(p)) per side
  std::vector<PendingMB> pending;
  // [r5] alignment-based input with the CIGAR error model: the matrices, this mini-batch's increments, the reads of the batch being accumulated
  ErrModel em; bool errOn = false; std::vector<uint64_t> errAcc[2]; const sq_aln_reads* reads = nullptr;
  // `-l A` (SPEC §D8): LibraryTypeDetector restated at mini-batch granularity
  bool detectActive = false, detected = false; uint64_t detCounts[64] = {0}; uint64_t detSamples = 0;
  void detect_format() {  // mostLikelyType, LibraryTypeDetector.hpp:33-152
    sq_quant_opts& o = op.o;
    if (o.lib_type == T_SE) {
      uint64_t nf = 0, nr = 0;
      for (int i = 0; i < 64; ++i) { int st = i >> 3; if (st == S_S) nf += detCounts[i]; if (st == S_A) nr += detCounts[i]; }
      double ratio = (nf + nr > 0) ? (double)nf / (double)(nf + nr) : -1.0;
      o.lib_orientation = O_NONE;
      if (ratio < 0.0) o.lib_strand = S_U; else if (ratio < 0.3) o.lib_strand = S_A; else if (ratio < 0.7) o.lib_strand = S_U; else o.lib_strand = S_S;
    } else {
      uint64_t nsf = 0, nsr = 0, nin = 0, nout = 0, nsame = 0;
      for (int i = 0; i < 64; ++i) {
        int orient = (i >> 1) & 3, st = i >> 3; uint64_t c = detCounts[i];
        if (st == S_S || st == S_SA) nsf += c;
        if (st == S_A || st == S_AS) nsr += c;
        if (orient == O_TOWARD) nin += c;
        if (orient == O_AWAY) nout += c;
        if (orient == O_SAME) nsame += c;
      }
      if (nin + nout + nsame > 0 && nsf + nsr > 0) {
        uint64_t no = nin + nout + nsame;
        double rin = (double)nin / (double)no, rout = (double)nout / (double)no, rsame = (double)nsame / (double)no; bool same = false;
        if (rin >= rout && rin >= rsame) o.lib_orientation = O_TOWARD; else if (rout >= rin && rout >= rsame) o.lib_orientation = O_AWAY; else { o.lib_orientation = O_SAME; same = true; }
        double rfw = (double)nsf / (double)(nsf + nsr);
        if (rfw < 0.3) o.lib_strand = same ? S_A : S_AS; else if (rfw < 0.7) o.lib_strand = S_U; else o.lib_strand = same ? S_S : S_SA;
      } else { o.lib_orientation = O_TOWARD; o.lib_strand = S_U; }
    }
    detectActive = false; detected = true;
  }
  void flush_pending() {
    for (PendingMB& p : pending) {
      for (auto& tq : p.massInc) mass[tq.first] = sq_log_add(mass[tq.first], p.logFM + sq_log(sq_from_fixed(tq.second, SQ_MFRAC_BITS)));
      if (p.anyFld) { fld.apply_counts(p.fldCnt, p.logFM); fld.minLen = std::min(fld.minLen, p.minLen); }
      for (int sd = 0; sd < 2; ++sd) {   // AtomicMatrix::increment: the cell and its row sum (SPEC D1: per mini-batch, in cell order)
        std::map<uint32_t, uint64_t> rowsum;
        for (auto& cq : p.errInc[sd]) { em.cell[sd][cq.first] = sq_log_add(em.cell[sd][cq.first], p.logFM + sq_log(sq_from_fixed(cq.second, SQ_MFRAC_BITS))); rowsum[cq.first / ErrModel::NS] += cq.second; }
        for (auto& rq : rowsum) em.row[sd][rq.first] = sq_log_add(em.row[sd][rq.first], p.logFM + sq_log(sq_from_fixed(rq.second, SQ_MFRAC_BITS)));
      }
    }
    pending.clear();
  }
  void init(const Index* i, const sq_quant_opts* o) {
    ix = i; make_opts(o, op);
    size_t M = ix->names.size();
    fld.init(o->fld_mean, o->fld_sd);
    ambigCMF.resize(1001);
    {
      double cum = SQ_LOG_0;
      for (int j = 0; j <= 1000; ++j) {
        cum = sq_log_add(cum, SQ_LOG_EPSILON);
        ambigCMF[j] = cum;
      }
    }
    liveCMF.resize(1001);
    {
      double cum = SQ_LOG_0;
      for (int j = 0; j <= 1000; ++j) {
        cum = sq_log_add(cum, fld.hist[j]);
        liveCMF[j] = cum - fld.totMass;
      }
    }
    {   // correctionFactorsFromMass over the prior's normalised pmf (x 100), lengths 1..999
      double sum = SQ_LOG_0; for (int j = 1; j <= 1000; ++j) sum = sq_log_add(sum, fld.pmf(j));
      condMeans.assign(1001, 0); double vals = 0.0, mult = 0.0;
      for (int j = 1; j <= 1000; ++j) { const double p = j < 1000 ? 100.0 * sq_exp(fld.pmf(j) - sum) : 0.0; vals = p * (double)j + vals; mult = p + mult; condMeans[j] = (int32_t)(mult > 0 ? vals / mult : 0.0); }
    }
    { std::vector<uint32_t> q; length_classes(*ix, q, lenClass); }
    mass.assign(M, SQ_LOG_0); priorMass.resize(M); logEffLen.resize(M); uniq.assign(M, 0); total.assign(M, 0); massAcc.assign(M, 0);
    // Transcript.hpp:48-56, ReadExperiment.inl:114
    for (size_t t = 0; t < M; ++t) {
      double len = (double)ix->ref_len[t];
      priorMass[t] = sq_log(0.005 * len);
      logEffLen[t] = sq_log(len);
    }
    libCounts.assign(64, 0);
    detectActive = o->lib_autodetect != 0;
    errOn = o->error_model != 0;
    if (errOn) { em.init(o->num_error_bins ? o->num_error_bins : 6); for (int sd = 0; sd < 2; ++sd) errAcc[sd].assign(em.cell[sd].size(), 0); }
  }
  double forgetting_mass(uint64_t b) {  // ForgettingMassCalculator.hpp:30-40 (prefill recurrence)
    while (fm.size() <= b) {
      if (fm.empty()) { fm.push_back(0.0); continue; }
      uint64_t i = fm.size() + 1;  // i = 2,3,... for index 1,2,...
      double ff = op.o.forgetting_factor;
      // `fm += a - b` in prefill (:30-33): the increment is formed first, then added — not (fm + a) - b
      const double inc = ff * std::log((double)(i - 1)) - std::log(std::pow((double)i, ff) - 1.0);
      fm.push_back(fm.back() + inc);
    }
    return fm[b];
  }
  void burnin_finalize();
};

// updateTranscriptLengthsAtomic (ReadExperiment.inl:62-94) + correctionFactorsFromMass /
// computeSmoothedEffectiveLengths (DistributionUtils.cpp:9-55)
static void compute_eff_lengths(const FLD& fld, const std::vector<uint32_t>& ref_len, std::vector<double>& logEffLen) {
  size_t minV = (fld.minLen == 1000) ? 1 : fld.minLen, maxV = 1000;
  std::vector<double> lp; for (size_t i = minV; i <= maxV; ++i) lp.push_back(fld.pmf(i));
  double sum = SQ_LOG_0; for (double v : lp) sum = sq_log_add(sum, v);
  for (double& v : lp) v -= sum;
  std::vector<double> pmf(maxV + 1, 0.0);
  for (size_t i = minV; i < maxV; ++i) pmf[i] = 100.0 * sq_exp(lp[i - minV]);
  size_t n = pmf.size(); std::vector<double> cf(n, 0.0), vals(n, 0.0), mult(n, 0.0);
  mult[0] = pmf[0];
  for (size_t i = 1; i < n; ++i) {
    double v = pmf[i];
    vals[i] = v * (double)i + vals[i - 1];
    mult[i] = v + mult[i - 1];
    if (mult[i] > 0) cf[i] = vals[i] / mult[i];
  }
  for (size_t t = 0; t < ref_len.size(); ++t) {
    double ol = (double)ref_len[t]; double c = (ol >= (double)n) ? cf[n - 1] : cf[ref_len[t]];
    double el = ol - c; if (el < 1.0) el = ol;
    logEffLen[t] = sq_log(el);
  }
}
void QuantState::burnin_finalize() { compute_eff_lengths(fld, ix->ref_len, logEffLen); fld.cache(); burnedIn = true; }

// LogCMFCache::getAmbigFragLengthProb (DistributionUtils.cpp:151-172): the probability that a fragment whose one read lies here is no longer than
// the transcript lets it be.  liveCMF = FLD::cmf before cacheCMF; ambigCMF = LogCMFCache's own table (evaluateLogCMF, :104-118)
static double ambig_frag_prob(const FLD& fld, const std::vector<double>& liveCMF, const std::vector<double>& ambigCMF, bool singleEnd, bool burned, bool fwd,
                              int32_t pos, int32_t rlen, int32_t tl) {
  int32_t maxFL;
  if (fwd) { int32_t p1 = pos < 0 ? 0 : pos; p1 = p1 > tl ? tl : p1; maxFL = tl - p1; }
  else { int32_t p1 = pos + rlen; p1 = p1 < 0 ? 0 : p1; p1 = p1 > tl ? tl : p1; maxFL = p1; }
  const bool useFLD = singleEnd || burned;
  auto cmfv = [&](size_t len) -> double { if (useFLD) return fld.cached ? fld.cmf(len) : liveCMF[std::min<size_t>(len, 1000)]; return len < 1001 ? ambigCMF[len] : ambigCMF[1000]; };
  const double refCM = cmfv((size_t)tl); const bool cm = !(refCM == SQ_LOG_0);
  return cm ? (cmfv((size_t)maxFL) - refCM) : SQ_LOG_EPSILON;
}
static inline double u01(uint64_t seed, uint64_t read, uint64_t aln) {
  uint64_t x = sq_mix64(seed ^ sq_mix64(read * 0x9E3779B97F4A7C15ULL + aln + 1));
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
static inline uint32_t frag_len_pedantic(const sq_aln& a, uint32_t txpLen) {  // ReadPair.hpp:149-168 semantics
  if (a.mate_status != SQ_MS_PAIRED_END_PAIRED || a.fwd == a.mate_fwd) return 0;
  int32_t T = (int32_t)txpLen;
  int32_t p1 = a.fwd ? a.pos : a.mate_pos; p1 = p1 < 0 ? 0 : p1; p1 = p1 > T ? T : p1;
  int32_t p2 = a.fwd ? a.mate_pos + (int32_t)a.mate_len : a.pos + (int32_t)a.read_len; p2 = p2 < 0 ? 0 : p2; p2 = p2 > T ? T : p2;
  return (uint32_t)(p1 > p2 ? p1 - p2 : p2 - p1);
}

// one mini-batch [r0, r1) of a CSR alignment batch
// ---- sequence-specific bias: the read-start context model (SBModel, src/model/SBModel.cpp) — SPEC §B2 --------------------------------
// A context is the 9 bases from 3 before a read's first base to 5 after it, as an 18-bit code with the first base in the high bits;
// position i of the model conditions on the SB_ORDER[i] bases before it: cell = (code >> (18 - 2 (i + 1))) & (4^(order + 1) - 1).
static const int SB_K = 9, SB_LEFT = 3, SB_RIGHT = 5;
static const int SB_ORDER[9] = {0, 1, 2, 2, 2, 2, 2, 2, 2};
static inline uint32_t sb_ctx(const Index& ix, uint32_t t, int32_t p) { uint32_t v = 0; for (int i = 0; i < SB_K; ++i) v = (v << 2) | base_at(ix.refseq.data(), ix.ref_accum[t] + (uint64_t)(p + i)); return v; }
static inline uint32_t sb_rc(uint32_t v) { uint32_t r = 0; for (int i = 0; i < SB_K; ++i) { r = (r << 2) | (3u - (v & 3u)); v >>= 2; } return r; }
static inline uint32_t sb_cell(uint32_t v, int i) { const int shift = 2 * SB_K - 2 * (i + 1), width = 2 * (SB_ORDER[i] + 1); return (uint32_t)i * 64u + ((v >> shift) & ((1u << width) - 1u)); }
// SalmonQuantify.cpp:1668-1747: one alignment of a paired-end fragment is drawn (index uniform over 0..n, n = "none"); if it is a proper
// pair on opposite strands whose two read starts have their whole contexts inside the transcript and the forward read starts before the
// reverse one, the forward read's context goes to the FW model and the reverse read's context, reverse-complemented, to the RC model —
// for the first num_bias_samples such fragments in read order (the reference counts down a shared counter in arrival order and draws
// from random_device; here the draw is u01(seed ^ C, read index) and the cap is applied in read order).
static void seq_observe(QuantState& S, uint32_t n, const uint64_t* off, const sq_aln* alns) {
  const sq_quant_opts& o = S.op.o; const Index& ix = *S.ix;
  if (!o.seq_bias) return;
  for (uint32_t r = 0; r < n; ++r) {
    if (S.seqSamples >= o.num_bias_samples) break;
    const uint64_t a0 = off[r], a1 = off[r + 1]; const uint32_t na = (uint32_t)(a1 - a0);
    if (!na) continue;
    const uint64_t x = sq_mix64((o.seed ^ 0x5EB1A5ULL) ^ sq_mix64((S.readCounter + r) * 0x9E3779B97F4A7C15ULL + 1));
    const uint32_t pick = (uint32_t)sq_mulhi64(x, (uint64_t)na + 1);
    if (pick >= na) continue;
    const sq_aln& h = alns[a0 + pick];
    if (o.lib_type != T_PE) {   // single-end library (:2211-2257): one context; the start of a reverse read is pos + readLen there
      const int32_t RL = (int32_t)ix.ref_len[h.tid]; const int32_t sp = h.fwd ? h.pos : h.pos + (int32_t)h.read_len; const bool rc = !h.fwd;
      const int32_t b = rc ? SB_RIGHT : SB_LEFT, a = rc ? SB_LEFT : SB_RIGHT;
      if (!(sp > 0 && sp < RL && sp >= b && sp + a < RL)) continue;
      uint32_t c = sb_ctx(ix, h.tid, sp - b); if (rc) c = sb_rc(c);
      for (int i = 0; i < SB_K; ++i) S.seqObs[h.fwd ? 0 : 1][sb_cell(c, i)]++;
      S.seqSamples++; continue;
    }
    if (h.mate_status != SQ_MS_PAIRED_END_PAIRED || h.fwd == h.mate_fwd) continue;
    const int32_t RL = (int32_t)ix.ref_len[h.tid];
    const int32_t s1 = h.fwd ? h.pos : h.pos + (int32_t)h.read_len - 1, s2 = h.mate_fwd ? h.mate_pos : h.mate_pos + (int32_t)h.mate_len - 1;
    if (!(s1 > 0 && s1 < RL && s2 > 0 && s2 < RL)) continue;
    const bool rc1 = !h.fwd, rc2 = !h.mate_fwd;
    const int32_t b1 = rc1 ? SB_RIGHT : SB_LEFT, a1c = rc1 ? SB_LEFT : SB_RIGHT, b2 = rc2 ? SB_RIGHT : SB_LEFT, a2c = rc2 ? SB_LEFT : SB_RIGHT;
    if (!(s1 >= b1 && s1 + a1c < RL && s2 >= b2 && s2 + a2c < RL)) continue;
    const int32_t fwPos = h.fwd ? s1 : s2, rcPos = h.fwd ? s2 : s1;
    if (!(fwPos < rcPos)) continue;
    uint32_t left = sb_ctx(ix, h.tid, s1 - b1), right = sb_ctx(ix, h.tid, s2 - b2);
    if (rc1) left = sb_rc(left); else right = sb_rc(right);
    for (int i = 0; i < SB_K; ++i) { S.seqObs[h.fwd ? 0 : 1][sb_cell(left, i)]++; S.seqObs[h.mate_fwd ? 0 : 1][sb_cell(right, i)]++; }
    S.seqSamples++;
  }
}
// SBModel::normalize (:216-252): counts (+ the 1e-10 prior every cell starts with) -> conditional probabilities per context -> logs
static void sb_normalize(const double* counts, double* logp) {
  for (int i = 0; i < SB_K; ++i) {
    const int nstates = 1 << (2 * SB_ORDER[i]);
    for (int c = 0; c < 64; ++c) logp[i * 64 + c] = 0.0;
    for (int g = 0; g < nstates; ++g) {
      const double* q = counts + i * 64 + 4 * g;
      const double tot = ((q[0] + q[1]) + q[2]) + q[3];
      for (int b = 0; b < 4; ++b) { const double pr = q[b] / tot; logp[i * 64 + 4 * g + b] = pr > 0.0 ? sq_log(pr) : sq_log(1e-5); }
    }
  }
}
static inline double sb_eval(const double* logp, uint32_t v) { double p = 0.0; for (int i = 0; i < SB_K; ++i) p += logp[sb_cell(v, i)]; return p; }
// ---- --posBias (SPEC §P) ------------------------------------------------------------------------------------------------------
// setTranscriptLengthClasses_ (ReadExperiment.inl:352-388): quantiles of the non-decoy lengths (:152, only those are collected), the
// class of a transcript = the number of quantiles <= its length, capped at the last class
static void length_classes(const Index& ix, std::vector<uint32_t>& quant, std::vector<uint8_t>& cls) {
  const size_t M = ix.ref_len.size(), n = std::min<size_t>(ix.first_decoy ? ix.first_decoy : M, M), nbins = 5;
  std::vector<uint32_t> len(ix.ref_len.begin(), ix.ref_len.begin() + n); std::sort(len.begin(), len.end());
  quant.clear();
  if (n > nbins) { const size_t step = n / nbins; size_t cum = 0; for (size_t i = 0; i < nbins; ++i) { cum += step; quant.push_back(len[std::min(cum, n - 1)]); } }
  else quant = len;
  cls.assign(M, 0);
  if (quant.empty()) return;
  const long maxQ = (long)quant.size() - 1;
  for (size_t t = 0; t < M; ++t) cls[t] = (uint8_t)std::min<long>(maxQ, std::upper_bound(quant.begin(), quant.end(), ix.ref_len[t]) - quant.begin());
}
// SimplePosBias::addMass(pos, length, mass) (SimplePosBias.cpp:20-28): the bin of a read start
static inline int pos_bin(int32_t pos, uint32_t length) {
  const double step = (double)length / 20.0; int b = (int)std::floor((double)pos / step); return b > 19 ? 19 : b;   // > 19 is out of the reference's array; unreachable for pos < length
}
static inline int32_t pos_clamp(int32_t p, uint32_t rl) { if (p < 0) p = 0; if (p >= (int32_t)rl) p = (int32_t)rl - 1; return p; }
// SimplePosBias::finalize (SimplePosBias.cpp:51-83) on linear masses + tk::spline's constructor (vendor/upstream/misc/spline.h:264-376:
// natural cubic spline, band-matrix LU with the rows scaled to a unit diagonal first), operation by operation
struct PosSpline { double x[22], y[22], a[22], b[22], c[22]; };
static void pos_spline_build(const double* xs, const double* ys, int n, PosSpline& S) {
  std::vector<double> lo(n, 0.0), di(n, 0.0), up(n, 0.0), rhs(n, 0.0), sd(n, 0.0);   // A(i,i-1), A(i,i), A(i,i+1)
  for (int i = 0; i < n; ++i) { S.x[i] = xs[i]; S.y[i] = ys[i]; }
  for (int i = 1; i < n - 1; ++i) {
    lo[i] = 1.0 / 3.0 * (xs[i] - xs[i - 1]); di[i] = 2.0 / 3.0 * (xs[i + 1] - xs[i - 1]); up[i] = 1.0 / 3.0 * (xs[i + 1] - xs[i]);
    rhs[i] = (ys[i + 1] - ys[i]) / (xs[i + 1] - xs[i]) - (ys[i] - ys[i - 1]) / (xs[i] - xs[i - 1]);
  }
  di[0] = 2.0; up[0] = 0.0; rhs[0] = 0.0; di[n - 1] = 2.0; lo[n - 1] = 0.0; rhs[n - 1] = 0.0;
  for (int i = 0; i < n; ++i) { sd[i] = 1.0 / di[i]; if (i > 0) lo[i] *= sd[i]; di[i] *= sd[i]; if (i < n - 1) up[i] *= sd[i]; di[i] = 1.0; }   // preconditioning
  for (int k = 0; k < n - 1; ++k) { const double x = -lo[k + 1] / di[k]; lo[k + 1] = -x; di[k + 1] = di[k + 1] + x * up[k]; }                   // LU
  std::vector<double> y(n), b(n);
  for (int i = 0; i < n; ++i) { double sum = 0; if (i > 0) sum += lo[i] * y[i - 1]; y[i] = (rhs[i] * sd[i]) - sum; }                             // L y = rhs
  for (int i = n - 1; i >= 0; --i) { double sum = 0; if (i < n - 1) sum += up[i] * b[i + 1]; b[i] = (y[i] - sum) / di[i]; }                      // R b = y
  for (int i = 0; i < n; ++i) { S.b[i] = b[i]; S.a[i] = 0.0; S.c[i] = 0.0; }
  for (int i = 0; i < n - 1; ++i) {
    S.a[i] = 1.0 / 3.0 * (b[i + 1] - b[i]) / (xs[i + 1] - xs[i]);
    S.c[i] = (ys[i + 1] - ys[i]) / (xs[i + 1] - xs[i]) - 1.0 / 3.0 * (2.0 * b[i] + b[i + 1]) * (xs[i + 1] - xs[i]);
  }
  const double h = xs[n - 1] - xs[n - 2]; S.a[n - 1] = 0.0; S.c[n - 1] = 3.0 * S.a[n - 2] * h * h + 2.0 * S.b[n - 2] * h + S.c[n - 2];
}
static inline double pos_spline_eval(const PosSpline& S, int n, double x) {   // spline::operator() for x in [x0, x_{n-1}] (:385-402)
  int idx = (int)(std::lower_bound(S.x, S.x + n, x) - S.x); idx = idx == 0 ? 0 : idx - 1;
  const double h = x - S.x[idx];
  return ((S.a[idx] * h + S.b[idx]) * h + S.c[idx]) * h + S.y[idx];
}
static const double POS_BINS[20] = {.02, .04, .06, .08, .10, .15, .2, .3, .4, .5, .6, .7, .8, .85, .9, .92, .94, .96, .98, 1.0};
static void pos_finalize(const double* mass /*[20] linear*/, PosSpline& S, double* norm /*[20] or null*/) {
  double sum = 0.0; for (int i = 0; i < 20; ++i) sum += mass[i];
  double ys[22], xs[22]; const double startKnot = mass[0] / sum, stopKnot = mass[19] / sum, splineSum = sum + startKnot + stopKnot;
  ys[0] = startKnot; for (int i = 0; i < 20; ++i) { ys[i + 1] = mass[i] / splineSum; if (norm) norm[i] = mass[i] / sum; } ys[21] = stopKnot;
  xs[0] = 0.0; for (int i = 0; i < 20; ++i) xs[i + 1] = POS_BINS[i] - 0.01; xs[21] = 1.0;
  pos_spline_build(xs, ys, 22, S);
}
static inline double pos_weight(const PosSpline& S, int32_t p, int32_t len) { const double f = (double)p / (double)len; return std::max(0.001, pos_spline_eval(S, 22, f)); }   // projectWeights (:32-39)

// the lane-strided sum the device uses for sums over fragment starts (SPEC §B2): lane l of 256 adds the terms l, l + 256, ... in order,
// then the lanes combine by strided halving
template <class F> static double lane_sum256(int64_t n, F term) {
  double v[256];
  for (int l = 0; l < 256; ++l) { double a = 0.0; for (int64_t i = l; i < n; i += 256) a += term(i); v[l] = a; }
  for (int st = 128; st >= 1; st >>= 1) for (int i = 0; i < st; ++i) v[i] = v[i] + v[i + st];
  return v[0];
}

static inline ErrRec err_rec(const sq_aln_reads* R, uint64_t ai, int k) {
  const uint64_t j = 2 * ai + (uint64_t)k; ErrRec e; e.pos = R->pos[j]; e.cig = R->cigar + R->cig_off[j]; e.ncig = (uint32_t)(R->cig_off[j + 1] - R->cig_off[j]);
  e.seq = R->seq + R->seq_off[j]; e.len = (int32_t)(R->seq_off[j + 1] - R->seq_off[j]); return e;
}
// AlignmentModel::logLikelihood(const ReadPair& / const UnpairedRead&) (:246-306): foreground minus background over the alignment's records
static double err_like_aln(const QuantState& S, uint64_t ai, uint32_t t) {
  double ll = 0.0, bg = 0.0;
  for (int k = 0; k < 2; ++k) { const ErrRec R = err_rec(S.reads, ai, k); if (R.ncig == 0 && R.len == 0) continue;   // no such record (an orphan's mate, a single-end read's second slot)
    double f, b; err_like_rec(S.em, k, *S.ix, t, R, &f, &b); ll += f; bg += b; }
  return ll - bg;
}
static void process_mini_batch(QuantState& S, const uint64_t* off, const sq_aln* alns, uint64_t r0, uint64_t r1) {
  const Opts& op = S.op; const sq_quant_opts& o = op.o; const Index& ix = *S.ix;
  const double logFM = S.forgetting_mass(S.batchNo);
  const bool burned = S.burnedIn; const uint64_t assigned0 = S.numAssigned;
  const LibFmt expf{o.lib_type, o.lib_orientation, o.lib_strand};
  const bool singleEnd = (o.lib_type == T_SE);
  std::vector<uint32_t> fldCnt(1001, 0); uint32_t minLen = S.fld.minLen;
  uint64_t local = 0;
  std::vector<double> aux, lp; std::vector<uint32_t> tids; std::vector<const sq_aln*> ka;
  for (uint64_t r = r0; r < r1; ++r) {
    uint64_t a0 = off[r], a1 = off[r + 1]; uint64_t readIdx = S.readCounter + (r - r0);
    if (a1 == a0) continue;
    const bool useAux = (assigned0 + local) >= o.num_pre_burnin_frags;
    const bool cond = burned || useAux;
    aux.clear(); lp.clear(); tids.clear(); ka.clear();
    double auxDenom = SQ_LOG_0, sumProbs = SQ_LOG_0; uint64_t fmtSeen = 0; bool hasCompat = false;
    for (uint64_t ai = a0; ai < a1; ++ai) {
      const sq_aln& a = alns[ai]; uint32_t t = a.tid;
      double refLength = ix.ref_len[t] > 0 ? (double)ix.ref_len[t] : 1.0;
      double logFragCov = a.est_aln_prob > 0 ? sq_log(a.est_aln_prob) : 0.0;
      if (S.errOn) logFragCov = (useAux && S.reads) ? err_like_aln(S, ai, t) : 0.0;   // errLike (SalmonQuantifyAlignments.cpp:513-523): LOG_1 until the auxiliary models count
      double logRefLength = o.no_length_correction ? 1.0 : ((o.no_eff_length_correction || !burned) ? sq_log((double)ix.ref_len[t]) : S.logEffLen[t]);
      double tlc = sq_log_add(S.priorMass[t], S.mass[t]);  // transcript.mass(initialRound = true)
      uint32_t flen = a.frag_len;
      if (a.mate_status == SQ_MS_PAIRED_END_PAIRED && a.fwd != a.mate_fwd) flen = frag_len_pedantic(a, ix.ref_len[t]);
      double logFragProb = 0.0;
      bool unexpectedOrphan = (expf.type == T_PE && a.mate_status != SQ_MS_PAIRED_END_PAIRED);
      if (o.model_single_frag_prob && o.use_frag_len_dist && (singleEnd || unexpectedOrphan)) {
        logFragProb = ambig_frag_prob(S.fld, S.liveCMF, S.ambigCMF, singleEnd, burned, a.fwd != 0, a.pos, (int32_t)a.read_len, (int32_t)ix.ref_clen[t]);
      } else if (unexpectedOrphan) logFragProb = SQ_LOG_EPSILON;
      if (flen > 0 && o.use_frag_len_dist && cond) {
        double lenProb = S.fld.pmf(flen);
        if (burned) {
          double cm = S.fld.cmf(flen);
          bool ok = ((double)flen < refLength) && !(cm == SQ_LOG_0);
          logFragProb = ok ? (lenProb - cm) : SQ_LOG_EPSILON;
        }
        else if (useAux) logFragProb = lenProb;
      }
      LibFmt obs{(uint8_t)(a.format_id & 1), (uint8_t)((a.format_id >> 1) & 3), (uint8_t)(a.format_id >> 3)};
      bool isCompat = is_compatible(obs, expf, a.fwd, a.mate_status);
      double logCompat = isCompat ? 0.0 : o.incompat_prior;
      if (!isCompat && o.ignore_incompat) continue;
      if (isCompat) hasCompat = true;                       // hasCompatibleMapping (:767-769)
      double startPosProb = -logRefLength;
      if (a.mate_status == SQ_MS_PAIRED_END_PAIRED && !o.no_length_correction)
        startPosProb = ((double)flen <= refLength) ? -sq_log(refLength - (double)flen + 1.0) : SQ_LOG_EPSILON;
      fmtSeen |= 1ULL << a.format_id;
      double auxProb = logFragProb + logFragCov + logCompat;
      double logProb = tlc + auxProb + startPosProb;
      if (std::fabs(logProb) == SQ_LOG_0) continue;
      sumProbs = sq_log_add(sumProbs, logProb);
      tids.push_back(t); aux.push_back(auxProb); lp.push_back(logProb); ka.push_back(&a);
      auxDenom = sq_log_add(auxDenom, auxProb);
    }
    if (sumProbs == SQ_LOG_0) continue;
    ++local; if (hasCompat) ++S.numCompat;                 // numCompatibleFragments (:811-815)
    const size_t n = tids.size();
    for (size_t i = 0; i < n; ++i) aux[i] = sq_exp(aux[i] - auxDenom);
    std::vector<uint32_t> label(tids);
    if (o.range_factorization_bins > 0) {  // SalmonQuantify.cpp:845-853
      int32_t rangeCount = (int32_t)(std::sqrt((double)n) + (double)o.range_factorization_bins);
      for (size_t i = 0; i < n; ++i) label.push_back((uint32_t)(int32_t)(aux[i] * (double)rangeCount));
    }
    EqVal& ev = S.eq[label]; if (ev.wq.empty()) ev.wq.assign(n, 0);
    ev.count++; for (size_t i = 0; i < n; ++i) ev.wq[i] += sq_to_fixed(aux[i], SQ_WFRAC_BITS);
    for (size_t i = 0; i < n; ++i) {
      double nlp = lp[i] - sumProbs; double pr = sq_exp(nlp);
      if (S.refOrder) S.mass[tids[i]] = sq_log_add(S.mass[tids[i]], logFM + nlp);      // transcript.addMass(logForgettingMass + aln.logProb) (:871-872), at once
      else S.massAcc[tids[i]] += sq_to_fixed(pr, SQ_MFRAC_BITS);
      S.total[tids[i]] += 1;
      if (o.gc_bias && (ka[i]->format_id & 1u) == 1u) {   // :938-951, paired-end observation
        const sq_aln& a = *ka[i];
        int32_t start = std::min(a.pos, a.mate_pos), stop = start + (int32_t)a.frag_len - 1, ff, cf;
        if (start >= 0 && stop < (int32_t)ix.ref_len[tids[i]] && stop >= start && gc_desc(ix, tids[i], start, stop, &ff, &cf))
          S.gcObs[gc_ctx_bin(cf) * 25 + gc_frag_bin(ff)] += sq_to_fixed(pr, 32);
      } else if (o.gc_bias && o.lib_type == T_SE) {   // :952-971: a single-end library takes every fragment to have the conditional mean length
        const sq_aln& a = *ka[i]; const int32_t RL = (int32_t)ix.ref_len[tids[i]];
        const int32_t cmean = S.condMeans[RL >= 1001 ? 1000 : RL];
        const int32_t start = a.fwd ? a.pos : std::max(0, a.pos - cmean), stop = start + cmean; int32_t ff, cf;
        if (start >= 0 && stop < RL && gc_desc(ix, tids[i], start, stop, &ff, &cf)) S.gcObs[gc_ctx_bin(cf) * 25 + gc_frag_bin(ff)] += sq_to_fixed(pr, 32);
      }
      if (o.pos_bias) {   // :895-934: read starts by length class, weighted like the masses
        const sq_aln& a = *ka[i]; const uint32_t RL = ix.ref_len[tids[i]]; const int li = S.lenClass[tids[i]];
        if (a.mate_status == SQ_MS_PAIRED_END_PAIRED) {
          if (a.fwd != a.mate_fwd) {
            const int32_t pf = pos_clamp(a.fwd ? a.pos : a.mate_pos, RL), pr_ = pos_clamp(a.fwd ? a.mate_pos : a.pos, RL);
            S.posObs[0][li * 20 + pos_bin(pf, RL)] += sq_to_fixed(pr, 32); S.posObs[1][li * 20 + pos_bin(pr_, RL)] += sq_to_fixed(pr, 32);
          }
        } else S.posObs[a.fwd ? 0 : 1][li * 20 + pos_bin(pos_clamp(a.pos, RL), RL)] += sq_to_fixed(pr, 32);
      }
      double rr_ref = 0.0; if (S.refOrder) rr_ref = S.drawPos < S.drawCap ? S.draws[S.drawPos] : 2.0, ++S.drawPos;     // uni(randEng) is drawn for every kept alignment (:974)
      if (!burned) {
        double rr = S.refOrder ? rr_ref : u01(o.seed, readIdx, i);
        if (rr < pr) {
          if (S.errOn && S.reads) {   // alnMod.update(*aln, ..., alignerScore, logForgettingMass) (:860-864): exp(p) joins every cell of the walk
            const uint64_t ai = (uint64_t)(ka[i] - alns); const uint64_t q = sq_to_fixed(sq_exp((double)S.reads->aligner_score[ai]), SQ_MFRAC_BITS); std::vector<uint32_t> cells;
            for (int k = 0; k < 2; ++k) { cells.clear(); err_update_rec(S.em.bins, ix, tids[i], err_rec(S.reads, ai, k), cells); for (uint32_t c : cells) S.errAcc[k][c] += q; }
          }
          uint32_t fl = frag_len_pedantic(*ka[i], ix.ref_len[tids[i]]);
          if (fl > 0) {
            if (fl > 1000) fl = 1000;
            if (S.refOrder) { S.fld.add_val(fl, logFM); if (fl < S.fld.minLen) S.fld.minLen = fl; }
            else { fldCnt[fl]++; if (fl < minLen) minLen = fl; }
          }
        }
      }
    }
    if (n == 1) S.uniq[tids[0]] += 1;
    for (int f = 0; f < 64; ++f) if (fmtSeen >> f & 1) S.libCounts[f]++;
  }
  // mini-batch end: its increments join the group's queue; the group is applied (in mini-batch order) once W mini-batches are in
  // it, at the mini-batch that reaches numBurninFrags (:1012-1018), or at the end of the mapped batch (orc_eq_accumulate)
  QuantState::PendingMB pm; pm.logFM = logFM; pm.anyFld = false; pm.minLen = S.refOrder ? S.fld.minLen : minLen;
  for (size_t t = 0; t < S.massAcc.size(); ++t) if (S.massAcc[t]) { pm.massInc.emplace_back((uint32_t)t, S.massAcc[t]); S.massAcc[t] = 0; }
  if (S.errOn && !burned) for (int sd = 0; sd < 2; ++sd) for (size_t c = 0; c < S.errAcc[sd].size(); ++c) if (S.errAcc[sd][c]) { pm.errInc[sd].emplace_back((uint32_t)c, S.errAcc[sd][c]); S.errAcc[sd][c] = 0; }
  if (!burned) {
    for (auto c : fldCnt) pm.anyFld |= (c != 0);
    if (pm.anyFld) pm.fldCnt = fldCnt;
  }
  S.pending.push_back(std::move(pm));
  S.numAssigned += local; S.numObserved += (r1 - r0); S.readCounter += (r1 - r0); S.batchNo++;
  const bool burnNow = S.numAssigned >= o.num_burnin_frags && !S.burnedIn;
  bool detectNow = false;
  if (S.detectActive) {   // addSample (LibraryTypeDetector.hpp:155-160): every alignment whose observed format has the library's read type
    for (uint64_t ai = off[r0]; ai < off[r1]; ++ai) if ((alns[ai].format_id & 1u) == o.lib_type) { S.detCounts[alns[ai].format_id & 63u]++; S.detSamples++; }
    detectNow = S.detSamples >= 50000;
  }
  const uint32_t W = S.refOrder ? 1u : std::max(1u, std::min(o.mini_batches_in_flight ? o.mini_batches_in_flight : 1u, 64u));
  if (S.pending.size() >= W || burnNow || detectNow) S.flush_pending();
  if (burnNow) S.burnin_finalize();
  if (detectNow) S.detect_format();   // the following mini-batches expect the detected format
}

// ================================================================================================
// Inference (CollapsedEMOptimizer.cpp) — deterministic restatement (SPEC §D4)
// ================================================================================================
static double canonical_sum(std::vector<double> x) {  // SPEC §D2
  for (;;) {
    size_t n = x.size(); if (n == 0) return 0.0;
    size_t g = (n + 63) / 64; std::vector<double> p(g);
    for (size_t b = 0; b < g; ++b) {
      double v[64]; for (int i = 0; i < 64; ++i) v[i] = (b * 64 + i < n) ? x[b * 64 + i] : 0.0;
      for (int s = 32; s >= 1; s >>= 1) for (int i = 0; i < s; ++i) v[i] = v[i] + v[i + s];
      p[b] = v[0];
    }
    if (g == 1) return p[0];
    x.swap(p);
  }
}

struct EMProblem {
  uint32_t M; uint64_t E; std::vector<uint64_t> off, count; std::vector<uint32_t> tid; std::vector<double> cw;  // combined weights
  std::vector<uint64_t> t_off; std::vector<uint64_t> t_cls, t_pos;  // transcript-major incidence (class order)
  std::vector<double> prior;
};
static void em_setup(EMProblem& P, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o) {
  P.M = txp->num_txp; P.E = eq->num_classes; P.off.assign(eq->off, eq->off + P.E + 1); P.count.assign(eq->count, eq->count + P.E);
  P.tid.assign(eq->tid, eq->tid + eq->num_labels); P.cw.resize(eq->num_labels);
  for (uint64_t c = 0; c < P.E; ++c) {  // CollapsedEMOptimizer.cpp:830-873
    double wsum = 0.0;
    for (uint64_t i = P.off[c]; i < P.off[c + 1]; ++i) {
      double el = txp->eff_len[P.tid[i]]; if (el <= 1.0) el = 1.0;
      double w = o->no_rich_eq_classes ? 1.0 : eq->w[i];
      double wt = o->eq_class_mode ? w : (double)P.count[c] * w * (1.0 / el);
      P.cw[i] = wt; wsum += wt;
    }
    double wn = 1.0 / wsum;
    for (uint64_t i = P.off[c]; i < P.off[c + 1]; ++i) P.cw[i] = P.cw[i] * wn;
  }
  P.prior.assign(P.M, o->vb_prior);
  if (!o->per_transcript_prior) for (uint32_t i = 0; i < P.M; ++i) P.prior[i] = o->vb_prior * txp->eff_len[i];  // :82-99
  P.t_off.assign(P.M + 1, 0);
  for (uint64_t i = 0; i < P.tid.size(); ++i) P.t_off[P.tid[i] + 1]++;
  for (uint32_t t = 0; t < P.M; ++t) P.t_off[t + 1] += P.t_off[t];
  P.t_cls.resize(P.tid.size()); P.t_pos.resize(P.tid.size());
  std::vector<uint64_t> cur(P.t_off.begin(), P.t_off.end() - 1);
  for (uint64_t c = 0; c < P.E; ++c) for (uint64_t i = P.off[c]; i < P.off[c + 1]; ++i) {
    uint64_t d = cur[P.tid[i]]++;
    P.t_cls[d] = c;
    P.t_pos[d] = i;
  }
}
// one update: returns alphaOut (EMUpdate_ :178-234 / VBEMUpdate_ :241-328), transcript-major sums
static void em_step(const EMProblem& P, const sq_em_opts* o, const std::vector<double>& alphaIn, std::vector<double>& alphaOut,
    std::vector<double>& theta,
    std::vector<double>& invDenom) {
  const uint32_t M = P.M;
  if (o->use_vbem) {
    std::vector<double> ap(M); for (uint32_t i = 0; i < M; ++i) ap[i] = alphaIn[i] + P.prior[i];
    double logNorm = sq_digamma(canonical_sum(ap));
    for (uint32_t i = 0; i < M; ++i) theta[i] = (ap[i] > 1e-10) ? sq_exp(sq_digamma(ap[i]) - logNorm) : 0.0;
  } else theta = alphaIn;
  for (uint64_t c = 0; c < P.E; ++c) {
    uint64_t a = P.off[c], b = P.off[c + 1];
    if (b - a <= 1) { invDenom[c] = 0.0; continue; }
    double denom = 0.0;
    for (uint64_t i = a; i < b; ++i) { double th = theta[P.tid[i]]; if (!o->use_vbem || th > 0.0) denom += th * P.cw[i]; }
    invDenom[c] = (denom <= 2.2250738585072014e-308) ? 0.0 : (double)P.count[c] / denom;
  }
  std::vector<double> v, w;
  for (uint32_t t = 0; t < M; ++t) {  // SPEC §D4: blocked-64 sums over the transcript's incidences (class order)
    const double th = theta[t];
    v.clear();
    for (uint64_t d = P.t_off[t]; d < P.t_off[t + 1]; ++d) {
      uint64_t c = P.t_cls[d]; double term = 0.0;
      if (P.off[c + 1] - P.off[c] == 1) term = (double)P.count[c];
      else if (invDenom[c] != 0.0 && !(o->use_vbem && !(th > 0.0))) { double x = th * P.cw[P.t_pos[d]]; term = x * invDenom[c]; }
      v.push_back(term);
    }
    if (v.empty()) { alphaOut[t] = 0.0; continue; }
    for (;;) {
      w.clear();
      for (size_t i = 0; i < v.size(); i += 64) {
        double acc = 0.0;
        for (size_t j = i; j < std::min(v.size(), i + 64); ++j) acc += v[j];
        w.push_back(acc);
      }
      v.swap(w);
      if (v.size() == 1) break;
    }
    alphaOut[t] = v[0];
  }
}

// iteration loop shared by optimize (minIter 100) and the bootstrap replicates (minIter 50)
static void em_loop(const EMProblem& P, const sq_em_opts* o, std::vector<double>& alpha, uint32_t min_iter, uint32_t* iters,
    bool* converged, double* max_rel, uint32_t it0 = 0, uint32_t stop_at = 0 /* > 0: the part before the bias hook — until the convergence test holds, stop_at iterations at most */,
    bool plus_one = false /* [r5] optimize() without VBEM: the reference's alphasPrime hold 1.0 when its first EMUpdate_ adds into them (:797-823, :178-234 — no clearing there, unlike VBEMUpdate_ :265-283); found by tests/test_vbem_pin.py */) {
  const uint32_t M = P.M;
  std::vector<double> alphaP(M), theta(M), inv(P.E);
  uint32_t it = it0; bool conv = false; double maxRel = -1.7976931348623157e308;
  while (stop_at ? (it < stop_at && !conv) : (it < min_iter || (it < o->max_iter && !conv))) {
    em_step(P, o, alpha, alphaP, theta, inv);
    if (plus_one && it == 0 && !o->use_vbem) for (uint32_t i = 0; i < M; ++i) alphaP[i] += 1.0;
    conv = true; maxRel = -1.7976931348623157e308;
    for (uint32_t i = 0; i < M; ++i) {
      if (alphaP[i] > 1e-2) {
        double rd = std::fabs(alpha[i] - alphaP[i]) / alphaP[i];
        if (rd > maxRel) maxRel = rd;
        if (rd > o->rel_diff_tolerance) conv = false;
      }
      alpha[i] = alphaP[i]; alphaP[i] = 0.0;
    }
    ++it;
  }
  *iters = it; *converged = conv; *max_rel = maxRel;
}

static int em_optimize(const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, double* alpha_out, sq_em_report* rep) {
  EMProblem P; em_setup(P, eq, txp, o);
  const uint32_t M = P.M;
  std::vector<double> alpha(M);
  // initialisation (CollapsedEMOptimizer.cpp:778-823)
  std::vector<double> pc(M); double totalWeight = 0.0;
  for (uint32_t i = 0; i < M; ++i) { pc[i] = txp->projected_counts ? txp->projected_counts[i] : 0.0; }
  totalWeight = canonical_sum(pc);
  double uniformPrior = totalWeight / (double)M;
  double fracObserved = std::min(0.999, totalWeight / o->num_required_fragments);
  const bool alt = o->alt_init_mode && txp->unique_count;   // metaGenomeMode or altInitMode (CollapsedEMOptimizer.cpp:817-818)
  for (uint32_t i = 0; i < M; ++i) {
    const double uni = alt ? ((double)txp->unique_count[i] + 0.5) * 1e-3 * txp->eff_len[i] : uniformPrior;   // alphasPrime (:790-792)
    alpha[i] = o->init_uniform ? 100.0 : (pc[i] * fracObserved + uni * (1.0 - fracObserved));
  }
  // markDegenerateClasses (:330-394): invalid classes are skipped by every update (:197, :289) = they count for nothing
  uint32_t ndeg = 0;
  for (uint64_t c = 0; c < P.E; ++c) {
    double denom = 0.0;
    for (uint64_t i = P.off[c]; i < P.off[c + 1]; ++i) { double v = alpha[P.tid[i]] * P.cw[i]; if (!std::isnan(v)) denom += v; }
    if (denom <= 2.2250738585072014e-308) { P.count[c] = 0; for (uint64_t i = P.off[c]; i < P.off[c + 1]; ++i) P.cw[i] = 0.0; ++ndeg; }   // no count, no weight (the weights may be NaN)
  }
  uint32_t it; bool conv; double maxRel;
  em_loop(P, o, alpha, o->min_iter, &it, &conv, &maxRel, 0, 0, true);
  for (uint32_t i = 0; i < M; ++i) if (alpha[i] <= 1e-8) alpha[i] = 0.0;  // truncateCountVector :64-76
  double asum = canonical_sum(alpha);
  for (uint32_t i = 0; i < M; ++i) alpha_out[i] = alpha[i];
  if (rep) {
    rep->iters = it;
    rep->converged = conv;
    rep->max_rel_diff = maxRel;
    rep->alpha_sum = asum;
    rep->device_ms = 0;
    rep->ms_per_iter = 0;
    rep->num_degenerate = ndeg; rep->_pad = 0;
  }
  return asum < 2.2250738585072014e-308 ? SQ_ERR_STATE : SQ_OK;
}

// a16 — gatherBootstraps / doBootstrap (CollapsedEMOptimizer.cpp:398-690); SPEC §a16
static int bias_gc_eff_lengths(const Index& ix, const double* gc_obs, const double* log_pmf, uint32_t M, const double* alphas, const double* eff_in,
                               double* eff_out, double* bias_row0_out);
// salmon::utils::updateEffectiveLengths, gcBiasCorrect branches only (SalmonUtils.cpp:1208-1985), restated loop by loop; the sums follow
// SPEC §B: windows are counted as integers per (transcript, sampled length, GC bin), per-transcript terms are added in length order, the
// expected model is the canonical (blocked-64) sum over the processed transcripts in ascending id
static int bias_gc_eff_lengths(const Index& ix, const double* gc_obs, const double* log_pmf, uint32_t M, const double* alphas, const double* eff_in,
                               double* eff_out, double* bias_row0_out) {
  const int MAXV = 1000; const int32_t gcSamp = 5;
  std::vector<double> pdf(MAXV + 1), cdf(MAXV + 1); int32_t fldLow = 0, fldHigh = 1; bool lb = false, ub = false;
  for (int i = 0; i <= MAXV; ++i) {
    pdf[i] = sq_exp(log_pmf[i]); cdf[i] = (i > 0) ? cdf[i - 1] + pdf[i] : pdf[i];
    if (!lb && cdf[i] >= 0.005) { lb = true; fldLow = i; }
    if (!ub && cdf[i] >= 1.0 - 0.005) { ub = true; fldHigh = i; }
  }
  std::vector<std::vector<double>> contrib(25);
  std::vector<uint32_t> processed;
  // per transcript: G/C prefix (GCCount_), then the two sweeps
  auto prefix = [&](uint32_t t) { std::vector<int32_t> g(ix.ref_len[t]); int32_t c = 0; for (uint32_t i = 0; i < ix.ref_len[t]; ++i) { c += is_gc(ix, t, (int32_t)i) ? 1 : 0; g[i] = c; } return g; };
  auto gcFrac = [](const std::vector<int32_t>& g, int32_t s, int32_t e) { int32_t cs = s > 0 ? g[s - 1] : 0, ce = g[e]; return (int32_t)std::lrint((100.0 * (double)(ce - cs)) / (double)(e - s + 1)); };
  for (uint32_t t = 0; t < M; ++t) {
    const int32_t refLen = (int32_t)ix.ref_len[t], elen = (int32_t)eff_in[t], unprocessedLen = std::max(0, refLen - elen);
    const int32_t cdfMaxArg = std::min(MAXV, refLen); const double cdfMaxVal = cdf[cdfMaxArg];
    if (cdfMaxVal < 1e-10) continue;
    if (alphas[t] < 1e-8 || unprocessedLen <= 0) continue;
    auto cCDF = [&](int32_t x) { return x > cdfMaxArg ? 1.0 : cdf[x] / cdfMaxVal; };
    processed.push_back(t);
    const double weight = alphas[t] / eff_in[t];
    const std::vector<int32_t> g = prefix(t);
    const int32_t locFLDLow = (refLen < cdfMaxArg) ? 1 : fldLow, locFLDHigh = (refLen < cdfMaxArg) ? cdfMaxArg : fldHigh;
    double E[25] = {0};
    double prev = cCDF(locFLDLow > 0 ? locFLDLow - 1 : 0);
    for (int32_t fl = locFLDLow; fl <= locFLDHigh; fl += gcSamp) {      // length-major form of the (start, length) double loop: same terms
      uint64_t N[25] = {0}; bool any = false;
      for (int32_t fragStart = 0; fragStart < refLen - 1; ++fragStart) {
        const int32_t fragEnd = fragStart + fl - 1;
        if (fragEnd < refLen) { N[gc_frag_bin(gcFrac(g, fragStart, fragEnd))]++; any = true; } else break;
      }
      if (fl > refLen || fl < 1) break;
      (void)any;
      const double d = cCDF(fl) - prev; prev = cCDF(fl);
      for (int b = 0; b < 25; ++b) E[b] += d * (double)N[b];
    }
    for (int b = 0; b < 25; ++b) contrib[b].push_back(weight * E[b]);
  }
  double expect[3][25] = {{0}};
  for (int b = 0; b < 25; ++b) expect[0][b] = canonical_sum(contrib[b]);
  double obsN[3][25], expN[3][25], bias[3][25];
  auto normalize = [](const double* in, double* out) {   // GCFragModel::normalize, LINEAR branch (prior 0.1)
    double rowMass = 0.0; for (int c = 0; c < 25; ++c) rowMass += (0.1 + in[c]);
    if (rowMass > 0.0) { double norm = 1.0 / rowMass; for (int c = 0; c < 25; ++c) out[c] = (0.1 + in[c]) * norm; } else for (int c = 0; c < 25; ++c) out[c] = in[c];
  };
  for (int r = 0; r < 3; ++r) {
    normalize(gc_obs + 25 * r, obsN[r]); normalize(expect[r], expN[r]);
    for (int c = 0; c < 25; ++c) { double rat = obsN[r][c] / expN[r][c]; if (rat > 1000.0) rat = 1000.0; if (rat < 1.0 / 1000.0) rat = 1.0 / 1000.0; bias[r][c] = rat; }   // ratio(other, 1000)
  }
  if (bias_row0_out) for (int c = 0; c < 25; ++c) bias_row0_out[c] = bias[0][c];
  for (uint32_t t = 0; t < M; ++t) {
    const int32_t refLen = (int32_t)ix.ref_len[t], elen = (int32_t)eff_in[t], unprocessedLen = std::max(0, refLen - elen);
    const int32_t cdfMaxArg = std::min(MAXV, refLen); const double cdfMaxVal = cdf[cdfMaxArg];
    auto cCDF = [&](int32_t x) { return x > cdfMaxArg ? 1.0 : cdf[x] / cdfMaxVal; };
    const int32_t locFLDLow = (refLen < cdfMaxArg) ? 1 : fldLow, locFLDHigh = (refLen < cdfMaxArg) ? cdfMaxArg : fldHigh;
    if (!(alphas[t] >= 1e-8 && unprocessedLen > 0 && cdfMaxVal > 1e-10)) { eff_out[t] = (double)elen; continue; }
    const std::vector<int32_t> g = prefix(t);
    double effLength = 0.0;
    int32_t fl = locFLDLow; const int32_t maxLen = std::min(refLen, locFLDHigh + 1); bool done = fl >= maxLen;
    double prevFLMass = cCDF(fl > 0 ? fl - 1 : 0);
    while (!done) {
      if (fl >= maxLen) { done = true; fl = maxLen - 1; }
      const double flWeight = cCDF(fl) - prevFLMass; prevFLMass = cCDF(fl);
      uint64_t N[25] = {0};
      for (int32_t kmerStartPos = 0; kmerStartPos < refLen - fl; ++kmerStartPos) {
        const int32_t fragStart = kmerStartPos, fragEnd = fragStart + fl - 1;
        if (fragStart < refLen && fragEnd < refLen && fl >= 1) N[gc_frag_bin(gcFrac(g, fragStart, fragEnd))]++; else break;
      }
      double flMassTotal = 0.0; for (int b = 0; b < 25; ++b) flMassTotal += (double)N[b] * bias[0][b];   // every fragment factor is one of 25 values
      effLength += flWeight * flMassTotal;
      fl += gcSamp;
    }
    const double thresh = (double)unprocessedLen, offset = std::max(1.0, thresh), effLengthNoBias = (double)elen;
    eff_out[t] = std::max(effLength, std::min(effLengthNoBias, offset));
  }
  return (int)processed.size();
}

// salmon::utils::updateEffectiveLengths with --seqBias, alone or with --gcBias (SalmonUtils.cpp:1208-1985), restated loop by loop.  SPEC §B2:
//  * expected context models: a transcript contributes weight x (C + F) to a cell, C = the integer count of its starts whose
//    conditional-CDF factor is exactly 1 (the fragment could be longer than the distribution reaches), F = the factors of the other
//    starts (the last <= 1000) added in start order; a model cell = 1e-10 + the canonical sum over the processed transcripts;
//  * expected GC model (with --gcBias): integer windows per (sampled length, context bin, GC bin), as in §B, over the starts the
//    reference's loop visits (fragStart < refLen - K, K = 9);
//  * effective length: for every sampled length the sum over fragment starts of seqFW[start] * seqRC[end] (* gcBias) is the lane-strided
//    sum (lane_sum256); the lengths are added in order.
//  * --posBias (SPEC §P; :1639-1652, 1708-1712, 1815-1835, 1941-1944): the expected read-start models get, per processed transcript and
//    bin, the lane-strided sum over the bin's starts of weight x conditional CDF (terms <= EPSILON dropped), summed canonically over the
//    transcripts; every model bin starts with mass 1 + T (the shared model and one local copy per worker, each initialised to LOG_1);
//    masses are kept linear (the reference keeps logs and exponentiates in finalize()).
struct SeqBiasOut { double exp_fw[576], exp_rc[576], obs_fw[576], obs_rc[576]; uint32_t processed; };
struct PosIn { const double* obs; uint32_t threads; };           // [2][100] linear observed masses without the prior; T
struct PosOut { double obs_norm[2][100], exp_norm[2][100]; };    // SimplePosBias::masses_ after finalize()
static int bias_seq_eff_lengths(const Index& ix, bool gc, const double* gc_obs, const uint64_t* seq_fw, const uint64_t* seq_rc, const double* log_pmf, uint32_t M,
                                const double* alphas, const double* eff_in, double* eff_out, SeqBiasOut* out, const PosIn* pos = nullptr, PosOut* pout = nullptr) {
  const int MAXV = 1000; const int32_t gcSamp = 5; const bool seq = seq_fw != nullptr; const int K = seq ? SB_K : 1;
  std::vector<uint32_t> lq; std::vector<uint8_t> lcls; if (pos) length_classes(ix, lq, lcls);
  std::vector<std::vector<double>> cpos(pos ? 200 : 0);
  std::vector<double> pdf(MAXV + 1), cdf(MAXV + 1); int32_t fldLow = 0, fldHigh = 1; bool lb = false, ub = false;
  for (int i = 0; i <= MAXV; ++i) {
    pdf[i] = sq_exp(log_pmf[i]); cdf[i] = (i > 0) ? cdf[i - 1] + pdf[i] : pdf[i];
    if (!lb && cdf[i] >= 0.005) { lb = true; fldLow = i; }
    if (!ub && cdf[i] >= 1.0 - 0.005) { ub = true; fldHigh = i; }
  }
  auto prefix = [&](uint32_t t) { std::vector<int32_t> g(ix.ref_len[t]); int32_t c = 0; for (uint32_t i = 0; i < ix.ref_len[t]; ++i) { c += is_gc(ix, t, (int32_t)i) ? 1 : 0; g[i] = c; } return g; };
  auto gcFrac = [](const std::vector<int32_t>& g, int32_t s, int32_t e) { int32_t cs = s > 0 ? g[s - 1] : 0, ce = g[e]; return (int32_t)std::lrint((100.0 * (double)(ce - cs)) / (double)(e - s + 1)); };
  // populateContextCounts (:1372-1424), the loop as written (including what it does once the window reaches the last base)
  auto context = [&](uint32_t t, const std::vector<int32_t>& g, std::vector<int32_t>& cFP, std::vector<int32_t>& cTP, std::vector<int32_t>& wFP, std::vector<int32_t>& wTP) {
    const int32_t refLen = (int32_t)ix.ref_len[t]; cFP.assign(refLen, 0); cTP.assign(refLen, 0); wFP.assign(refLen, 0); wTP.assign(refLen, 0);
    const int outside = 3, inside = 2, csize = outside + inside;
    if (refLen > csize) {
      int windowEnd = inside - 1, windowStart = -outside, fp = 0, tp = windowStart + (inside - 1);
      int32_t count = g[windowEnd - 1];
      for (; tp < refLen; ++fp, ++tp) {
        if (windowStart > 0 && is_gc(ix, t, windowStart - 1)) count -= 1;
        if (windowEnd < refLen && is_gc(ix, t, windowEnd)) count += 1;
        const int32_t wl = (windowEnd < csize) ? windowEnd + 1 : (windowEnd - windowStart + 1);
        if (fp < refLen) { cFP[fp] = count; wFP[fp] = wl; }
        if (tp >= 0) { cTP[tp] = count; wTP[tp] = wl; }
        if (windowEnd < refLen - 1) ++windowEnd;
        ++windowStart;
      }
    }
  };
  auto ctxFrac = [](const std::vector<int32_t>& cFP, const std::vector<int32_t>& cTP, const std::vector<int32_t>& wFP, const std::vector<int32_t>& wTP, int32_t s, int32_t e) {
    const double cl = (double)(wFP[s] + wTP[e]);
    return cl > 0 ? (int32_t)std::lrint(100.0 * (double)(cFP[s] + cTP[e]) / cl) : 0;
  };
  std::vector<uint32_t> processed;
  std::vector<std::vector<double>> cfw(576), crc(576), cgc(75);
  for (uint32_t t = 0; t < M; ++t) {
    const int32_t refLen = (int32_t)ix.ref_len[t], elen = (int32_t)eff_in[t], unprocessedLen = std::max(0, refLen - elen);
    const int32_t cdfMaxArg = std::min(MAXV, refLen); const double cdfMaxVal = cdf[cdfMaxArg];
    if (cdfMaxVal < 1e-10) continue;
    if (alphas[t] < 1e-8 || unprocessedLen <= 0) continue;
    auto cCDF = [&](int32_t x) { return x > cdfMaxArg ? 1.0 : cdf[x] / cdfMaxVal; };
    processed.push_back(t);
    const double weight = alphas[t] / eff_in[t];
    // sequence contexts
    if (seq) { uint64_t Cf[576] = {0}, Cr[576] = {0}; double Ff[576] = {0}, Fr[576] = {0};
      for (int32_t fsp = 0; fsp < refLen - K; ++fsp) {
        const int32_t maxFragLen = refLen - (fsp + SB_LEFT);
        if (!(maxFragLen >= 0 && maxFragLen < refLen)) continue;
        const uint32_t fw = sb_ctx(ix, t, fsp), rc = sb_rc(sb_ctx(ix, t, refLen - K - fsp));
        if (maxFragLen > cdfMaxArg) { for (int i = 0; i < K; ++i) { Cf[sb_cell(fw, i)]++; Cr[sb_cell(rc, i)]++; } }
        else { const double cd = cdf[maxFragLen] / cdfMaxVal; for (int i = 0; i < K; ++i) { Ff[sb_cell(fw, i)] += cd; Fr[sb_cell(rc, i)] += cd; } }
      }
      for (int c = 0; c < 576; ++c) { cfw[c].push_back(weight * ((double)Cf[c] + Ff[c])); crc[c].push_back(weight * ((double)Cr[c] + Fr[c])); } }
    if (gc) {
      const std::vector<int32_t> g = prefix(t); std::vector<int32_t> cFP, cTP, wFP, wTP;
      if (seq) context(t, g, cFP, cTP, wFP, wTP); else { cFP.assign(refLen, 0); cTP.assign(refLen, 0); wFP.assign(refLen, 0); wTP.assign(refLen, 0); }   // :1566: contexts only with both
      const int32_t locFLDLow = (refLen < cdfMaxArg) ? 1 : fldLow, locFLDHigh = (refLen < cdfMaxArg) ? cdfMaxArg : fldHigh;
      double E[75] = {0};
      double prev = cCDF(locFLDLow > 0 ? locFLDLow - 1 : 0);
      for (int32_t fl = locFLDLow; fl <= locFLDHigh; fl += gcSamp) {
        if (fl > refLen || fl < 1) break;
        uint64_t N[75] = {0};
        for (int32_t fs = 0; fs < refLen - K; ++fs) { const int32_t fe = fs + fl - 1; if (fe < refLen) N[gc_ctx_bin(ctxFrac(cFP, cTP, wFP, wTP, fs, fe)) * 25 + gc_frag_bin(gcFrac(g, fs, fe))]++; else break; }
        const double d = cCDF(fl) - prev; prev = cCDF(fl);
        for (int b = 0; b < 75; ++b) E[b] += d * (double)N[b];
      }
      for (int b = 0; b < 75; ++b) cgc[b].push_back(weight * E[b]);
    }
    if (pos) {   // :1639-1652
      double e[200] = {0}; const int li = lcls[t]; const int32_t ns = refLen - K;
      int32_t s0 = 0;
      while (s0 < ns) {
        const int b = pos_bin(s0, (uint32_t)refLen); int32_t s1 = s0; while (s1 < ns && pos_bin(s1, (uint32_t)refLen) == b) ++s1;
        e[li * 20 + b] = lane_sum256((int64_t)(s1 - s0), [&](int64_t i) { const double v = weight * cCDF(refLen - (s0 + (int32_t)i) + 1); return v > 0.375e-10 ? v : 0.0; });
        e[100 + li * 20 + b] = lane_sum256((int64_t)(s1 - s0), [&](int64_t i) { const double v = weight * cCDF(s0 + (int32_t)i); return v > 0.375e-10 ? v : 0.0; });
        s0 = s1;
      }
      for (int c = 0; c < 200; ++c) cpos[c].push_back(e[c]);
    }
  }
  PosSpline O5[5], O3[5], E5[5], E3[5];
  if (pos) {
    const double prior = 1.0 + (double)pos->threads;
    for (int li = 0; li < 5; ++li) {
      double mo5[20], mo3[20], me5[20], me3[20];
      for (int b = 0; b < 20; ++b) { mo5[b] = prior + pos->obs[li * 20 + b]; mo3[b] = prior + pos->obs[100 + li * 20 + b];
        me5[b] = prior + canonical_sum(cpos[li * 20 + b]); me3[b] = prior + canonical_sum(cpos[100 + li * 20 + b]); }
      pos_finalize(mo5, O5[li], pout ? &pout->obs_norm[0][li * 20] : nullptr); pos_finalize(mo3, O3[li], pout ? &pout->obs_norm[1][li * 20] : nullptr);
      pos_finalize(me5, E5[li], pout ? &pout->exp_norm[0][li * 20] : nullptr); pos_finalize(me3, E3[li], pout ? &pout->exp_norm[1][li * 20] : nullptr);
    }
  }
  // models
  double cnt_efw[576], cnt_erc[576], cnt_ofw[576], cnt_orc[576], efw[576], erc[576], ofw[576], orc_[576];
  if (!seq) { for (int c = 0; c < 576; ++c) { cnt_efw[c] = cnt_erc[c] = cnt_ofw[c] = cnt_orc[c] = 1.0; cfw[c].clear(); crc[c].clear(); } }
  else for (int c = 0; c < 576; ++c) { cnt_efw[c] = 1e-10 + canonical_sum(cfw[c]); cnt_erc[c] = 1e-10 + canonical_sum(crc[c]); cnt_ofw[c] = 1e-10 + (double)seq_fw[c]; cnt_orc[c] = 1e-10 + (double)seq_rc[c]; }
  sb_normalize(cnt_efw, efw); sb_normalize(cnt_erc, erc); sb_normalize(cnt_ofw, ofw); sb_normalize(cnt_orc, orc_);
  if (out) { memcpy(out->exp_fw, efw, sizeof(efw)); memcpy(out->exp_rc, erc, sizeof(erc)); memcpy(out->obs_fw, ofw, sizeof(ofw)); memcpy(out->obs_rc, orc_, sizeof(orc_)); out->processed = (uint32_t)processed.size(); }
  double bias[3][25]; for (int r = 0; r < 3; ++r) for (int c = 0; c < 25; ++c) bias[r][c] = 1.0;
  if (gc) {
    double expect[3][25], obsN[3][25], expN[3][25];
    for (int b = 0; b < 75; ++b) expect[b / 25][b % 25] = canonical_sum(cgc[b]);
    auto normalize = [](const double* in, double* o2) { double rowMass = 0.0; for (int c = 0; c < 25; ++c) rowMass += (0.1 + in[c]);
      if (rowMass > 0.0) { double norm = 1.0 / rowMass; for (int c = 0; c < 25; ++c) o2[c] = (0.1 + in[c]) * norm; } else for (int c = 0; c < 25; ++c) o2[c] = in[c]; };
    for (int r = 0; r < 3; ++r) { normalize(gc_obs + 25 * r, obsN[r]); normalize(expect[r], expN[r]);
      for (int c = 0; c < 25; ++c) { double rat = obsN[r][c] / expN[r][c]; if (rat > 1000.0) rat = 1000.0; if (rat < 1.0 / 1000.0) rat = 1.0 / 1000.0; bias[r][c] = rat; } }
  }
  // effective lengths
  for (uint32_t t = 0; t < M; ++t) {
    const int32_t refLen = (int32_t)ix.ref_len[t], elen = (int32_t)eff_in[t], unprocessedLen = std::max(0, refLen - elen);
    const int32_t cdfMaxArg = std::min(MAXV, refLen); const double cdfMaxVal = cdf[cdfMaxArg];
    auto cCDF = [&](int32_t x) { return x > cdfMaxArg ? 1.0 : cdf[x] / cdfMaxVal; };
    const int32_t locFLDLow = (refLen < cdfMaxArg) ? 1 : fldLow, locFLDHigh = (refLen < cdfMaxArg) ? cdfMaxArg : fldHigh;
    if (!(alphas[t] >= 1e-8 && unprocessedLen > 0 && cdfMaxVal > 1e-10)) { eff_out[t] = (double)elen; continue; }
    std::vector<double> sFW(refLen, 1.0), sRCt(refLen, 1.0), sRC(refLen, 1.0);
    if (seq) for (int32_t fs = 0; fs < refLen - K; ++fs) {
      const int32_t readStart = fs + SB_LEFT;
      if (readStart < refLen) {
        const uint32_t fw = sb_ctx(ix, t, fs), rc = sb_rc(sb_ctx(ix, t, refLen - K - fs));
        sFW[readStart] = sq_exp(sb_eval(ofw, fw) - sb_eval(efw, fw));
        sRCt[readStart] = sq_exp(sb_eval(orc_, rc) - sb_eval(erc, rc));
      }
    }
    for (int32_t j = 0; j < refLen; ++j) sRC[j] = sRCt[refLen - 1 - j];
    std::vector<int32_t> g, cFP, cTP, wFP, wTP;
    if (gc) { g = prefix(t); if (seq) context(t, g, cFP, cTP, wFP, wTP); else { cFP.assign(refLen, 0); cTP.assign(refLen, 0); wFP.assign(refLen, 0); wTP.assign(refLen, 0); } }
    std::vector<double> pFW, pRC;
    if (pos) { pFW.assign(refLen, 1.0); pRC.assign(refLen, 1.0); const int li = lcls[t];
      for (int32_t fs = 0; fs < refLen - K; ++fs) { pFW[fs] = pos_weight(O5[li], fs, refLen) / pos_weight(E5[li], fs, refLen); pRC[fs] = pos_weight(O3[li], fs, refLen) / pos_weight(E3[li], fs, refLen); } }
    double effLength = 0.0;
    int32_t fl = locFLDLow; const int32_t maxLen = std::min(refLen, locFLDHigh + 1); bool done = fl >= maxLen;
    double prevFLMass = cCDF(fl > 0 ? fl - 1 : 0);
    while (!done) {
      if (fl >= maxLen) { done = true; fl = maxLen - 1; }
      const double flWeight = cCDF(fl) - prevFLMass; prevFLMass = cCDF(fl);
      const int32_t cur = fl;
      const double flMassTotal = lane_sum256((int64_t)std::max(0, refLen - cur), [&](int64_t s0) {
        const int32_t fs = (int32_t)s0, fe = fs + cur - 1;
        double f = sFW[fs] * sRC[fe];
        if (gc) f *= bias[gc_ctx_bin(ctxFrac(cFP, cTP, wFP, wTP, fs, fe))][gc_frag_bin(gcFrac(g, fs, fe))];
        if (pos) f *= pFW[fs] * pRC[fe];
        return f; });
      effLength += flWeight * flMassTotal;
      fl += gcSamp;
    }
    const double thresh = (double)unprocessedLen, offset = std::max(1.0, thresh), effLengthNoBias = (double)elen;
    eff_out[t] = std::max(effLength, std::min(effLengthNoBias, offset));
  }
  return (int)processed.size();
}

// optimize() with the bias hook (CollapsedEMOptimizer.cpp:901-928: `itNum > targetIt or converged`): after 11 updates — or at the first update
// after which the convergence test holds, if that comes earlier — updateEffectiveLengths, new priors
// (populatePriorAlphas_) and combined weights (updateEqClassWeights :160-176; degenerate classes stay dropped), then on to convergence
static int em_optimize_gc(const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, const Index& ix, const double* gc_obs, const double* log_pmf,
                          double* alpha_out, double* eff_out, sq_em_report* rep, const uint64_t* seq_fw = nullptr, const uint64_t* seq_rc = nullptr, const PosIn* pos = nullptr) {
  EMProblem P; em_setup(P, eq, txp, o);
  const uint32_t M = P.M;
  std::vector<double> alpha(M), pc(M), eff(txp->eff_len, txp->eff_len + M), eff2(M);
  for (uint32_t i = 0; i < M; ++i) pc[i] = txp->projected_counts ? txp->projected_counts[i] : 0.0;
  const double totalWeight = canonical_sum(pc), uniformPrior = totalWeight / (double)M, fracObserved = std::min(0.999, totalWeight / o->num_required_fragments);
  const bool alt = o->alt_init_mode && txp->unique_count;
  for (uint32_t i = 0; i < M; ++i) { const double uni = alt ? ((double)txp->unique_count[i] + 0.5) * 1e-3 * txp->eff_len[i] : uniformPrior; alpha[i] = o->init_uniform ? 100.0 : (pc[i] * fracObserved + uni * (1.0 - fracObserved)); }
  uint32_t ndeg = 0; std::vector<uint8_t> dropped(P.E, 0);
  for (uint64_t c = 0; c < P.E; ++c) {
    double denom = 0.0;
    for (uint64_t i = P.off[c]; i < P.off[c + 1]; ++i) { double v = alpha[P.tid[i]] * P.cw[i]; if (!std::isnan(v)) denom += v; }
    if (denom <= 2.2250738585072014e-308) { P.count[c] = 0; dropped[c] = 1; for (uint64_t i = P.off[c]; i < P.off[c + 1]; ++i) P.cw[i] = 0.0; ++ndeg; }
  }
  uint32_t it; bool conv; double maxRel;
  em_loop(P, o, alpha, o->min_iter, &it, &conv, &maxRel, 0, 11, true);
  if (seq_fw || pos) bias_seq_eff_lengths(ix, gc_obs != nullptr, gc_obs, seq_fw, seq_rc, log_pmf, M, alpha.data(), eff.data(), eff2.data(), nullptr, pos);   // --seqBias / --posBias [+ --gcBias]
  else bias_gc_eff_lengths(ix, gc_obs, log_pmf, M, alpha.data(), eff.data(), eff2.data(), nullptr);
  for (uint64_t c = 0; c < P.E; ++c) {   // updateEqClassWeights with the new lengths
    if (dropped[c]) continue;
    double wsum = 0.0;
    for (uint64_t i = P.off[c]; i < P.off[c + 1]; ++i) {
      double el = eff2[P.tid[i]]; if (el <= 1.0) el = 1.0;
      double w = o->no_rich_eq_classes ? 1.0 : eq->w[i];
      double wt = o->eq_class_mode ? w : (double)eq->count[c] * w * (1.0 / el);
      P.cw[i] = wt; wsum += wt;
    }
    double wn = 1.0 / wsum;
    for (uint64_t i = P.off[c]; i < P.off[c + 1]; ++i) P.cw[i] = P.cw[i] * wn;
  }
  if (!o->per_transcript_prior) for (uint32_t i = 0; i < M; ++i) P.prior[i] = o->vb_prior * eff2[i];
  em_loop(P, o, alpha, o->min_iter, &it, &conv, &maxRel, it, 0);
  for (uint32_t i = 0; i < M; ++i) if (alpha[i] <= 1e-8) alpha[i] = 0.0;
  double asum = canonical_sum(alpha);
  for (uint32_t i = 0; i < M; ++i) { alpha_out[i] = alpha[i]; if (eff_out) eff_out[i] = eff2[i]; }
  if (rep) { rep->iters = it; rep->converged = conv; rep->max_rel_diff = maxRel; rep->alpha_sum = asum; rep->device_ms = 0; rep->ms_per_iter = 0; rep->num_degenerate = ndeg; rep->_pad = 0; }
  return asum < 2.2250738585072014e-308 ? SQ_ERR_STATE : SQ_OK;
}

static int bootstrap(const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, uint32_t B, uint64_t seed, uint64_t num_mapped,
    double* out) {
  EMProblem P; em_setup(P, eq, txp, o);
  const uint32_t M = P.M; const uint64_t E = P.E;
  std::vector<uint64_t> cum(E), orig(P.count); uint64_t total = 0; for (uint64_t c = 0; c < E; ++c) { total += orig[c]; cum[c] = total; }
  std::vector<uint8_t> active(M, 0); for (uint32_t t : P.tid) active[t] = 1;
  uint32_t nact = 0; for (auto a : active) nact += a;
  if (!nact || !total) return SQ_ERR_STATE;
  const double scale = 1.0 / (double)nact;
  for (uint32_t b = 0; b < B; ++b) {
    std::fill(P.count.begin(), P.count.end(), 0);
    for (uint64_t i = 0; i < total; ++i) {
      uint64_t idx = sq_mulhi64(sq_r64(seed, b, i), total);
      size_t c = std::upper_bound(cum.begin(), cum.end(), idx) - cum.begin();
      P.count[c]++;
    }
    std::vector<double> alpha(M); for (uint32_t i = 0; i < M; ++i) alpha[i] = active[i] ? scale * (double)num_mapped : 0.0;
    uint32_t it; bool conv; double mr; em_loop(P, o, alpha, 50, &it, &conv, &mr);
    for (uint32_t i = 0; i < M; ++i) out[(size_t)b * M + i] = alpha[i] <= 1e-8 ? 0.0 : alpha[i];
  }
  return SQ_OK;
}

// a17 — CollapsedGibbsSampler::sample + sampleRoundNonCollapsedMultithreaded_ (CollapsedGibbsSampler.cpp:92-278, 317-508); SPEC §a17
static int gibbs(const sq_eq_table* eq, const sq_txp_in* txp, const sq_gibbs_opts* go, const double* alpha_init, uint32_t S, uint64_t seed,
    uint64_t num_mapped,
    double* out) {
  const uint32_t M = txp->num_txp; const uint64_t E = eq->num_classes;
  const bool perTxp = go->use_vbem ? go->per_transcript_prior != 0 : true;
  double pv = 1e-3; if (go->use_vbem) pv = perTxp ? (go->vb_prior < 1.0 ? 1.0 : go->vb_prior) : (go->vb_prior < 1e-3 ? 1e-3 : go->vb_prior);
  std::vector<double> prior(M, pv); if (!perTxp) for (uint32_t i = 0; i < M; ++i) prior[i] = pv * txp->eff_len[i];
  std::vector<uint8_t> active(M, 0); for (uint64_t i = 0; i < eq->num_labels; ++i) active[eq->tid[i]] = 1;
  std::vector<double> init(alpha_init, alpha_init + M); for (uint32_t i = 0; i < M; ++i) if (!active[i]) init[i] = 0.0;
  std::vector<uint64_t> draw_off(E + 1, 0); for (uint64_t c = 0; c < E; ++c) draw_off[c + 1] = draw_off[c] + eq->count[c];
  uint32_t nchains = 1; if (S >= 50) nchains = 2; if (S >= 100) nchains = 4; if (S >= 200) nchains = 8;
  const uint32_t step = nchains > 1 ? S / nchains : S + 1; const uint32_t thin = go->thinning_factor ? go->thinning_factor : 16;
  std::vector<double> cf(init), mu(M, 0.0), me(M); std::vector<uint64_t> ci(M);
  for (uint32_t sid = 0; sid < S; ++sid) {
    if (sid > 0 && nchains > 1 && sid % step == 0 && sid / step < nchains) cf = init;
    for (uint32_t r = 0; r < thin; ++r) {
      const uint64_t key = (uint64_t)sid * thin + r;
      for (uint32_t i = 0; i < M; ++i) {
        if (!active[i]) {
          mu[i] = 0.0;
          continue;
        }
        double c = cf[i] + prior[i];
        mu[i] = go->no_gamma_draw ? c / txp->eff_len[i] : sq_gamma_draw(c, 1.0 / (0.1 + txp->eff_len[i]), seed, key, i);
      }
      std::fill(ci.begin(), ci.end(), 0);
      for (uint64_t c = 0; c < E; ++c) {
        const uint64_t a = eq->off[c]; const uint32_t n = (uint32_t)(eq->off[c + 1] - a); const uint64_t cnt = eq->count[c];
        if (n == 0 || cnt == 0) continue;
        if (n == 1) { ci[eq->tid[a]] += cnt; continue; }
        auto pf = [&](int mode, uint32_t i) {
          uint32_t t = eq->tid[a + i];
          return mode == 0 ? (1000.0 * mu[t]) * eq->w[a + i] : (mode == 1 ? 1.0 / txp->eff_len[t] : 1.0);
        };
        int mode = 0; double denom = 0.0; for (uint32_t i = 0; i < n; ++i) denom += pf(0, i);
        if (denom <= 2.2250738585072014e-308) {
          mode = 1;
          denom = 0.0;
          for (uint32_t i = 0; i < n; ++i) denom += pf(1, i);
          if (denom <= 2.2250738585072014e-308) {
            mode = 2;
            denom = (double)n;
          }
        }
        for (uint64_t sidx = 0; sidx < cnt; ++sidx) {
          double u = sq_u01(sq_r64(seed ^ 0xC1A55ULL, key, draw_off[c] + sidx)) * denom;
          double acc = 0.0;
          uint32_t pick = n - 1;
          for (uint32_t i = 0; i < n; ++i) {
            acc += pf(mode, i);
            if (u < acc) {
              pick = i;
              break;
            }
          }
          ci[eq->tid[a + pick]]++;
        }
      }
      for (uint32_t i = 0; i < M; ++i) cf[i] = (double)ci[i];
    }
    for (uint32_t i = 0; i < M; ++i) me[i] = mu[i] * txp->eff_len[i];
    double scale = (double)num_mapped / canonical_sum(me);
    for (uint32_t i = 0; i < M; ++i) { double v = me[i] * scale; out[(size_t)sid * M + i] = v > 1e-8 ? v : 0.0; }
  }
  return SQ_OK;
}

}  // namespace orc

// ================================================================================================
// C interface used by tests / smoke / bench cpu_baseline (ctypes)
// ================================================================================================
using namespace orc;
struct orc_index { Index ix; };
struct orc_state { QuantState S; };

extern "C" {

// Build the checker's index from the product's host-side sections (unitigs + contig table are the
// shared *input*; the dictionary itself is rebuilt here as a brute-force hash map).
orc_index* orc_index_from_view(const sq_index_view* v, const char* const* names) {
  orc_index* o = new orc_index(); Index& ix = o->ix;
  ix.k = v->k; ix.first_decoy = v->first_decoy;
  for (uint32_t i = 0; i < v->num_refs; ++i) ix.names.push_back(names ? names[i] : std::to_string(i));
  ix.ref_len.assign(v->ref_len, v->ref_len + v->num_refs); ix.ref_clen.assign(v->ref_clen, v->ref_clen + v->num_refs);
  ix.ref_accum.assign(v->ref_accum, v->ref_accum + v->num_refs + 1);
  ix.refseq.assign(v->refseq, v->refseq + (v->total_ref_nt + 31) / 32 + 1);
  ix.useq.assign(v->useq, v->useq + (v->total_unitig_nt + 31) / 32 + 1);
  ix.uoff.assign(v->uoff, v->uoff + v->num_unitigs + 1); ix.ctab_off.assign(v->ctab_off, v->ctab_off + v->num_unitigs + 1);
  ix.ctab.assign(v->ctab, v->ctab + v->num_occ);
  ix.build_dict();
  return o;
}
void orc_index_free(orc_index* o) { delete o; }
uint64_t orc_index_num_kmers(const orc_index* o) { return o->ix.num_kmers; }
int orc_index_lookup(const orc_index* o, uint64_t kmer, uint64_t* u, uint32_t* off, int* fw) {
  bool f; uint64_t uu; uint32_t oo; if (!o->ix.lookup(kmer & kmask(o->ix.k), uu, oo, f)) return 0; *u = uu; *off = oo; *fw = f; return 1;
}

// Independent brute-force check of the compacted de Bruijn graph held in a view: returns 0 if the
// unitigs (a) tile every reference exactly as the contig table says, (b) contain every canonical
// k-mer exactly once, (c) are maximal under the rules of SPEC §I; else a positive error code.
int orc_check_cdbg(const sq_index_view* v) {
  const uint32_t k = v->k; const uint64_t km = kmask(k);
  std::unordered_map<uint64_t, uint32_t> seen;  // canonical k-mer -> count in unitigs
  for (uint64_t u = 0; u < v->num_unitigs; ++u) {
    uint64_t b = v->uoff[u], e = v->uoff[u + 1]; if (e - b < k) return 1;
    uint64_t fw = 0;
    for (uint64_t i = 0; i < e - b; ++i) {
      fw = (fw >> 2) | ((uint64_t)base_at(v->useq, b + i) << (2 * (k - 1)));
      if (i + 1 < k) continue;
      fw &= km;
      uint64_t rc = revcomp(fw, k);
      if (++seen[std::min(fw, rc)] > 1) return 2;
    }
  }
  // every reference k-mer present; occurrences reproduce references
  std::vector<uint8_t> covered;
  for (uint32_t r = 0; r < v->num_refs; ++r) {
    uint32_t L = v->ref_len[r]; covered.assign(L, 0);
    if (L >= k) {
      uint64_t fw = 0;
      for (uint32_t i = 0; i < L; ++i) {
        fw = (fw >> 2) | ((uint64_t)base_at(v->refseq, v->ref_accum[r] + i) << (2 * (k - 1)));
        if (i + 1 < k) continue;
        fw &= km;
        uint64_t rc = revcomp(fw, k);
        if (!seen.count(std::min(fw, rc))) return 3;
      }
    }
  }
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> cov(v->num_refs);
  for (uint64_t u = 0; u < v->num_unitigs; ++u) {
    uint64_t ulen = v->uoff[u + 1] - v->uoff[u];
    if (v->ctab_off[u + 1] == v->ctab_off[u]) return 4;
    for (uint64_t i = v->ctab_off[u]; i < v->ctab_off[u + 1]; ++i) {
      uint64_t e = v->ctab[i]; uint32_t t = (uint32_t)(e >> 32); bool fw = (e >> 31) & 1; uint32_t pos = (uint32_t)(e & 0x7FFFFFFF);
      if (pos + ulen > v->ref_len[t]) return 5;
      for (uint64_t j = 0; j < ulen; ++j) {
        uint32_t rb = base_at(v->refseq, v->ref_accum[t] + pos + j);
        uint32_t ub = fw ? base_at(v->useq, v->uoff[u] + j) : 3 - base_at(v->useq, v->uoff[u] + (ulen - 1 - j));
        if (rb != ub) return 6;
      }
      cov[t].push_back({pos, (uint32_t)ulen});
    }
  }
  for (uint32_t r = 0; r < v->num_refs; ++r) {
    if (v->ref_len[r] < k) { if (!cov[r].empty()) return 7; continue; }
    std::sort(cov[r].begin(), cov[r].end());
    uint32_t expect = 0;  // consecutive unitigs overlap by k-1
    for (auto& pr : cov[r]) { if (pr.first != expect) return 8; expect = pr.first + pr.second - (k - 1); }
    if (expect + (k - 1) != v->ref_len[r]) return 9;
  }
  // maximality: rebuild edge sets by brute force and check that no two adjacent unitig ends could merge
  std::unordered_map<uint64_t, uint32_t> info;  // canonical -> bits (R mask 0-3, L mask 4-7, Rterm 8, Lterm 9)
  for (uint32_t r = 0; r < v->num_refs; ++r) {
    uint32_t L = v->ref_len[r]; if (L < k) continue; uint32_t nk = L - k + 1; uint64_t fw = 0;
    for (uint32_t i = 0; i < L; ++i) {
      fw = (fw >> 2) | ((uint64_t)base_at(v->refseq, v->ref_accum[r] + i) << (2 * (k - 1))); if (i + 1 < k) continue; fw &= km;
      uint32_t p = i + 1 - k; uint64_t rc = revcomp(fw, k); bool o1 = fw < rc; uint32_t bits = 0;
      if (p + 1 < nk) {
        uint32_t s = base_at(v->refseq, v->ref_accum[r] + p + k);
        bits |= o1 ? (1u << s) : (1u << (4 + 3 - s));
      } else bits |= o1 ? 256u : 512u;
      if (p > 0) {
        uint32_t q = base_at(v->refseq, v->ref_accum[r] + p - 1);
        bits |= o1 ? (1u << (4 + q)) : (1u << (3 - q));
      } else bits |= o1 ? 512u : 256u;
      info[std::min(fw, rc)] |= bits;
    }
  }
  auto side_break = [&](uint64_t can, bool right) {
    uint32_t inf = info[can];
    uint32_t m = right ? (inf & 15) : ((inf >> 4) & 15);
    bool t = right ? (inf >> 8) & 1 : (inf >> 9) & 1;
    return t || __builtin_popcount(m) != 1;
  };
  for (uint64_t u = 0; u < v->num_unitigs; ++u) {
    uint64_t b = v->uoff[u], ulen = v->uoff[u + 1] - b;
    // interior joins must all be non-breaking; the two outer sides must be breaking
    uint64_t fw = 0, prevc = 0; bool prevo = false;
    for (uint64_t i = 0; i < ulen; ++i) {
      fw = (fw >> 2) | ((uint64_t)base_at(v->useq, b + i) << (2 * (k - 1))); if (i + 1 < k) continue; fw &= km;
      uint64_t rc = revcomp(fw, k); bool o1 = fw < rc; uint64_t c = o1 ? fw : rc; uint64_t p = i + 1 - k;
      if (p == 0) {
        // outer left side must break, unless merging is blocked by a hairpin (neighbour is itself)
        if (!side_break(c, !o1)) {
          // unique predecessor exists: breaking is still legal if the predecessor's facing side breaks or it is a hairpin
          uint32_t inf = info[c]; uint32_t m = (!o1) ? (inf & 15) : ((inf >> 4) & 15); uint32_t bs = __builtin_ctz(m);
          // reconstruct predecessor k-mer in walk orientation
          uint32_t base = o1 ? bs : 3 - bs;  // base preceding fw
          uint64_t pf = ((fw << 2) | base) & km; uint64_t prc = revcomp(pf, k); bool po = pf < prc; uint64_t pcn = po ? pf : prc;
          if (!(side_break(pcn, po) || pcn == c)) return 10;
        }
      } else {
        if (side_break(prevc, prevo) || side_break(c, !o1) || prevc == c) return 11;
      }
      prevc = c; prevo = o1;
    }
  }
  return 0;
}

void orc_map_batch(const orc_index* oi, const sq_quant_opts* o, const sq_read_batch* in, uint32_t nthreads,
                   uint64_t* read_off /*[n+1]*/, sq_aln* alns, uint64_t aln_cap, uint8_t* map_type, sq_map_stats* stats,
                       uint64_t* n_alns_out) {
  Opts op; make_opts(o, op);
  const uint32_t n = in->n; std::vector<FragResult> res(n);
  std::vector<sq_map_stats> sts(std::max(1u, nthreads)); for (auto& s : sts) memset(&s, 0, sizeof(s));
  std::atomic<uint32_t> next(0);
  auto work = [&](uint32_t t) {
    for (;;) {
      uint32_t b = next.fetch_add(256); if (b >= n) break; uint32_t e = std::min(n, b + 256);
      for (uint32_t i = b; i < e; ++i) {
        if (in->paired) { const uint8_t* s1 = in->seq + in->seq_off[2 * i]; uint32_t n1 = (uint32_t)(in->seq_off[2 * i + 1] - in->seq_off[2 * i]); const uint8_t* s2 = in->seq + in->seq_off[2 * i + 1]; uint32_t n2 = (uint32_t)(in->seq_off[2 * i + 2] - in->seq_off[2 * i + 1]);
          map_fragment(oi->ix, op, i, s1, n1, s2, n2, true, res[i], sts[t], nullptr); }
        else {
          const uint8_t* s1 = in->seq + in->seq_off[i];
          uint32_t n1 = (uint32_t)(in->seq_off[i + 1] - in->seq_off[i]);
          map_fragment(oi->ix, op, i, s1, n1, nullptr, 0, false, res[i], sts[t], nullptr);
        }
      }
    }
  };
  if (nthreads <= 1) work(0);
  else {
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  uint64_t tot = 0; read_off[0] = 0;
  for (uint32_t i = 0; i < n; ++i) {
    for (auto& a : res[i].alns) {
      if (tot < aln_cap) alns[tot] = a;
      ++tot;
    }
    read_off[i + 1] = tot;
    if (map_type) map_type[i] = res[i].map_type;
  }
  if (n_alns_out) *n_alns_out = tot;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    uint64_t* d = (uint64_t*)stats;
    for (auto& s : sts) {
      const uint64_t* p = (const uint64_t*)&s;
      for (size_t j = 0; j < sizeof(sq_map_stats) / 8; ++j) d[j] += p[j];
    }
  }
}

// stage taps for one batch (single-threaded): fills caller buffers, returns counts via n_out[4]
void orc_map_taps(const orc_index* oi, const sq_quant_opts* o, const sq_read_batch* in, sq_unimem* um, uint64_t um_cap, sq_mem* mm,
    uint64_t mm_cap,
                  sq_chain* ch, uint64_t ch_cap, sq_cand* cd, uint64_t cd_cap, uint64_t* n_out) {
  Opts op; make_opts(o, op); Taps tp; tp.on = true; sq_map_stats st; memset(&st, 0, sizeof(st)); FragResult fr;
  for (uint32_t i = 0; i < in->n; ++i) {
    if (in->paired) {
      const uint8_t* s1 = in->seq + in->seq_off[2 * i];
      uint32_t n1 = (uint32_t)(in->seq_off[2 * i + 1] - in->seq_off[2 * i]);
      const uint8_t* s2 = in->seq + in->seq_off[2 * i + 1];
      uint32_t n2 = (uint32_t)(in->seq_off[2 * i + 2] - in->seq_off[2 * i + 1]);
      map_fragment(oi->ix, op, i, s1, n1, s2, n2, true, fr, st, &tp);
    }
    else {
      const uint8_t* s1 = in->seq + in->seq_off[i];
      uint32_t n1 = (uint32_t)(in->seq_off[i + 1] - in->seq_off[i]);
      map_fragment(oi->ix, op, i, s1, n1, nullptr, 0, false, fr, st, &tp);
    }
  }
  n_out[0] = tp.unimems.size(); n_out[1] = tp.mems.size(); n_out[2] = tp.chains.size(); n_out[3] = tp.cands.size();
  for (size_t i = 0; i < tp.unimems.size() && i < um_cap; ++i) um[i] = tp.unimems[i];
  for (size_t i = 0; i < tp.mems.size() && i < mm_cap; ++i) mm[i] = tp.mems[i];
  for (size_t i = 0; i < tp.chains.size() && i < ch_cap; ++i) ch[i] = tp.chains[i];
  for (size_t i = 0; i < tp.cands.size() && i < cd_cap; ++i) cd[i] = tp.cands[i];
}

orc_state* orc_state_create(const orc_index* oi, const sq_quant_opts* o) {
  orc_state* s = new orc_state();
  s->S.init(&oi->ix, o);
  return s;
}
void orc_state_free(orc_state* s) { delete s; }
// SPEC D1r: reference-order mode for the next orc_eq_accumulate calls; `draws` (n of them, caller-owned, must outlive the calls) are consumed one per kept alignment
void orc_state_reference_order(orc_state* s, const double* draws, uint64_t n) { s->S.refOrder = true; s->S.draws = draws; s->S.drawCap = n; s->S.drawPos = 0; }
uint64_t orc_state_draws_used(orc_state* s) { return s->S.drawPos; }
// feed one mapped batch (CSR) through the online model in mini-batches, in input order
void orc_eq_accumulate(orc_state* s, uint32_t n, const uint64_t* read_off, const sq_aln* alns, uint64_t num_with_joint_hits) {
  QuantState& S = s->S; uint32_t mb = S.op.o.mini_batch_size ? S.op.o.mini_batch_size : 5000;
  seq_observe(S, n, read_off, alns);
  for (uint64_t r0 = 0; r0 < n; r0 += mb) process_mini_batch(S, read_off, alns, r0, std::min<uint64_t>(n, r0 + mb));
  S.flush_pending();   // a group never straddles two mapped batches
  S.numMappedUB += num_with_joint_hits;
}
// [r5] the same with the reads behind the alignments (alignment-based input with the CIGAR error model)
void orc_eq_accumulate_reads(orc_state* s, uint32_t n, const uint64_t* read_off, const sq_aln* alns, const sq_aln_reads* reads, uint64_t num_with_joint_hits) {
  s->S.reads = reads; orc_eq_accumulate(s, n, read_off, alns, num_with_joint_hits); s->S.reads = nullptr;
}
// [r5] the error model alone, for its pin against the compiled reference (tests/test_alnmodel_pin.py): a one-transcript world; kind 0 = proper pair (records in
// file order; the one with the smaller position is the left one, on a tie the second: host/sam_reader.cpp makes the same choice), 1 / 2 = left / right orphan,
// 3 = single-end.  Returns the alignment's log-likelihood; do_update applies update(p, mass) as a mini-batch of its own
struct orc_errmodel { Index ix; ErrModel em; };
orc_errmodel* orc_errmodel_new(uint32_t bins, const uint8_t* txp_bases, uint32_t txp_len) {
  orc_errmodel* h = new orc_errmodel(); h->em.init(bins); h->ix.ref_len.assign(1, txp_len); h->ix.ref_accum.assign(2, 0); h->ix.ref_accum[1] = txp_len; h->ix.refseq.assign((txp_len + 31) / 32 + 1, 0);
  for (uint32_t i = 0; i < txp_len; ++i) h->ix.refseq[i >> 5] |= (uint64_t)(txp_bases[i] & 3) << ((i & 31) * 2);
  return h;
}
void orc_errmodel_free(orc_errmodel* h) { delete h; }
double orc_errmodel_eval(orc_errmodel* h, int kind, int32_t pos1, const uint32_t* cig1, uint32_t n1, const uint8_t* seq1, int32_t len1,
                         int32_t pos2, const uint32_t* cig2, uint32_t n2, const uint8_t* seq2, int32_t len2, int do_update, double p, double mass) {
  ErrRec r[2]; bool have[2] = {false, false}; const ErrRec a{pos1, cig1, n1, seq1, len1}, b{pos2, cig2, n2, seq2, len2};
  if (kind == 0) { const bool first_left = pos1 < pos2; r[first_left ? 0 : 1] = a; r[first_left ? 1 : 0] = b; have[0] = have[1] = true; }
  else { const int k = kind == 2 ? 1 : 0; r[k] = a; have[k] = true; }
  double ll = 0.0, bg = 0.0;
  for (int k = 0; k < 2; ++k) if (have[k]) { double f, g; err_like_rec(h->em, k, h->ix, 0, r[k], &f, &g); ll += f; bg += g; }
  if (do_update && mass != SQ_LOG_0) {
    const uint64_t q = sq_to_fixed(sq_exp(p), SQ_MFRAC_BITS);
    for (int k = 0; k < 2; ++k) if (have[k]) {
      std::vector<uint32_t> cells; err_update_rec(h->em.bins, h->ix, 0, r[k], cells); std::map<uint32_t, uint64_t> inc, rows;
      for (uint32_t c : cells) { inc[c] += q; rows[c / ErrModel::NS] += q; }
      for (auto& cq : inc) h->em.cell[k][cq.first] = sq_log_add(h->em.cell[k][cq.first], mass + sq_log(sq_from_fixed(cq.second, SQ_MFRAC_BITS)));
      for (auto& rq : rows) h->em.row[k][rq.first] = sq_log_add(h->em.row[k][rq.first], mass + sq_log(sq_from_fixed(rq.second, SQ_MFRAC_BITS)));
    }
  }
  return ll - bg;
}
// the error model's matrices, for the tests: log-space cells [2][bins * 82 * 82] and row sums [2][bins * 82]
uint32_t orc_state_err_model(orc_state* s, double* cells, double* rows) {
  const ErrModel& E = s->S.em; if (!s->S.errOn) return 0;
  if (cells) for (int sd = 0; sd < 2; ++sd) memcpy(cells + (size_t)sd * E.cell[0].size(), E.cell[sd].data(), E.cell[sd].size() * 8);
  if (rows) for (int sd = 0; sd < 2; ++sd) memcpy(rows + (size_t)sd * E.row[0].size(), E.row[sd].data(), E.row[sd].size() * 8);
  return E.bins;
}
// finalisation when burn-in was never reached (SalmonQuantify.cpp:2734-2745)
void orc_state_finish(orc_state* s) { QuantState& S = s->S; if (!S.burnedIn) { compute_eff_lengths(S.fld, S.ix->ref_len, S.logEffLen); } }
// SPEC §MG: fold the state of rank `src` into `dst` (call in rank order 1, 2, ... on rank 0's state): class counts and fixed-point
// weight sums add, unique / total counts and fragment counters add, masses combine by logAdd(dst, src); the fragment-length
// distribution and the effective lengths stay rank 0's
void orc_state_merge(orc_state* dst, const orc_state* src) {
  QuantState& D = dst->S; const QuantState& R = src->S;
  for (auto& kv : R.eq) {
    EqVal& ev = D.eq[kv.first];
    if (ev.wq.empty()) ev.wq.assign(kv.second.wq.size(), 0);
    ev.count += kv.second.count;
    for (size_t i = 0; i < ev.wq.size(); ++i) ev.wq[i] += kv.second.wq[i];
  }
  for (size_t t = 0; t < D.mass.size(); ++t) { D.mass[t] = sq_log_add(D.mass[t], R.mass[t]); D.uniq[t] += R.uniq[t]; D.total[t] += R.total[t]; }
  D.numObserved += R.numObserved; D.numAssigned += R.numAssigned; D.numMappedUB += R.numMappedUB; D.numCompat += R.numCompat;
  for (int f = 0; f < 64; ++f) D.libCounts[f] += R.libCounts[f];
}
// SPEC MG, end of the shared burn-in prefix on a rank other than 0: what the prefix added is forgotten, what it taught is kept; the masses move
// into the prior term (logAdd(prior, mass) is unchanged), `mass` restarts and ends as the rank's own increments
void orc_state_drop_counts(orc_state* s) {
  QuantState& S = s->S;
  for (size_t t = 0; t < S.mass.size(); ++t) { S.priorMass[t] = sq_log_add(S.priorMass[t], S.mass[t]); S.mass[t] = SQ_LOG_0; S.uniq[t] = 0; S.total[t] = 0; }
  S.eq.clear(); S.numObserved = 0; S.numAssigned = 0; S.numMappedUB = 0; S.numCompat = 0;
  for (auto& v : S.libCounts) v = 0;
  memset(S.gcObs, 0, sizeof(S.gcObs)); memset(S.posObs, 0, sizeof(S.posObs)); memset(S.seqObs, 0, sizeof(S.seqObs)); S.seqSamples = 0;
}
void orc_state_summary(orc_state* s, sq_model_summary* m) {
  m->lib_format_id = (uint32_t)(s->S.op.o.lib_type | (s->S.op.o.lib_orientation << 1) | (s->S.op.o.lib_strand << 3)); m->lib_detected = s->S.detected ? 1u : 0u;
  m->num_observed = s->S.numObserved;
  m->num_assigned = s->S.numAssigned;
  m->num_mapped_ub = s->S.numMappedUB;
  m->burned_in = s->S.burnedIn;
  m->num_compatible = s->S.numCompat;
}
void orc_state_lib_counts(orc_state* s, uint64_t* out64) { for (int i = 0; i < 64; ++i) out64[i] = s->S.libCounts[i]; }
void orc_state_fetch(orc_state* s, double* log_mass, uint64_t* uniq, uint64_t* total, double* log_eff_len, double* fld_logpmf) {
  QuantState& S = s->S; size_t M = S.mass.size();
  for (size_t t = 0; t < M; ++t) {
    if (log_mass) log_mass[t] = S.mass[t];
    if (uniq) uniq[t] = S.uniq[t];
    if (total) total[t] = S.total[t];
    if (log_eff_len) log_eff_len[t] = S.logEffLen[t];
  }
  if (fld_logpmf) for (int i = 0; i <= 1000; ++i) fld_logpmf[i] = S.fld.pmf(i);
}
// label hash shared with the product (two independent 64-bit mixes over tids+bins)
static void label_hash(const std::vector<uint32_t>& lab, uint64_t* h1, uint64_t* h2) {
  uint64_t a = 0x243F6A8885A308D3ULL ^ lab.size(), b = 0x13198A2E03707344ULL + lab.size();
  for (uint32_t x : lab) {
    a = sq_mix64(a ^ (uint64_t)x) + 0x9E3779B97F4A7C15ULL;
    b = sq_mix64(b + (uint64_t)x * 0xD6E8FEB86659FD93ULL) ^ (b >> 29);
  }
  *h1 = sq_mix64(a); *h2 = sq_mix64(b);
  if (*h1 == ~0ULL) *h1 = ~0ULL - 1;  // ~0 marks an empty slot in the device table
  if (*h2 == 0) *h2 = 1;              // 0 marks "second hash not published yet"
}
// eq table in canonical order (ascending (first transcript id, h1, h2) — SPEC §D7); call with NULL arrays to get sizes
void orc_eq_finish(orc_state* s, sq_eq_table* out) {
  QuantState& S = s->S;
  struct Row { uint64_t h1, h2; const std::vector<uint32_t>* lab; const EqVal* v; };
  std::vector<Row> rows; rows.reserve(S.eq.size()); uint64_t L = 0;
  for (auto& kv : S.eq) {
    Row r;
    label_hash(kv.first, &r.h1, &r.h2);
    r.lab = &kv.first;
    r.v = &kv.second;
    rows.push_back(r);
    L += kv.second.wq.size();
  }
  std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { const uint32_t ta = (*a.lab)[0],
      tb = (*b.lab)[0]; if (ta != tb) return ta < tb; return a.h1 < b.h1 || (a.h1 == b.h1 && a.h2 < b.h2); });
  out->num_classes = rows.size(); out->num_labels = L;
  if (!out->off) return;
  uint64_t p = 0;
  for (size_t c = 0; c < rows.size(); ++c) {
    const Row& r = rows[c];
    size_t n = r.v->wq.size();
    out->off[c] = p;
    out->count[c] = r.v->count;
    if (out->h1) out->h1[c] = r.h1;
    if (out->h2) out->h2[c] = r.h2;
    double sum = 0.0; for (size_t i = 0; i < n; ++i) sum += sq_from_fixed(r.v->wq[i], SQ_WFRAC_BITS);
    double norm = 1.0 / sum;  // TGValue::normalizeAux (EquivalenceClassBuilder.hpp:116-125)
    for (size_t i = 0; i < n; ++i) {
      out->tid[p + i] = (*r.lab)[i];
      out->w[p + i] = sq_from_fixed(r.v->wq[i], SQ_WFRAC_BITS) * norm;
      if (out->wq) out->wq[p + i] = r.v->wq[i];
      if (out->bins) out->bins[p + i] = r.lab->size() > n ? (*r.lab)[n + i] : 0;
    }
    p += n;
  }
  out->off[rows.size()] = p;
}

// normalizeAlphas (SalmonUtils.cpp:461-529) + TranscriptCluster::projectToPolytope (TranscriptCluster.hpp:46-102)
// over clusters = connected components of the eq-class labels; members in ascending tid (SPEC §D5).
void orc_normalize_alphas(uint32_t M, const sq_eq_table* eq, const double* log_mass, const uint64_t* uniq, const uint64_t* total,
    double* projected) {
  std::vector<uint32_t> parent(M); for (uint32_t i = 0; i < M; ++i) parent[i] = i;
  auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
  for (uint64_t c = 0; c < eq->num_classes; ++c) for (uint64_t i = eq->off[c] + 1; i < eq->off[c + 1]; ++i) {
    uint32_t a = find(eq->tid[eq->off[c]]), b = find(eq->tid[i]);
    if (a != b) {
      if (a < b) parent[b] = a;
      else parent[a] = b;
    }
  }
  std::vector<double> hits(M, 0.0);
  for (uint64_t c = 0; c < eq->num_classes; ++c) hits[find(eq->tid[eq->off[c]])] += (double)eq->count[c];
  std::vector<std::vector<uint32_t>> members(M);
  for (uint32_t t = 0; t < M; ++t) members[find(t)].push_back(t);
  for (uint32_t r = 0; r < M; ++r) {
    auto& mem = members[r]; if (mem.empty()) continue;
    double logClusterMass = SQ_LOG_0; for (uint32_t t : mem) logClusterMass = sq_log_add(logClusterMass, log_mass[t]);
    double logClusterCount = hits[r] > 0 ? sq_log(hits[r]) : SQ_LOG_0;  // salmon::math-free std::log(0) = -inf in the reference; exp(-inf)=0 either way
    bool need = false;
    for (uint32_t t : mem) {
      if (log_mass[t] == SQ_LOG_0) projected[t] = 0;
      else {
        projected[t] = (hits[r] > 0) ? sq_exp(log_mass[t] - logClusterMass + logClusterCount) : 0.0;
        need |= projected[t] > (double)total[t] || projected[t] < (double)uniq[t];
      }
    }
    if (mem.size() > 1 && need) {
      double clusterCounts = hits[r]; std::vector<uint8_t> bound(mem.size(), 0); size_t round = 0;
      for (;;) {
        double unb = 0.0, bnd = 0.0;
        for (size_t i = 0; i < mem.size(); ++i) { uint32_t t = mem[i];
          if (projected[t] > (double)total[t]) {
            projected[t] = (double)total[t];
            bound[i] = 1;
          } else if (projected[t] < (double)uniq[t]) {
            projected[t] = (double)uniq[t];
            bound[i] = 1;
          }
          if (bound[i]) bnd += projected[t]; else unb += projected[t]; }
        if (std::fabs(unb + bnd - clusterCounts) <= 0.375e-10) break;
        if (unb == 0) { std::fill(bound.begin(), bound.end(), 0); unb = bnd; bnd = 0; }
        double nz = (clusterCounts - bnd) / unb;
        for (size_t i = 0; i < mem.size(); ++i) if (!bound[i]) projected[mem[i]] *= nz;
        if (++round > 5000) break;
      }
    }
  }
}

void orc_state_seq_observed(orc_state* s, uint64_t* fw576, uint64_t* rc576, uint64_t* nsamples) { memcpy(fw576, s->S.seqObs[0], 576 * 8); memcpy(rc576, s->S.seqObs[1], 576 * 8); if (nsamples) *nsamples = s->S.seqSamples; }
int orc_bias_seq_eff_lengths(const orc_index* oi, int gc, const double* gc_obs, const uint64_t* seq_fw, const uint64_t* seq_rc, const double* log_pmf, uint32_t M, const double* alphas,
                             const double* eff_in, double* eff_out, double* models4x576) {
  SeqBiasOut o; int rc = bias_seq_eff_lengths(oi->ix, gc != 0, gc_obs, seq_fw, seq_rc, log_pmf, M, alphas, eff_in, eff_out, &o);
  if (models4x576) { memcpy(models4x576, o.exp_fw, 576 * 8); memcpy(models4x576 + 576, o.exp_rc, 576 * 8); memcpy(models4x576 + 1152, o.obs_fw, 576 * 8); memcpy(models4x576 + 1728, o.obs_rc, 576 * 8); }
  return rc;
}
void orc_state_pos_observed(orc_state* s, double* out200) { for (int i = 0; i < 200; ++i) out200[i] = sq_from_fixed(s->S.posObs[i / 100][i % 100], 32); }
int orc_length_classes(const orc_index* oi, uint32_t* quant5, uint8_t* cls) { std::vector<uint32_t> q; std::vector<uint8_t> c; length_classes(oi->ix, q, c);
  for (size_t i = 0; i < q.size() && i < 5; ++i) quant5[i] = q[i]; memcpy(cls, c.data(), c.size()); return (int)q.size(); }
int orc_pos_bin(int32_t pos, uint32_t len) { return pos_bin(pos, len); }
void orc_pos_project(const double* mass20, int32_t len, double* out, double* norm20) { PosSpline S; pos_finalize(mass20, S, norm20); for (int32_t p = 0; p < len; ++p) out[p] = pos_weight(S, p, len); }
void orc_spline_eval(const double* xs, const double* ys, int n, const double* q, int nq, double* out) { PosSpline S; pos_spline_build(xs, ys, n, S); for (int i = 0; i < nq; ++i) out[i] = pos_spline_eval(S, n, q[i]); }
// every bias combination that needs the per-position sweep: seq_fw / pos_obs may be NULL
int orc_bias_eff_lengths(const orc_index* oi, int gc, const double* gc_obs, const uint64_t* seq_fw, const uint64_t* seq_rc, const double* pos_obs, uint32_t threads,
                         const double* log_pmf, uint32_t M, const double* alphas, const double* eff_in, double* eff_out, double* pos_models_out /*[4][100] or NULL*/) {
  PosIn pi{pos_obs, threads}; PosOut po;
  int rc = bias_seq_eff_lengths(oi->ix, gc != 0, gc_obs, seq_fw, seq_rc, log_pmf, M, alphas, eff_in, eff_out, nullptr, pos_obs ? &pi : nullptr, &po);
  if (pos_obs && pos_models_out) { memcpy(pos_models_out, po.obs_norm, 1600); memcpy(pos_models_out + 200, po.exp_norm, 1600); }
  return rc;
}
int orc_em_optimize_bias_pos(const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, const orc_index* oi, const double* gc_obs, const uint64_t* seq_fw, const uint64_t* seq_rc,
                             const double* pos_obs, uint32_t threads, const double* log_pmf, double* alpha_out, double* eff_out, sq_em_report* rep) {
  PosIn pi{pos_obs, threads}; return em_optimize_gc(eq, txp, o, oi->ix, gc_obs, log_pmf, alpha_out, eff_out, rep, seq_fw, seq_rc, pos_obs ? &pi : nullptr); }
void orc_state_gc_observed(orc_state* s, double* out75) { for (int i = 0; i < 75; ++i) out75[i] = sq_from_fixed(s->S.gcObs[i], 32); }

int orc_bias_gc_eff_lengths(const orc_index* oi, const double* gc_obs, const double* log_pmf, uint32_t M, const double* alphas, const double* eff_in,
                            double* eff_out, double* bias_row0_out) { return bias_gc_eff_lengths(oi->ix, gc_obs, log_pmf, M, alphas, eff_in, eff_out, bias_row0_out); }
int orc_em_optimize_gc(const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, const orc_index* oi, const double* gc_obs, const double* log_pmf,
                       double* alpha_out, double* eff_out, sq_em_report* rep) { return em_optimize_gc(eq, txp, o, oi->ix, gc_obs, log_pmf, alpha_out, eff_out, rep); }
int orc_em_optimize_bias(const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, const orc_index* oi, const double* gc_obs /* or NULL */, const uint64_t* seq_fw, const uint64_t* seq_rc,
                         const double* log_pmf, double* alpha_out, double* eff_out, sq_em_report* rep) { return em_optimize_gc(eq, txp, o, oi->ix, gc_obs, log_pmf, alpha_out, eff_out, rep, seq_fw, seq_rc); }

int orc_em_optimize(const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, double* alpha_out, sq_em_report* rep) {
  return em_optimize(eq, txp, o, alpha_out, rep);
}
int orc_bootstrap(const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, uint32_t B, uint64_t seed, uint64_t num_mapped,
    double* out) {
  return bootstrap(eq, txp, o, B, seed, num_mapped, out);
}
int orc_gibbs(const sq_eq_table* eq, const sq_txp_in* txp, const sq_gibbs_opts* go, const double* alpha_init, uint32_t S, uint64_t seed,
    uint64_t num_mapped,
    double* out) {
  return gibbs(eq, txp, go, alpha_init, S, seed, num_mapped, out);
}
int orc_em_steps(const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, const double* alpha_in, uint32_t iters,
    double* alpha_out) {
  EMProblem P; em_setup(P, eq, txp, o); std::vector<double> a(alpha_in, alpha_in + P.M), b(P.M), th(P.M), inv(P.E);
  for (uint32_t i = 0; i < iters; ++i) { em_step(P, o, a, b, th, inv); a.swap(b); }
  for (uint32_t i = 0; i < P.M; ++i) alpha_out[i] = a[i];
  return 0;
}
// [r4] the combined weights em_setup forms (CollapsedEMOptimizer.cpp:830-873), label by label: tests/test_em_pin.py hands them to the reference's EMUpdate_
int orc_em_combined_weights(const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, double* cw_out) { EMProblem P; em_setup(P, eq, txp, o); memcpy(cw_out, P.cw.data(), P.cw.size() * 8); return 0; }
// multi-threaded EM timing leg for the CPU baseline: class pass + transcript pass over thread ranges
double orc_em_time_iters(const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, uint32_t iters, uint32_t nthreads);

double orc_canonical_sum(const double* x, uint64_t n) { return canonical_sum(std::vector<double>(x, x + n)); }
double orc_exp(double x) { return sq_exp(x); }
double orc_log(double x) { return sq_log(x); }
double orc_digamma(double x) { return sq_digamma(x); }
double orc_log_add(double x, double y) { return sq_log_add(x, y); }
int orc_compatible_pe(int et, int eo, int es, int ot, int oo, int os) {
  return compatible_hit_pe(LibFmt{(uint8_t)et, (uint8_t)eo, (uint8_t)es}, LibFmt{(uint8_t)ot, (uint8_t)oo, (uint8_t)os});
}
int orc_compatible_se(int et, int eo, int es, int fwd, int ms) {
  return compatible_hit_se(LibFmt{(uint8_t)et, (uint8_t)eo, (uint8_t)es}, fwd != 0, (uint8_t)ms);
}
int orc_format_id(int t, int o, int s) { return format_id(LibFmt{(uint8_t)t, (uint8_t)o, (uint8_t)s}); }
int orc_dp_align(const sq_quant_opts* o, const uint8_t* q, int n, const uint8_t* t, int tl, int mode) {
  Opts op;
  make_opts(o, op);
  return dp_align(op, q, n, t, tl, mode);
}
// a5: the infix aligner alone (codes 0..3, other values match nothing); returns 1 and (distance, start, end) or 0
int orc_infix_align(const uint8_t* q, int n, const uint8_t* t, int m, int k, int* ed, int* start, int* end) {
  return infix_align(q, n, t, m, k, ed, start, end) ? 1 : 0;
}
void orc_fld_prior(double mu, double sd, double* log_hist_1001, double* tot) {
  FLD f;
  f.init(mu, sd);
  for (int i = 0; i <= 1000; ++i) log_hist_1001[i] = f.hist[i];
  *tot = f.totMass;
}
// [r4] the FLD on its own (tests/test_fld_pin.py holds it against the reference's FragmentLengthDistribution.cpp compiled into oracle/_ref/libfld_ref.so):
// counts_1001[len] fragments of every length enter with the same log mass, as one mini-batch does it (SPEC §D3)
void* orc_fld_new(double mu, double sd) { FLD* f = new FLD(); f->init(mu, sd); return f; }
void orc_fld_free(void* h) { delete (FLD*)h; }
void orc_fld_apply(void* h, const uint32_t* counts_1001, double log_mass) { FLD* f = (FLD*)h; std::vector<uint32_t> c(counts_1001, counts_1001 + 1001); for (int l = 0; l <= 1000; ++l) if (c[l] && (uint32_t)l < f->minLen) f->minLen = (uint32_t)l; f->apply_counts(c, log_mass); }
void orc_fld_cache(void* h) { ((FLD*)h)->cache(); }
void orc_fld_pmf(void* h, double* out, uint32_t n) { FLD* f = (FLD*)h; for (uint32_t i = 0; i < n; ++i) out[i] = f->pmf(i); }
void orc_fld_cmf(void* h, double* out, uint32_t n) { FLD* f = (FLD*)h; for (uint32_t i = 0; i < n; ++i) out[i] = f->cmf(i); }
uint32_t orc_fld_min(void* h) { return ((FLD*)h)->minLen; }
void orc_fld_eff_lengths(void* h, const uint32_t* ref_len, uint32_t n, double* log_eff_len) { std::vector<uint32_t> rl(ref_len, ref_len + n); std::vector<double> out(n); compute_eff_lengths(*(FLD*)h, rl, out); memcpy(log_eff_len, out.data(), (size_t)n * 8); }
// the two tables QuantState::init builds beside the FLD, then the question processMiniBatch asks per orphan / single-end alignment
double orc_fld_ambig_prob(void* h, int single_end, int burned, int fwd, int32_t pos, int32_t rlen, int32_t tlen) {
  const FLD& f = *(FLD*)h; std::vector<double> live(1001), amb(1001); double c1 = SQ_LOG_0, c2 = SQ_LOG_0;
  for (int j = 0; j <= 1000; ++j) { c1 = sq_log_add(c1, f.hist[j]); live[j] = c1 - f.totMass; c2 = sq_log_add(c2, SQ_LOG_EPSILON); amb[j] = c2; }
  return ambig_frag_prob(f, live, amb, single_end != 0, burned != 0, fwd != 0, pos, rlen, tlen);
}
double orc_forgetting_mass(double ff, uint64_t b) { QuantState S; S.op.o.forgetting_factor = ff; return S.forgetting_mass(b); }

// ---- [r4] the bias models' arithmetic on its own, so that tests/test_models_pin.py can hold it against the reference's SBModel.cpp / GCFragModel.hpp
// compiled into oracle/_ref/libmodels_ref.so ----
// counts[9][64] (the cells sb_cell addresses, WITHOUT the 1e-10 every cell of the reference starts with: added here) -> log probabilities
void orc_sb_normalize(const double* counts576, double* logp576) { double c[576]; for (int i = 0; i < 576; ++i) c[i] = counts576[i] + 1e-10; sb_normalize(c, logp576); }
uint32_t orc_sb_cell(uint32_t ctx18, int pos) { return sb_cell(ctx18, pos); }
uint32_t orc_sb_rc(uint32_t ctx18) { return sb_rc(ctx18); }
double orc_sb_eval(const double* logp576, uint32_t ctx18) { return sb_eval(logp576, ctx18); }
int orc_gc_frag_bin(int frag_frac) { return gc_frag_bin(frag_frac); }
int orc_gc_ctx_bin(int ctx_frac) { return gc_ctx_bin(ctx_frac); }
// normalize (prior 0.1) both models and clamp the ratio to [1/1000, 1000]: the lines of bias_gc_eff_lengths / bias_seq_eff_lengths above
void orc_gc_ratio(const double* obs75, const double* exp75, double* out75) {
  auto normalize = [](const double* in, double* out) { double rowMass = 0.0; for (int c = 0; c < 25; ++c) rowMass += (0.1 + in[c]);
    if (rowMass > 0.0) { double norm = 1.0 / rowMass; for (int c = 0; c < 25; ++c) out[c] = (0.1 + in[c]) * norm; } else for (int c = 0; c < 25; ++c) out[c] = in[c]; };
  for (int r = 0; r < 3; ++r) { double on[25], en[25]; normalize(obs75 + 25 * r, on); normalize(exp75 + 25 * r, en);
    for (int c = 0; c < 25; ++c) { double rat = on[c] / en[c]; if (rat > 1000.0) rat = 1000.0; if (rat < 1.0 / 1000.0) rat = 1.0 / 1000.0; out75[r * 25 + c] = rat; } }
}
}  // extern "C"

// threads split classes / transcripts statically; per-iteration barrier via join (coarse but fair)
double orc_em_time_iters(const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, uint32_t iters, uint32_t nthreads) {
  EMProblem P; em_setup(P, eq, txp, o); const uint32_t M = P.M;
  std::vector<double> a(M, 100.0), b(M), th(M), inv(P.E);
  auto t0 = std::chrono::steady_clock::now();
  for (uint32_t it = 0; it < iters; ++it) {
    if (o->use_vbem) { std::vector<double> ap(M); for (uint32_t i = 0; i < M; ++i) ap[i] = a[i] + P.prior[i]; double ln = sq_digamma(canonical_sum(ap));
      auto f1 = [&](uint32_t t) {
        for (uint32_t i = t; i < M; i += nthreads) th[i] = ap[i] > 1e-10 ? sq_exp(sq_digamma(ap[i]) - ln) : 0.0;
      };
      std::vector<std::thread> tv; for (uint32_t t = 0; t < nthreads; ++t) tv.emplace_back(f1,
          t); for (auto& x : tv) x.join(); } else th = a;
    auto f2 = [&](uint32_t t) { uint64_t c0 = P.E * t / nthreads, c1 = P.E * (t + 1) / nthreads;
      for (uint64_t c = c0; c < c1; ++c) { uint64_t x = P.off[c],
          y = P.off[c + 1]; if (y - x <= 1) { inv[c] = 0; continue; } double d = 0; for (uint64_t i = x; i < y; ++i) { double v = th[P.tid[i]]; if (!o->use_vbem ||
          v > 0) d += v * P.cw[i]; } inv[c] = d <= 2.2250738585072014e-308 ? 0.0 : (double)P.count[c] / d; } };
    { std::vector<std::thread> tv; for (uint32_t t = 0; t < nthreads; ++t) tv.emplace_back(f2, t); for (auto& x : tv) x.join(); }
    auto f3 = [&](uint32_t t) { uint32_t m0 = (uint64_t)M * t / nthreads, m1 = (uint64_t)M * (t + 1) / nthreads;
      for (uint32_t m = m0; m < m1; ++m) { double acc = 0,
          v0 = th[m]; for (uint64_t d = P.t_off[m]; d < P.t_off[m + 1]; ++d) { uint64_t c = P.t_cls[d]; if (P.off[c + 1] - P.off[c] == 1) { acc += (double)P.count[c]; continue; } if (inv[c] == 0.0 ||
          (o->use_vbem && !(v0 > 0))) continue; acc += v0 * P.cw[P.t_pos[d]] * inv[c]; } b[m] = acc; } };
    { std::vector<std::thread> tv; for (uint32_t t = 0; t < nthreads; ++t) tv.emplace_back(f3, t); for (auto& x : tv) x.join(); }
    a.swap(b);
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// the selection of a fragment's alignments from its candidates' scores (select_hits): what tests/test_selection_pin.py holds to the reference's
// updateRefMappings / haveOnlyDecoyMappings / filterAndCollectAlignments compiled from include/salmon/internal/quant/SalmonMappingUtils.hpp.
// info: [0] bestScore, [1] bestDecoyScore, [2] 1 when the fragment has only decoy mappings.  Returns the number of alignments kept (their candidate indices, in
// emission order, and estAlnProb).
extern "C" uint32_t orc_select_hits(uint32_t n, const uint32_t* tid, const int32_t* score, const uint8_t* compat, const uint8_t* scored, uint32_t first_decoy, double decoy_threshold,
                                    int hard_filter, double score_exp, double min_aln_prob, uint32_t* kept_out, double* prob_out, int32_t* info) {
  Selection S; select_hits(tid, score, compat, scored, n, first_decoy, decoy_threshold, hard_filter != 0, score_exp, min_aln_prob, S);
  for (size_t k = 0; k < S.kept.size(); ++k) { kept_out[k] = (uint32_t)S.kept[k]; prob_out[k] = S.prob[k]; }
  info[0] = S.bestScore; info[1] = S.bestDecoy; info[2] = S.onlyDecoy ? 1 : 0;
  return (uint32_t)S.kept.size();
}

