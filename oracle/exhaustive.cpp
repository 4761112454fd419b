// oracle/exhaustive.cpp — TEST INFRASTRUCTURE.  An exhaustive cross-check for rows a1–a4 of SURVEY.md §8(a) (k-mer seeding, uni-MEMs,
// chaining, joining, selective-alignment scoring), which live in pufferfish and cannot be pinned to reference vectors.
//
// It shares nothing with oracle.cpp or the product: no index, no k-mers, no seeds, no chains, no band.  Every read end, on both strands,
// is aligned against EVERY position of EVERY transcript by a full affine-gap dynamic programme (query consumed end to end, reference ends
// free) with the scoring salmon configures for selective alignment (include/salmon/internal/quant/SalmonMappingUtils.hpp:168-206:
// match 2, mismatch -4, gap open 6 + extend 2 per base, minScoreFraction 0.65); the complete set of valid end alignments is then
// paired and filtered by the rules the in-tree reference applies to whatever the aligner returns (:225-485: best hit per transcript,
// estAlnProb = exp(-scoreExp (best - score)) >= minAlnProb; pairs need both ends on one transcript, opposite strands, no dovetail,
// fragment length in (0, fldMax]; orphans only when no pair exists).  What comes out per fragment is the set of transcripts a
// seed-chain-extend heuristic can at best reproduce; tests compare the HIP path's and the checker's labels with it and SPEC.md lists
// the classes of disagreement.
//
// 16 reference positions are processed per instruction (int16 lanes, GCC vector extensions): lane l walks its own slice of the
// concatenated transcripts (with a warm-up overlap), all lanes against the same read.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {
typedef int16_t v16 __attribute__((vector_size(32)));
static inline v16 vmax(v16 a, v16 b) { return (a > b) ? a : b; }
static inline v16 splat(int x) { v16 r; for (int i = 0; i < 16; ++i) r[i] = (int16_t)x; return r; }

struct Hit { uint32_t tid; int32_t start, end; int32_t score; };   // [start, end] on the transcript, 0-based, end inclusive

struct Text {
  std::vector<uint8_t> code;          // concatenated transcripts, sentinel 5 between them (and at both ends)
  std::vector<uint64_t> tstart;       // position of transcript t's first base in `code`
  uint32_t tid_of(uint64_t p) const { return (uint32_t)(std::upper_bound(tstart.begin(), tstart.end(), p) - tstart.begin() - 1); }
};
static inline uint8_t enc(uint8_t c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }

struct Scoring { int ma, mp, go, ge; };

// all alignments of query q (codes 0..3, 4 = N: mismatches everything) against the text with score >= min_score, one Hit per end position
// (start is filled in later, for the hits that survive thinning)
static void scan(const Text& T, const std::vector<uint8_t>& q, const Scoring& sc, int min_score, std::vector<Hit>& out) {
  const int m = (int)q.size(); if (m == 0) return;
  const int64_t N = (int64_t)T.code.size();
  const int OV = m + 72;                       // warm-up: a valid alignment spans at most m + (2m - min_score - go) / ge reference bases
  const int BLK = 1 << 15;                     // columns a lane reports per super-block
  struct Cell { v16 H, E; };
  std::vector<Cell> col_(m + 1); Cell* C = col_.data();
  const v16 v_ma = splat(sc.ma), v_mp = splat(sc.mp), v_go = splat(sc.go), v_ge = splat(sc.ge), v_sent = splat(-8000), v_min = splat(min_score);
  const v16 neg = splat(-9000), zero = splat(0), five = splat(5);
  std::vector<v16> qv_(m + 1); v16* qv = qv_.data(); for (int i = 1; i <= m; ++i) qv[i] = splat(q[i - 1] <= 3 ? q[i - 1] : 100);
  for (int64_t base = 0; base < N; base += (int64_t)16 * BLK) {
    // lane l: columns [base + l*BLK - OV, base + (l+1)*BLK), reporting from base + l*BLK on
    for (int i = 0; i <= m; ++i) { C[i].H = (i == 0) ? zero : splat(-(sc.go + i * sc.ge)); C[i].E = neg; }
    const int ncol = (int)std::min<int64_t>(BLK, N - base) + OV;      // the last super-block may be short: lanes past the text see sentinels
    for (int col = 0; col < ncol; ++col) {
      v16 rc;
      for (int l = 0; l < 16; ++l) { const int64_t p = base + (int64_t)l * BLK - OV + col; rc[l] = (p >= 0 && p < N) ? T.code[p] : 5; }
      const v16 mis = (rc == five) ? v_sent : v_mp;
      v16 diagH = C[0].H;          // row 0 is 0 in every column: a fresh start anywhere
      v16 F = neg, Hup = zero;
      for (int i = 1; i <= m; ++i) {
        const v16 s = (rc == qv[i]) ? v_ma : mis;
        const v16 Hold = C[i].H;
        const v16 Ei = vmax(C[i].E, Hold - v_go) - v_ge;       // gap that consumes reference (previous column, same row)
        F = vmax(F, Hup - v_go) - v_ge;                         // gap that consumes query (row above, same column)
        const v16 Hn = vmax(vmax(diagH + s, Ei), F);
        diagH = Hold; Hup = Hn; C[i].H = Hn; C[i].E = Ei;
      }
      if (col >= OV) {
        const v16 ok = (C[m].H >= v_min);
        bool any = false; for (int l = 0; l < 16; ++l) any |= ok[l] != 0;
        if (any) for (int l = 0; l < 16; ++l) if (ok[l]) {
          const int64_t p = base + (int64_t)l * BLK - OV + col; if (p < 0 || p >= N || T.code[p] > 4) continue;
          const uint32_t t = T.tid_of((uint64_t)p);
          out.push_back({t, -1, (int32_t)(p - (int64_t)T.tstart[t]), (int32_t)C[m].H[l]});
        }
      }
    }
  }
}

// start of an optimal alignment that ends at h.end with score h.score: the same recurrence run backwards from the end (scalar; a hit is rare)
static void find_start(const Text& T, const std::vector<uint8_t>& q, const Scoring& sc, Hit& h) {
  const int m = (int)q.size(); const int W = m + 80; const int NEG = -100000;
  const int64_t pend = (int64_t)T.tstart[h.tid] + h.end;
  std::vector<int> H(m + 1), E(m + 1), Hn(m + 1), En(m + 1);
  // reversed query against the reference read leftwards from pend; row i = i reversed-query bases consumed; the alignment must START (in
  // reverse) exactly at pend: column 0 is not free
  for (int i = 0; i <= m; ++i) { H[i] = i == 0 ? 0 : -(sc.go + i * sc.ge); E[i] = NEG; }
  h.start = h.end;   // fallback
  for (int j = 1; j <= W; ++j) {
    const int64_t p = pend - (j - 1); if (p < (int64_t)T.tstart[h.tid]) break;
    const uint8_t r = T.code[p];
    Hn[0] = NEG; En[0] = NEG;   // the reverse alignment cannot skip reference bases before its first query base: the end is fixed
    int F = NEG;
    for (int i = 1; i <= m; ++i) {
      const uint8_t qc = q[m - i];
      const int s = (qc <= 3 && qc == r) ? sc.ma : sc.mp;
      En[i] = std::max(E[i], H[i] - sc.go) - sc.ge;
      F = std::max(F, Hn[i - 1] - sc.go) - sc.ge;
      Hn[i] = std::max(std::max(H[i - 1] + s, En[i]), F);
    }
    H.swap(Hn); E.swap(En);
    if (H[m] == h.score) { h.start = h.end - (j - 1); return; }     // shortest reference span that reaches the score
  }
}

// keep, per transcript, the hits that are local maxima over their end position (a shifted copy of an alignment scores less and adds nothing)
static void thin(std::vector<Hit>& h) {
  std::sort(h.begin(), h.end(), [](const Hit& a, const Hit& b) { return a.tid != b.tid ? a.tid < b.tid : a.end < b.end; });
  std::vector<Hit> o;
  for (size_t i = 0; i < h.size(); ++i) {
    bool peak = true;
    for (size_t j = i; j-- > 0 && h[j].tid == h[i].tid && h[i].end - h[j].end <= 40;) if (h[j].score > h[i].score) { peak = false; break; }
    for (size_t j = i + 1; peak && j < h.size() && h[j].tid == h[i].tid && h[j].end - h[i].end <= 40; ++j) if (h[j].score >= h[i].score) peak = false;
    if (peak) o.push_back(h[i]);
  }
  h.swap(o);
}
}  // namespace

extern "C" {
// kind[f]: 0 unmapped, 1 paired, 2 orphan(s) only.  lab_off[n+1] CSR into lab_tid / lab_score (ascending tid).  Returns 0, or -1 if cap is too small.
int exh_labels(uint32_t n_tx, const char* const* tx_seq, const uint32_t* tx_len, uint32_t n_pairs, const uint8_t* reads, const uint64_t* off,
               int ma, int mp, int go, int ge, double min_score_frac, uint32_t fld_max, int allow_orphans, int allow_dovetail, double score_exp,
               double min_aln_prob, uint32_t threads, uint64_t* lab_off, uint32_t* lab_tid, int32_t* lab_score, uint64_t cap, uint8_t* kind) {
  Text T; T.tstart.resize(n_tx);
  T.code.push_back(5);
  for (uint32_t t = 0; t < n_tx; ++t) { T.tstart[t] = T.code.size(); for (uint32_t i = 0; i < tx_len[t]; ++i) { uint8_t c = enc((uint8_t)tx_seq[t][i]); T.code.push_back(c <= 3 ? c : 6); } T.code.push_back(5); }
  const Scoring sc{ma, mp, go, ge};
  struct Res { std::vector<std::pair<uint32_t, int32_t>> lab; uint8_t kind = 0; };
  std::vector<Res> res(n_pairs);
  std::atomic<uint32_t> next(0); std::vector<std::thread> th;
  auto work = [&]() {
    std::vector<uint8_t> q; std::vector<Hit> hits[2][2];   // [end][strand]: strand 0 = the read as given matches the transcript
    for (;;) {
      const uint32_t f = next.fetch_add(1); if (f >= n_pairs) break;
      int L[2];
      for (int e = 0; e < 2; ++e) {
        const uint8_t* s = reads + off[2 * f + e]; L[e] = (int)(off[2 * f + e + 1] - off[2 * f + e]);
        const int min_score = (int)(min_score_frac * (double)ma * (double)L[e]);
        for (int st = 0; st < 2; ++st) {
          q.resize(L[e]);
          if (st == 0) for (int i = 0; i < L[e]; ++i) q[i] = enc(s[i]);
          else for (int i = 0; i < L[e]; ++i) { const uint8_t c = enc(s[L[e] - 1 - i]); q[i] = c <= 3 ? (uint8_t)(3 - c) : 4; }
          hits[e][st].clear(); scan(T, q, sc, min_score, hits[e][st]); thin(hits[e][st]);
          for (Hit& h : hits[e][st]) find_start(T, q, sc, h);
        }
      }
      // pairs: same transcript, opposite strands, forward mate first (no dovetail), fragment length in (0, fldMax]
      std::vector<std::pair<uint32_t, int32_t>> cand;   // (tid, score)
      for (int e = 0; e < 2; ++e) {   // e = the end that is forward on the transcript
        const auto& fw = hits[e][0]; const auto& rc = hits[1 - e][1];
        size_t j0 = 0;
        for (const Hit& a : fw) {
          while (j0 < rc.size() && rc[j0].tid < a.tid) ++j0;
          for (size_t j = j0; j < rc.size() && rc[j].tid == a.tid; ++j) {
            const Hit& b = rc[j];
            if (!allow_dovetail && b.start < a.start) continue;
            const int64_t fl = (int64_t)b.start + L[1 - e] - a.start;
            if (fl <= 0 || fl > (int64_t)fld_max) continue;
            cand.emplace_back(a.tid, a.score + b.score);
          }
        }
      }
      uint8_t kd = cand.empty() ? 0 : 1;
      if (cand.empty() && allow_orphans) { for (int e = 0; e < 2; ++e) for (int st = 0; st < 2; ++st) for (const Hit& a : hits[e][st]) cand.emplace_back(a.tid, a.score); if (!cand.empty()) kd = 2; }
      std::sort(cand.begin(), cand.end(), [](const std::pair<uint32_t, int32_t>& x, const std::pair<uint32_t, int32_t>& y) { return x.first != y.first ? x.first < y.first : x.second > y.second; });
      std::vector<std::pair<uint32_t, int32_t>> bestp; int32_t best = -1000000;
      for (auto& c : cand) { if (bestp.empty() || bestp.back().first != c.first) bestp.push_back(c); best = std::max(best, c.second); }
      for (auto& c : bestp) if (std::exp(-score_exp * (double)(best - c.second)) >= min_aln_prob) res[f].lab.push_back(c);
      res[f].kind = res[f].lab.empty() ? 0 : kd;
    }
  };
  const unsigned nt = std::max(1u, threads);
  for (unsigned t = 0; t < nt; ++t) th.emplace_back(work);
  for (auto& x : th) x.join();
  uint64_t w = 0; lab_off[0] = 0;
  for (uint32_t f = 0; f < n_pairs; ++f) {
    for (auto& c : res[f].lab) { if (w >= cap) return -1; lab_tid[w] = c.first; lab_score[w] = c.second; ++w; }
    lab_off[f + 1] = w; if (kind) kind[f] = res[f].kind;
  }
  return 0;
}
}
