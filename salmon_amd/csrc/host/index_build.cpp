// host/index_build.cpp — `salmon index` for the MI355X path: transcript FASTA -> reference-coloured
// compacted de Bruijn graph (unitigs broken at reference ends, as pufferfish requires) -> contig
// table -> SSHash-style minimizer dictionary with a partitioned pilot MPHF (see sq_internal.h).
//
// Replaces: salmonIndex() -> SalmonIndex::build -> pufferfishIndex(IndexOptions&)
// (reference src/index/BuildSalmonIndex.cpp:49-262; include/salmon/internal/index/SalmonIndex.hpp:
// 106-118).  The pufferfish/TwoPaCo build itself is external to the reference tree; this is a new
// sort-free design: one concurrent k-mer hash table holds, per canonical k-mer, its 4+4 edge masks
// and reference-terminal flags; unitig boundaries are then *local* predicates evaluated while
// walking each reference, so unitigs are found without ever walking the graph.
#include "index.h"
#include "sha2.h"
#include <chrono>
#include <zlib.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <unordered_set>
#include <sys/stat.h>

static thread_local char g_err[1024] = "";
void sq_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* sq_last_error(void) { return g_err; }
extern "C" const char* sq_version(void) { return "salmon-hip 0.1 (salmon 1.11.4 hot-path semantics)"; }

// [r6] where the builder's k-mer table is made: -2 = a GPU when there is one and the input is large enough to gain (default), -1 = the host, >= 0 = that device (an error if it cannot)
static std::atomic<int> g_build_device{-2};
extern "C" int sq_index_build_set_device(int device) { if (device < -2) { sq_set_error("sq_index_build_set_device: -2 (automatic), -1 (host) or a device number"); return SQ_ERR_ARG; } g_build_device.store(device); return SQ_OK; }
int sq_index_breaks_dev(int device, const uint64_t* refseq, uint64_t refseq_words, const uint32_t* ref_len, const uint64_t* ref_accum, uint32_t nrefs, uint32_t k,
                        uint64_t total_nt, uint64_t* brkR, uint64_t* brkL);   // hip/index_build_dev.hip

namespace {

inline int base_code(char c) {
  switch (c) {
    case 'A': case 'a': return 0; case 'C': case 'c': return 1;
    case 'G': case 'g': return 2; case 'T': case 't': case 'U': case 'u': return 3;
    default: return -1;
  }
}

// Concurrent open-addressing table over canonical k-mers -> what was seen next to them.  [r2] It holds ONE PARTITION of the k-mers at a
// time (build_core walks the references once per partition), so its size is a parameter and not the genome: 10 bytes per slot.
struct KTable {
  uint64_t cap = 0;
  std::vector<uint64_t> keys;
  std::vector<uint16_t> info;  // bits0-3 R edge mask, 4-7 L edge mask, 8 Rterm, 9 Lterm
  void init(uint64_t need) {
    cap = need < 1024 ? 1024 : need;
    keys.assign(cap, ~0ULL); info.assign(cap, 0);
  }
  inline uint64_t home(uint64_t c) const { return (uint64_t)(((unsigned __int128)sq_mix64(c) * (unsigned __int128)cap) >> 64); }
  inline uint64_t insert(uint64_t c) {
    uint64_t h = home(c);
    for (uint64_t probes = 0; probes < cap; ++probes) {
      uint64_t cur = __atomic_load_n(&keys[h], __ATOMIC_RELAXED);
      if (cur == c) return h;
      if (cur == ~0ULL) {
        uint64_t exp = ~0ULL;
        if (__atomic_compare_exchange_n(&keys[h], &exp, c, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return h;
        if (exp == c) return h;
      }
      if (++h == cap) h = 0;
    }
    return ~0ULL;   // every slot taken by other k-mers: the partition got more distinct k-mers than it was sized for (the caller retries larger)
  }
  inline uint64_t find(uint64_t c) const {
    uint64_t h = home(c);
    for (;;) {
      uint64_t cur = keys[h];
      if (cur == c) return h;
      if (cur == ~0ULL) return ~0ULL;
      if (++h == cap) h = 0;
    }
  }
};
// the unitigs: key k-mer (the smaller of a unitig's two end k-mers) -> first occurrence (atomic min), then the unitig id
struct UMap {
  uint64_t cap = 0; std::vector<uint64_t> keys; std::vector<uint32_t> aux;
  void init(uint64_t need) { cap = need < 1024 ? 1024 : need; keys.assign(cap, ~0ULL); aux.assign(cap, 0xFFFFFFFFu); }
  inline uint64_t home(uint64_t c) const { return (uint64_t)(((unsigned __int128)sq_mix64(c ^ 0xA24BAED4963EE407ULL) * (unsigned __int128)cap) >> 64); }
  inline uint64_t insert(uint64_t c) {
    uint64_t h = home(c);
    for (;;) {
      uint64_t cur = __atomic_load_n(&keys[h], __ATOMIC_RELAXED);
      if (cur == c) return h;
      if (cur == ~0ULL) { uint64_t exp = ~0ULL; if (__atomic_compare_exchange_n(&keys[h], &exp, c, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return h; if (exp == c) return h; }
      if (++h == cap) h = 0;
    }
  }
  inline uint64_t find(uint64_t c) const { uint64_t h = home(c); for (;;) { uint64_t cur = keys[h]; if (cur == c) return h; if (cur == ~0ULL) return ~0ULL; if (++h == cap) h = 0; } }
};

struct Seg { uint32_t ref, pos, nk, key_lo, key_hi; };  // one unitig occurrence; key = the smaller of its two end k-mers (canonical)

inline void pool_or_bases(uint64_t* pool, uint64_t dst, const uint64_t* src, uint64_t sp, uint64_t n) {
  // copy n bases from src@sp to pool@dst using atomic OR (pool pre-zeroed; ranges may share words)
  while (n) {
    uint32_t take = (uint32_t)std::min<uint64_t>(n, 32 - (dst & 31));
    uint64_t v = sq_fetch_bases(src, sp, take);
    __atomic_fetch_or(&pool[dst >> 5], v << ((dst & 31) * 2), __ATOMIC_RELAXED);
    dst += take; sp += take; n -= take;
  }
}

struct MiniEnt { uint64_t v; uint64_t e; uint32_t kstart; uint32_t nk; };  // minimizer value, unitig<<30|off

}  // namespace

// ------------------------------------------------------------------------------------------------
static int build_core(const sq_index_opts* o, std::vector<std::string>& names, std::vector<std::string>& seqs,
                      std::vector<uint32_t>& clen, uint32_t first_decoy, sq_index* idx) {
  const uint32_t k = idx->k, m = idx->m;
  const uint32_t nthreads = std::max(1u, o && o->threads ? o->threads : std::thread::hardware_concurrency());
  const bool timing = getenv("SQ_TIMING") != nullptr; auto tph = std::chrono::steady_clock::now();
  auto phase = [&](const char* what) { if (!timing) return; const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[sq-timing] index %s %.1f s\n", what, std::chrono::duration<double>(t1 - tph).count()); tph = t1; };
  const uint32_t nrefs = (uint32_t)names.size();
  idx->names = names; idx->first_decoy = first_decoy;
  idx->ref_len.resize(nrefs); idx->ref_clen = clen; idx->ref_accum.assign(nrefs + 1, 0);
  for (uint32_t r = 0; r < nrefs; ++r) {
    idx->ref_len[r] = (uint32_t)seqs[r].size();
    idx->ref_accum[r + 1] = idx->ref_accum[r] + seqs[r].size();
  }
  const uint64_t total_nt = idx->ref_accum[nrefs];
  // ---- pack references (non-ACGT -> deterministic pseudo-random base; pufferfish fixFasta does
  // the same with an RNG) ----
  idx->refseq.assign((total_nt + 31) / 32 + 2, 0);
  {
    // [r6] in pieces of 4 M bases (a decoy chromosome is one reference of 10^8 nt: by reference, one thread packed a genome), a 64-bit word at a time: only a piece's
    // first and last word can be shared with a neighbour, and only those are OR-ed atomically (21 s -> ~1 s for 3.5 Gnt on the GPU box's 16 cores)
    struct PPiece { uint32_t r; uint32_t i0, i1; };
    std::vector<PPiece> pp;
    for (uint32_t r = 0; r < nrefs; ++r) { const uint32_t L = (uint32_t)seqs[r].size(); for (uint32_t i0 = 0; i0 < L; i0 += (4u << 20)) pp.push_back({r, i0, std::min(L, i0 + (4u << 20))}); }
    sq_parallel_for(pp.size(), nthreads, 16, [&](uint64_t b, uint64_t e, uint32_t) {
      for (uint64_t pi = b; pi < e; ++pi) {
        const PPiece pc = pp[pi]; const uint64_t r = pc.r; const std::string& s = seqs[r]; const uint64_t g = idx->ref_accum[r];
        const uint64_t w_first = (g + pc.i0) >> 5, w_last = (g + pc.i1 - 1) >> 5;
        uint64_t cur_w = w_first, acc = 0;
        auto flush = [&](uint64_t wi, uint64_t v) { if (wi == w_first || wi == w_last) __atomic_fetch_or(&idx->refseq[wi], v, __ATOMIC_RELAXED); else idx->refseq[wi] = v; };
        for (uint32_t i = pc.i0; i < pc.i1; ++i) {
          int c = base_code(s[i]);
          if (c < 0) c = (int)(sq_mix64((r << 32) ^ (uint64_t)i ^ 0x5bd1e995ULL) & 3);
          const uint64_t gp = g + i, wi = gp >> 5;
          if (wi != cur_w) { flush(cur_w, acc); cur_w = wi; acc = 0; }
          acc |= (uint64_t)c << ((gp & 31) * 2);
        }
        flush(cur_w, acc);
      }
    });
  }
  std::vector<std::string>().swap(seqs);  // free ASCII
  phase("pack references");
  const uint64_t* rs = idx->refseq.data();
  // ---- k-mer table: edges + terminal flags -> two bits per reference position ----
  // [r2] What the unitig walk needs of a k-mer is local: does the unitig end on its right / on its left (a terminal, or not exactly one
  // neighbour).  So the table is built one PARTITION of the k-mers at a time (partition = hash of the canonical k-mer): a pass inserts the
  // k-mers of its partition with what was seen next to them, then writes, for every reference position whose k-mer it owns, the two
  // verdicts into bit arrays indexed by position.  The table's size is a budget (SQ_INDEX_TABLE_GB, default 12 GB), not the genome: a
  // transcriptome takes one pass, a 3-Gnt decoy genome five; the bit arrays are 2 bits per nucleotide.
  uint64_t npos = 0;
  for (uint32_t r = 0; r < nrefs; ++r) if (idx->ref_len[r] >= k) npos += idx->ref_len[r] - k + 1;
  const uint64_t km = sq_kmask(k);
  const double budget_gb = getenv("SQ_INDEX_TABLE_GB") ? std::max(0.001, atof(getenv("SQ_INDEX_TABLE_GB"))) : 12.0;
  const uint64_t slot_budget = (uint64_t)(budget_gb * 1e9 / 10.0);                        // 10 bytes per slot
  const uint32_t nparts_k = (uint32_t)std::max<uint64_t>(1, ((uint64_t)(npos * 1.35) + slot_budget - 1) / slot_budget);
  auto kmer_part = [&](uint64_t c) -> uint32_t { return nparts_k == 1 ? 0u : (uint32_t)((sq_mix64(c ^ 0x51ED270B1A2C3D4FULL) >> 17) % nparts_k); };
  // positions in pieces of at most 4 M k-mers, so that a few chromosomes still fill every thread
  struct Piece { uint32_t ref, i0, i1; };
  std::vector<Piece> pieces;
  for (uint32_t r = 0; r < nrefs; ++r) { const uint32_t L = idx->ref_len[r]; if (L < k) continue; const uint32_t nk = L - k + 1;
    for (uint32_t i0 = 0; i0 < nk; i0 += (4u << 20)) pieces.push_back({r, i0, std::min(nk, i0 + (4u << 20))}); }
  std::vector<uint64_t> brkR((total_nt + 63) / 64 + 1, 0), brkL((total_nt + 63) / 64 + 1, 0);   // by global position of the k-mer's first base
  auto side_break = [](uint32_t inf, bool right) -> bool {
    uint32_t msk = right ? (inf & 15u) : ((inf >> 4) & 15u);
    bool term = right ? ((inf >> 8) & 1u) : ((inf >> 9) & 1u);
    return term || __builtin_popcount(msk) != 1;
  };
  // [r6] with a GPU at hand the table is built there (hip/index_build_dev.hip: one partition in HBM, two launches); the host's passes below are what runs without one
  bool on_device = false;
  {
    const int choice = g_build_device.load();
    const uint64_t min_pos = getenv("SQ_INDEX_DEVICE_MIN_POS") ? (uint64_t)atoll(getenv("SQ_INDEX_DEVICE_MIN_POS")) : 20000000ull;
    if (choice >= 0 || (choice == -2 && npos >= min_pos)) {
      const int rc = sq_index_breaks_dev(choice >= 0 ? choice : 0, rs, idx->refseq.size(), idx->ref_len.data(), idx->ref_accum.data(), nrefs, k, total_nt, brkR.data(), brkL.data());
      if (rc == SQ_OK) on_device = true;
      else if (choice >= 0) throw std::runtime_error(std::string("index builder on device ") + std::to_string(choice) + ": " + sq_last_error());
      else {   // automatic choice: without a device the host builds (silently); anything else is said
        if (rc != SQ_ERR_DEVICE) fprintf(stderr, "[salmon-hip] index: the k-mer table could not be built on the device (%s): building it on the host\n", sq_last_error());
        std::fill(brkR.begin(), brkR.end(), 0); std::fill(brkL.begin(), brkL.end(), 0);
      }
    }
  }
  if (!on_device) {
    KTable T;
    double grow = 1.0;
    for (uint32_t part = 0; part < nparts_k; ++part) {
      // a partition holds ~1/nparts of the DISTINCT k-mers; sized for 1/nparts of the positions (+ 8 % for the spread of the hash);
      // a partition that overflows all the same (a skewed hash under a tiny SQ_INDEX_TABLE_GB) is redone with a larger table
      T.init((uint64_t)((double)npos / nparts_k * (nparts_k > 1 ? 1.08 : 1.0) * 1.35 * grow) + 1024);
      std::atomic<int> full(0);
      sq_parallel_for(pieces.size(), nthreads, 1, [&](uint64_t b, uint64_t e, uint32_t) {
        for (uint64_t pi = b; pi < e; ++pi) {
          const Piece pc = pieces[pi]; const uint32_t L = idx->ref_len[pc.ref], nk = L - k + 1; const uint64_t g = idx->ref_accum[pc.ref];
          uint64_t fw = sq_fetch_bases(rs, g + pc.i0, k), rc = sq_revcomp(fw, k);
          for (uint32_t i = pc.i0; i < pc.i1; ++i) {
            if (i > pc.i0) { uint64_t nb = sq_fetch_base(rs, g + i + k - 1); fw = (fw >> 2) | (nb << (2 * (k - 1))); rc = ((rc << 2) | (3 - nb)) & km; }
            bool o1 = fw < rc; uint64_t c = o1 ? fw : rc;
            if (kmer_part(c) != part) continue;
            uint64_t slot = T.insert(c);
            if (slot == ~0ULL) { full.store(1, std::memory_order_relaxed); return; }
            uint32_t bits = 0;
            if (i + 1 < nk) { uint32_t sb = sq_fetch_base(rs, g + i + k); bits |= o1 ? (1u << sb) : (1u << (4 + (3 - sb))); }
            else bits |= o1 ? (1u << 8) : (1u << 9);
            if (i > 0) { uint32_t pb = sq_fetch_base(rs, g + i - 1); bits |= o1 ? (1u << (4 + pb)) : (1u << (3 - pb)); }
            else bits |= o1 ? (1u << 9) : (1u << 8);
            __atomic_fetch_or(&T.info[slot], (uint16_t)bits, __ATOMIC_RELAXED);
          }
        }
      });
      if (full.load()) {
        if (grow > 64.0) throw std::runtime_error("k-mer table partition keeps overflowing (SQ_INDEX_TABLE_GB too small for this input?)");
        grow *= 1.5; --part; continue;
      }
      sq_parallel_for(pieces.size(), nthreads, 1, [&](uint64_t b, uint64_t e, uint32_t) {
        for (uint64_t pi = b; pi < e; ++pi) {
          const Piece pc = pieces[pi]; const uint64_t g = idx->ref_accum[pc.ref];
          uint64_t fw = sq_fetch_bases(rs, g + pc.i0, k), rc = sq_revcomp(fw, k);
          for (uint32_t i = pc.i0; i < pc.i1; ++i) {
            if (i > pc.i0) { uint64_t nb = sq_fetch_base(rs, g + i + k - 1); fw = (fw >> 2) | (nb << (2 * (k - 1))); rc = ((rc << 2) | (3 - nb)) & km; }
            bool o1 = fw < rc; uint64_t c = o1 ? fw : rc;
            if (kmer_part(c) != part) continue;
            const uint32_t inf = T.info[T.find(c)]; const uint64_t gp = g + i;
            if (side_break(inf, o1)) __atomic_fetch_or(&brkR[gp >> 6], 1ULL << (gp & 63), __ATOMIC_RELAXED);     // the unitig ends after this occurrence
            if (side_break(inf, !o1)) __atomic_fetch_or(&brkL[gp >> 6], 1ULL << (gp & 63), __ATOMIC_RELAXED);    // ... before it
          }
        }
      });
    }
  }
  phase(on_device ? "k-mer table (device)" : "k-mer table passes");
  // ---- walk references: local boundary predicate -> segments (unitig occurrences) ----
  std::vector<std::vector<Seg>> rsegs(nrefs);
  sq_parallel_for(nrefs, nthreads, 16, [&](uint64_t b, uint64_t e, uint32_t) {
    for (uint64_t r = b; r < e; ++r) {
      uint32_t L = idx->ref_len[r]; if (L < k) continue;
      uint64_t g = idx->ref_accum[r];
      uint64_t fw = sq_fetch_bases(rs, g, k), rc = sq_revcomp(fw, k);
      uint32_t nk = L - k + 1;
      bool o1 = fw < rc; uint64_t c = o1 ? fw : rc;
      uint32_t seg_start = 0; uint64_t first_c = c;
      auto& out = rsegs[r];
      for (uint32_t i = 0; i < nk; ++i) {
        bool last = (i + 1 == nk);
        bool brk = true; uint64_t nc = 0; bool no1 = false;
        if (!last) {
          uint64_t nb = sq_fetch_base(rs, g + i + k);
          fw = (fw >> 2) | (nb << (2 * (k - 1))); rc = ((rc << 2) | (3 - nb)) & km;
          no1 = fw < rc; nc = no1 ? fw : rc;
          const uint64_t gp = g + i, gn = gp + 1;
          brk = ((brkR[gp >> 6] >> (gp & 63)) & 1) || ((brkL[gn >> 6] >> (gn & 63)) & 1) || (nc == c);
        }
        if (brk) {
          const uint64_t key = (c < first_c) ? c : first_c;  // min(c_first, c_last): the unitig's identity
          Seg s; s.ref = (uint32_t)r; s.pos = seg_start; s.nk = i - seg_start + 1;
          s.key_lo = (uint32_t)key; s.key_hi = (uint32_t)(key >> 32);
          out.push_back(s);
          seg_start = i + 1; first_c = nc;
        }
        c = nc; o1 = no1;
      }
    }
  });
  std::vector<uint64_t>().swap(brkR); std::vector<uint64_t>().swap(brkL);
  phase("walk");
  std::vector<uint64_t> seg_off(nrefs + 1, 0);
  for (uint32_t r = 0; r < nrefs; ++r) seg_off[r + 1] = seg_off[r] + rsegs[r].size();
  const uint64_t S = seg_off[nrefs];
  if (S >= 0xFFFFFFFFull) {
    sq_set_error("too many unitig occurrences (%llu) for this index format", (unsigned long long)S);
    return SQ_ERR_OVERFLOW;
  }
  std::vector<Seg> segs(S);
  sq_parallel_for(nrefs, nthreads, 64, [&](uint64_t b, uint64_t e, uint32_t) {
    for (uint64_t r = b; r < e; ++r) {
      std::copy(rsegs[r].begin(), rsegs[r].end(), segs.begin() + seg_off[r]);
      std::vector<Seg>().swap(rsegs[r]);
    }
  });
  phase("  segments gathered");
  auto skey = [](const Seg& s) { return (uint64_t)s.key_lo | ((uint64_t)s.key_hi << 32); };
  // first occurrence (in reference order) of every unitig defines its id, orientation and sequence
  UMap T; T.init((uint64_t)(S * 1.5) + 1024);
  std::vector<uint64_t> seg_slot(S);
  sq_parallel_for(S, nthreads, 1 << 16, [&](uint64_t b, uint64_t e, uint32_t) {
    for (uint64_t i = b; i < e; ++i) {
      const uint64_t sl = T.insert(skey(segs[i])); seg_slot[i] = sl;
      uint32_t* a = &T.aux[sl]; uint32_t v = (uint32_t)i;
      uint32_t cur = __atomic_load_n(a, __ATOMIC_RELAXED);
      while (v < cur && !__atomic_compare_exchange_n(a, &cur, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    }
  });
  phase("  unitig keys hashed");
  auto kslot = [&](const Seg& s) { return seg_slot[(uint64_t)(&s - segs.data())]; };
  std::vector<uint8_t> defining(S);
  std::vector<uint64_t> def_idx;
  for (uint64_t i = 0; i < S; ++i) { defining[i] = (T.aux[kslot(segs[i])] == (uint32_t)i); if (defining[i]) def_idx.push_back(i); }
  const uint64_t U = def_idx.size();
  if (U >= (1ULL << 30)) { sq_set_error("too many unitigs (%llu)", (unsigned long long)U); return SQ_ERR_OVERFLOW; }
  idx->uoff.assign(U + 1, 0);
  for (uint64_t u = 0; u < U; ++u) {
    const Seg& s = segs[def_idx[u]];
    uint64_t ulen = (uint64_t)s.nk + k - 1;
    if (ulen >= (1ULL << SQ_UOFF_BITS)) { sq_set_error("unitig too long (%llu nt)", (unsigned long long)ulen); return SQ_ERR_OVERFLOW; }
    idx->uoff[u + 1] = idx->uoff[u] + ulen;
    T.aux[kslot(s)] = (uint32_t)u;  // now: unitig id
  }
  phase("  unitig ids");
  const uint64_t pool_nt = idx->uoff[U];
  if (pool_nt >= (1ULL << SQ_APOS_BITS)) {
    sq_set_error("unitig pool too large (%llu nt)", (unsigned long long)pool_nt);
    return SQ_ERR_OVERFLOW;
  }
  idx->useq.assign((pool_nt + 31) / 32 + 2, 0);
  sq_parallel_for(U, nthreads, 4096, [&](uint64_t b, uint64_t e, uint32_t) {
    for (uint64_t u = b; u < e; ++u) {
      const Seg& s = segs[def_idx[u]];
      pool_or_bases(idx->useq.data(), idx->uoff[u], rs, idx->ref_accum[s.ref] + s.pos, (uint64_t)s.nk + k - 1);
    }
  });
  phase("unitigs");
  // ---- contig table (stable counting sort by unitig id keeps (tid,pos) order) ----
  std::vector<uint32_t> seg_uid(S); std::vector<uint8_t> seg_fw(S);
  sq_parallel_for(S, nthreads, 1 << 14, [&](uint64_t b, uint64_t e, uint32_t) {
    for (uint64_t i = b; i < e; ++i) {
      const Seg& s = segs[i]; uint32_t u = T.aux[kslot(s)]; seg_uid[i] = u;
      uint64_t a = sq_fetch_bases(rs, idx->ref_accum[s.ref] + s.pos, k);
      uint64_t ub = sq_fetch_bases(idx->useq.data(), idx->uoff[u], k);
      seg_fw[i] = (a == ub);
      if (a == ub && (uint64_t)s.nk + k - 1 == idx->uoff[u + 1] - idx->uoff[u]) {
        // possible ambiguity only if the unitig is its own reverse complement (excluded by the
        // hairpin rule); nothing to do
      }
    }
  });
  phase("  orientations");
  idx->ctab_off.assign(U + 1, 0);
  for (uint64_t i = 0; i < S; ++i) idx->ctab_off[seg_uid[i] + 1]++;
  for (uint64_t u = 0; u < U; ++u) idx->ctab_off[u + 1] += idx->ctab_off[u];
  idx->ctab.assign(S, 0);
  {
    std::vector<uint64_t> cur(idx->ctab_off.begin(), idx->ctab_off.end() - 1);
    for (uint64_t i = 0; i < S; ++i) {
      const Seg& s = segs[i];
      idx->ctab[cur[seg_uid[i]]++] = ((uint64_t)s.ref << 32) | ((uint64_t)seg_fw[i] << 31) | (uint64_t)s.pos;
    }
  }
  uint64_t nk_total = 0;
  for (uint64_t u = 0; u < U; ++u) nk_total += idx->uoff[u + 1] - idx->uoff[u] - (k - 1);
  idx->num_kmers = nk_total;
  // free the table before the dictionary build
  std::vector<Seg>().swap(segs);
  std::vector<uint64_t>().swap(T.keys);
  std::vector<uint32_t>().swap(T.aux);
  std::vector<uint64_t>().swap(seg_slot);

  phase("contig table");
  // ---- minimizers / super-k-mers ----
  const uint64_t* up = idx->useq.data();
  const uint32_t w = k - m; const uint64_t mm = sq_kmask(m);
  std::vector<std::vector<MiniEnt>> tent(nthreads);
  // [r2] k-mer positions in pieces of at most 4 M per task: a decoy chromosome can be ONE unitig of 10^8 nt, and a thread per unitig
  // left the others idle for minutes.  A super-k-mer cut by a piece boundary comes out as two entries with the same minimizer
  // occurrence; they are merged after the sort below (an occurrence is the minimizer of one contiguous run of k-mers).
  struct UPiece { uint64_t u; uint32_t p0, p1; };
  std::vector<UPiece> upieces;
  for (uint64_t u = 0; u < U; ++u) { const uint32_t nk = (uint32_t)(idx->uoff[u + 1] - idx->uoff[u]) - k + 1;
    for (uint32_t p0 = 0; p0 < nk; p0 += (4u << 20)) upieces.push_back({u, p0, std::min(nk, p0 + (4u << 20))}); }
  sq_parallel_for(upieces.size(), nthreads, 64, [&](uint64_t b, uint64_t e, uint32_t t) {
    std::vector<uint64_t> hv, cv;
    auto& out = tent[t];
    for (uint64_t pi = b; pi < e; ++pi) {
      const uint64_t u = upieces[pi].u; const uint32_t p0 = upieces[pi].p0, p1 = upieces[pi].p1;
      const uint64_t ub = idx->uoff[u];
      const uint32_t nm = (p1 - p0) + w;               // m-mers p0 .. p1 - 1 + w
      hv.resize(nm); cv.resize(nm);
      uint64_t f = sq_fetch_bases(up, ub + p0, m), r = sq_revcomp(f, m);
      for (uint32_t i = 0; i < nm; ++i) {
        if (i) { uint64_t nb = sq_fetch_base(up, ub + p0 + i + m - 1); f = (f >> 2) | (nb << (2 * (m - 1))); r = ((r << 2) | (3 - nb)) & mm; }
        uint64_t c = f < r ? f : r; cv[i] = c; hv[i] = sq_mhash(c);
      }
      uint32_t prev = 0xFFFFFFFFu;
      for (uint32_t p = p0; p < p1; ++p) {
        const uint32_t q = p - p0; uint32_t bj = q; uint64_t bh = hv[q];
        // leftmost minimum of (sq_mhash, value)
        for (uint32_t j = q + 1; j <= q + w; ++j) if (hv[j] < bh || (hv[j] == bh && cv[j] < cv[bj])) {
          bh = hv[j];
          bj = j;
        }
        if (bj != prev) { out.push_back({cv[bj], (u << SQ_APOS_BITS) | (ub + p0 + bj), p, 1}); prev = bj; }
        else out.back().nk++;
      }
    }
  });
  std::vector<MiniEnt> ents;
  {
    uint64_t n = 0;
    for (auto& v : tent) n += v.size();
    ents.reserve(n);
    for (auto& v : tent) {
      ents.insert(ents.end(), v.begin(), v.end());
      std::vector<MiniEnt>().swap(v);
    }
  }
  idx->num_superkmers = ents.size();
  phase("minimizers");
  // parallel sort by (v, e): bucket by top bits of mix(v) is unnecessary; a plain parallel merge is enough
  {
    uint64_t n = ents.size(); uint32_t P = nthreads; std::vector<uint64_t> cut(P + 1);
    for (uint32_t i = 0; i <= P; ++i) cut[i] = n * i / P;
    auto cmp = [](const MiniEnt& a, const MiniEnt& b) { return a.v < b.v || (a.v == b.v && a.e < b.e); };
    sq_parallel_for(P, P, 1, [&](uint64_t b, uint64_t e, uint32_t) { for (uint64_t i = b; i < e; ++i) std::sort(ents.begin() + cut[i],
        ents.begin() + cut[i + 1],
        cmp); });
    for (uint32_t step = 1; step < P; step <<= 1) {
      std::vector<std::thread> th;
      for (uint32_t i = 0; i + step < P; i += 2 * step) {
        uint64_t a = cut[i], mid = cut[i + step], en = cut[std::min(P, i + 2 * step)];
        th.emplace_back([&, a, mid, en]() { std::inplace_merge(ents.begin() + a, ents.begin() + mid, ents.begin() + en, cmp); });
      }
      for (auto& x : th) x.join();
    }
  }
  // the halves of super-k-mers cut by a piece boundary: same (minimizer, occurrence), adjacent k-mer runs
  { uint64_t wq = 0;
    for (uint64_t i = 0; i < ents.size(); ++i) {
      if (wq && ents[wq - 1].v == ents[i].v && ents[wq - 1].e == ents[i].e) { ents[wq - 1].kstart = std::min(ents[wq - 1].kstart, ents[i].kstart); ents[wq - 1].nk += ents[i].nk; }
      else ents[wq++] = ents[i];
    }
    ents.resize(wq); idx->num_superkmers = wq; }
  phase("sort super-k-mers");
  // distinct minimizers
  std::vector<uint64_t> keys, kstart;  // key value, first index into ents
  for (uint64_t i = 0; i < ents.size(); ++i) if (i == 0 || ents[i].v != ents[i - 1].v) { keys.push_back(ents[i].v); kstart.push_back(i); }
  kstart.push_back(ents.size());
  const uint64_t NK = keys.size();
  idx->num_minimizers = NK;
  phase("distinct minimizers");
  // ---- partitioned pilot MPHF ----
  uint32_t nparts = (uint32_t)std::max<uint64_t>(1, (NK + SQ_MPHF_PART_KEYS - 1) / SQ_MPHF_PART_KEYS);
  idx->n_parts = nparts;
  std::vector<uint64_t> kh(NK); std::vector<uint32_t> kpart(NK);
  std::vector<uint64_t> pcount(nparts + 1, 0);
  for (uint64_t i = 0; i < NK; ++i) {
    kh[i] = sq_mix64(keys[i] ^ 0x9E3779B97F4A7C15ULL);
    kpart[i] = sq_fastrange32((uint32_t)(kh[i] >> 32), nparts);
    pcount[kpart[i] + 1]++;
  }
  for (uint32_t p = 0; p < nparts; ++p) pcount[p + 1] += pcount[p];
  std::vector<uint64_t> pkeys(NK);  // key indices grouped by partition
  { std::vector<uint64_t> cur(pcount.begin(), pcount.end() - 1); for (uint64_t i = 0; i < NK; ++i) pkeys[cur[kpart[i]]++] = i; }
  idx->part_slot_off.assign(nparts + 1, 0); idx->part_bkt_off.assign(nparts + 1, 0);
  std::vector<uint32_t> pns(nparts), pnb(nparts);
  for (uint32_t p = 0; p < nparts; ++p) {
    uint64_t np = pcount[p + 1] - pcount[p];
    pns[p] = (uint32_t)std::max<uint64_t>(1, (uint64_t)(np / SQ_MPHF_ALPHA) + 1);
    pnb[p] = (uint32_t)std::max<uint64_t>(1, (uint64_t)(np / SQ_MPHF_LAMBDA) + 1);
    idx->part_slot_off[p + 1] = idx->part_slot_off[p] + pns[p];
    idx->part_bkt_off[p + 1] = idx->part_bkt_off[p] + pnb[p];
  }
  idx->pilots.assign(idx->part_bkt_off[nparts], 0);
  std::vector<uint64_t> slot_key(idx->part_slot_off[nparts], ~0ULL);  // slot -> key index
  std::atomic<int> mphf_fail(0);
  sq_parallel_for(nparts, nthreads, 1, [&](uint64_t pb, uint64_t pe, uint32_t) {
    std::vector<std::vector<uint64_t>> bk; std::vector<uint32_t> bord; std::vector<uint8_t> taken; std::vector<uint32_t> pos;
    for (uint64_t p = pb; p < pe; ++p) {
      uint32_t ns = pns[p], nb = pnb[p]; uint64_t s0 = idx->part_slot_off[p]; uint32_t b0 = idx->part_bkt_off[p];
      bk.assign(nb, {}); taken.assign(ns, 0);
      for (uint64_t q = pcount[p]; q < pcount[p + 1]; ++q) {
        uint64_t ki = pkeys[q];
        bk[sq_fastrange32((uint32_t)kh[ki], nb)].push_back(ki);
      }
      bord.resize(nb); for (uint32_t i = 0; i < nb; ++i) bord[i] = i;
      std::stable_sort(bord.begin(), bord.end(), [&](uint32_t a, uint32_t b) { return bk[a].size() > bk[b].size(); });
      for (uint32_t bi : bord) {
        auto& B = bk[bi]; if (B.empty()) continue;
        bool ok = false;
        for (uint32_t pilot = 0; pilot < 65536 && !ok; ++pilot) {
          uint64_t pm = sq_pilot_mix(pilot);
          pos.clear(); bool good = true;
          for (uint64_t ki : B) {
            uint32_t s = sq_fastrange32(sq_slot_mix(kh[ki], pm), ns);
            if (taken[s]) { good = false; break; }
            for (uint32_t x : pos) if (x == s) { good = false; break; }
            if (!good) break;
            pos.push_back(s);
          }
          if (good) {
            for (size_t i = 0; i < B.size(); ++i) {
              taken[pos[i]] = 1;
              slot_key[s0 + pos[i]] = B[i];
            }
            idx->pilots[b0 + bi] = (uint16_t)pilot;
            ok = true;
          }
        }
        if (!ok) { mphf_fail.store(1); return; }
      }
    }
  });
  if (mphf_fail.load()) { sq_set_error("MPHF pilot search failed (increase slack)"); return SQ_ERR_STATE; }
  phase("MPHF");
  // ---- slot records, entry lists, skew table ----
  idx->slots.assign(slot_key.size(), SQ_SLOT_EMPTY);
  idx->entries.clear();
  std::vector<uint64_t> skew_k, skew_v;
  uint64_t maxb = 0;
  {
    // [r6] three steps instead of one thread's walk over every slot (33 s for the 3.5 Gnt index, most of it cache misses on `kstart` / `ents`): the buckets' sizes in
    // parallel, their places in `entries` by one running sum in slot order (the order the walk gave them), the records and entry lists in parallel again.  The skew
    // table's k-mers (buckets of more than SQ_SKEW_THRESH occurrences: few) are still collected in slot order by one thread.
    const uint64_t NS = slot_key.size(); const uint64_t CH = 1u << 16; const uint64_t nch = (NS + CH - 1) / CH;
    std::vector<uint64_t> ch_ent(nch + 1, 0), ch_max(nch, 0); std::atomic<int> too_large(0);
    sq_parallel_for(nch, nthreads, 4, [&](uint64_t b, uint64_t e, uint32_t) {
      for (uint64_t c = b; c < e; ++c) { uint64_t n = 0, mx = 0;
        for (uint64_t s = c * CH; s < std::min(NS, (c + 1) * CH); ++s) { const uint64_t ki = slot_key[s]; if (ki == ~0ULL) continue; const uint64_t cnt = kstart[ki + 1] - kstart[ki]; mx = std::max(mx, cnt);
          if (cnt >= (1ULL << 23)) too_large.store(1); if (cnt > 1) n += cnt; }
        ch_ent[c + 1] = n; ch_max[c] = mx; }
    });
    if (too_large.load()) { sq_set_error("minimizer bucket too large"); return SQ_ERR_OVERFLOW; }
    for (uint64_t c = 0; c < nch; ++c) { ch_ent[c + 1] += ch_ent[c]; maxb = std::max(maxb, ch_max[c]); }
    idx->entries.resize(ch_ent[nch]);
    sq_parallel_for(nch, nthreads, 4, [&](uint64_t b, uint64_t e, uint32_t) {
      for (uint64_t c = b; c < e; ++c) { uint64_t at = ch_ent[c];
        for (uint64_t s = c * CH; s < std::min(NS, (c + 1) * CH); ++s) { const uint64_t ki = slot_key[s]; if (ki == ~0ULL) continue; const uint64_t a = kstart[ki], bb = kstart[ki + 1], cnt = bb - a;
          if (cnt == 1) { idx->slots[s] = SQ_SLOT_INLINE | ents[a].e; continue; }
          idx->slots[s] = at | (cnt << SQ_POS_BITS);
          for (uint64_t i = a; i < bb; ++i) idx->entries[at++] = ents[i].e; } }
    });
    if (maxb > SQ_SKEW_THRESH) for (uint64_t s = 0; s < NS; ++s) {
      const uint64_t ki = slot_key[s]; if (ki == ~0ULL) continue; const uint64_t a = kstart[ki], bb = kstart[ki + 1];
      if (bb - a <= SQ_SKEW_THRESH) continue;
      for (uint64_t i = a; i < bb; ++i) {
        uint64_t u = ents[i].e >> SQ_APOS_BITS;
        for (uint32_t q = 0; q < ents[i].nk; ++q) {
          uint64_t st = ents[i].kstart + q;
          uint64_t f = sq_fetch_bases(up, idx->uoff[u] + st, k), r = sq_revcomp(f, k);
          skew_k.push_back(f < r ? f : r); skew_v.push_back((u << SQ_APOS_BITS) | (idx->uoff[u] + st));
        }
      }
    }
  }
  if (idx->entries.empty()) idx->entries.push_back(0);
  idx->max_bucket = maxb; idx->num_skew_kmers = skew_k.size();
  if (!skew_k.empty()) {
    uint64_t cap = 16; while (cap < skew_k.size() * 2) cap <<= 1;
    idx->skew_keys.assign(cap, ~0ULL); idx->skew_vals.assign(cap, 0);
    for (size_t i = 0; i < skew_k.size(); ++i) {
      uint64_t h = sq_mix64(skew_k[i]) & (cap - 1);
      while (idx->skew_keys[h] != ~0ULL) h = (h + 1) & (cap - 1);
      idx->skew_keys[h] = skew_k[i]; idx->skew_vals[h] = skew_v[i];
    }
  }
  phase("slot records");
  return SQ_OK;
}

// ------------------------------------------------------------------------------------------------
// Digests of the input records, in input order, before anything is changed (upstream pufferfish hashes every record as it is read —
// sequence bytes as given, name = the first token, cut at '|' with --gencode — the targets into SeqHash / NameHash (+ the 512-bit
// forms), the decoys into DecoySeqHash / DecoyNameHash; recalled, not verifiable here: SURVEY.md Appendix B).  Six independent
// streams, one thread each.
static void hash_refs(const std::vector<std::string>& names, const std::vector<std::string>& seqs, const std::vector<uint8_t>& is_decoy, bool gencode, std::string out[6]) {
  auto name_of = [&](size_t i) { const std::string& s = names[i]; const size_t p = gencode ? s.find('|') : std::string::npos; return p == std::string::npos ? s.size() : p; };
  std::thread th[6];
  th[0] = std::thread([&] { sqsha::Sha256 h; for (size_t i = 0; i < seqs.size(); ++i) if (!is_decoy[i]) h.update(seqs[i].data(), seqs[i].size()); out[0] = h.hex(); });
  th[1] = std::thread([&] { sqsha::Sha256 h; for (size_t i = 0; i < seqs.size(); ++i) if (!is_decoy[i]) h.update(names[i].data(), name_of(i)); out[1] = h.hex(); });
  th[2] = std::thread([&] { sqsha::Sha512 h; for (size_t i = 0; i < seqs.size(); ++i) if (!is_decoy[i]) h.update(seqs[i].data(), seqs[i].size()); out[2] = h.hex(); });
  th[3] = std::thread([&] { sqsha::Sha512 h; for (size_t i = 0; i < seqs.size(); ++i) if (!is_decoy[i]) h.update(names[i].data(), name_of(i)); out[3] = h.hex(); });
  th[4] = std::thread([&] { sqsha::Sha256 h; for (size_t i = 0; i < seqs.size(); ++i) if (is_decoy[i]) h.update(seqs[i].data(), seqs[i].size()); out[4] = h.hex(); });
  th[5] = std::thread([&] { sqsha::Sha256 h; for (size_t i = 0; i < seqs.size(); ++i) if (is_decoy[i]) h.update(names[i].data(), name_of(i)); out[5] = h.hex(); });
  for (auto& t : th) t.join();
}

static void prep_refs(const sq_index_opts* o, std::vector<std::string>& names, std::vector<std::string>& seqs,
                      std::vector<uint32_t>& clen, std::vector<uint8_t>& is_decoy, uint32_t* first_decoy,
                      std::vector<std::pair<std::string, std::string>>& dups, sq_index* idx = nullptr) {
  const bool clip = !(o && o->no_clip_polya), keepdup = (o && o->keep_duplicates), gencode = (o && o->gencode);
  size_t n = names.size();
  if (idx) { idx->keep_duplicates = keepdup; hash_refs(names, seqs, is_decoy, gencode, idx->hashes); }
  clen.resize(n);
  for (size_t i = 0; i < n; ++i) {
    if (gencode) { size_t p = names[i].find('|'); if (p != std::string::npos) names[i].resize(p); }
    std::string& s = seqs[i];
    for (auto& c : s) c = (char)toupper((unsigned char)c);
    clen[i] = (uint32_t)s.size();
    if (clip && !is_decoy[i]) {  // clip poly-A tails (BuildSalmonIndex.cpp:120-121 "--no-clip")
      size_t e = s.size(); while (e > 0 && s[e - 1] == 'A') --e;
      if (s.size() - e >= 10) s.resize(e);
    }
  }
  // duplicates (sequence-identical transcripts are discarded unless --keepDuplicates)
  std::vector<uint8_t> drop(n, 0);
  if (!keepdup) {
    std::unordered_map<uint64_t, std::vector<uint32_t>> byh;
    for (size_t i = 0; i < n; ++i) {
      if (is_decoy[i]) continue;
      uint64_t h = 1469598103934665603ULL; for (unsigned char c : seqs[i]) { h ^= c; h *= 1099511628211ULL; }
      h = sq_mix64(h ^ seqs[i].size());
      auto& v = byh[h]; bool dup = false;
      for (uint32_t j : v) if (seqs[j] == seqs[i]) { dups.emplace_back(names[j], names[i]); dup = true; break; }
      if (dup) drop[i] = 1; else v.push_back((uint32_t)i);
    }
  }
  // keep order: targets first, then decoys
  std::vector<std::string> n2, s2; std::vector<uint32_t> c2;
  for (int pass = 0; pass < 2; ++pass)
    for (size_t i = 0; i < n; ++i) if (!drop[i] && (is_decoy[i] != 0) == (pass == 1)) {
      n2.push_back(std::move(names[i]));
      s2.push_back(std::move(seqs[i]));
      c2.push_back(clen[i]);
    }
  uint32_t fd = 0; for (size_t i = 0; i < n; ++i) if (!drop[i] && !is_decoy[i]) ++fd;
  *first_decoy = fd;
  names.swap(n2); seqs.swap(s2); clen.swap(c2);
}

static int read_fasta(const char* path, std::vector<std::string>& names, std::vector<std::string>& seqs) {
  gzFile f = gzopen(path, "rb");
  if (!f) { sq_set_error("cannot open FASTA '%s'", path); return SQ_ERR_IO; }
  gzbuffer(f, 1 << 20);
  std::vector<char> buf(1 << 20); std::string line; bool have = false;
  auto flush_line = [&](const std::string& l) {
    if (l.empty()) return;
    if (l[0] == '>') {
      size_t e = l.find_first_of(" \t", 1);
      names.push_back(l.substr(1, e == std::string::npos ? std::string::npos : e - 1));
      seqs.emplace_back();
      have = true;
    }
    else if (have) { for (char c : l) if (!isspace((unsigned char)c)) seqs.back().push_back(c); }
  };
  int n;
  while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) {
    for (int i = 0; i < n; ++i) {
      char c = buf[i];
      if (c == '\n') {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        flush_line(line);
        line.clear();
      } else line.push_back(c);
    }
  }
  flush_line(line);
  gzclose(f);
  if (names.empty()) { sq_set_error("no sequences in '%s'", path); return SQ_ERR_IO; }
  return SQ_OK;
}

static int finish_opts(const sq_index_opts* o, sq_index* idx) {
  uint32_t k = (o && o->k) ? o->k : 31;
  if (k % 2 == 0) { sq_set_error("k must be an odd value, you chose %u", k); return SQ_ERR_ARG; }   // BuildSalmonIndex.cpp:204
  if (k > 31) { sq_set_error("k must not be larger than 31, you chose %u", k); return SQ_ERR_ARG; }  // :207
  uint32_t m = (o && o->m) ? o->m : std::min(20u, std::max(4u, k - 4));                              // :78-81
  if (m >= k) { sq_set_error("minimizer length must be less than k"); return SQ_ERR_ARG; }
  idx->k = k; idx->m = m;
  return SQ_OK;
}

static int index_build_mem_impl(const sq_index_opts* opts, uint32_t nrefs, const char* const* names, const char* const* seqs, const uint32_t* lens,
                                uint32_t first_decoy, const char* outdir, sq_index** out);
static int index_build_impl(const sq_index_opts* opts, const char* fasta_path, const char* decoys_path, const char* outdir, sq_index** out);
// the C ABI never lets a C++ exception out (a ctypes / cgo host would abort): allocation failures become SQ_ERR_NOMEM
extern "C" int sq_index_build_mem(const sq_index_opts* opts, uint32_t nrefs, const char* const* names, const char* const* seqs, const uint32_t* lens,
                                  uint32_t first_decoy, const char* outdir, sq_index** out) {
  try { return index_build_mem_impl(opts, nrefs, names, seqs, lens, first_decoy, outdir, out); }
  catch (const std::bad_alloc&) { sq_set_error("out of memory while building the index"); return SQ_ERR_NOMEM; }
  catch (const std::exception& e) { sq_set_error("index build failed: %s", e.what()); return SQ_ERR_STATE; }
}
extern "C" int sq_index_build(const sq_index_opts* opts, const char* fasta_path, const char* decoys_path, const char* outdir) {
  try { return index_build_impl(opts, fasta_path, decoys_path, outdir, nullptr); }
  catch (const std::bad_alloc&) { sq_set_error("out of memory while building the index"); return SQ_ERR_NOMEM; }
  catch (const std::exception& e) { sq_set_error("index build failed: %s", e.what()); return SQ_ERR_STATE; }
}
// [r5] the same, kept in memory (nothing is written): the targets of alignment-based mode
extern "C" int sq_index_build_fasta_mem(const sq_index_opts* opts, const char* fasta_path, const char* decoys_path, sq_index** out) {
  if (!out) { sq_set_error("sq_index_build_fasta_mem: bad arguments"); return SQ_ERR_ARG; }
  try { return index_build_impl(opts, fasta_path, decoys_path, nullptr, out); }
  catch (const std::bad_alloc&) { sq_set_error("out of memory while building the index"); return SQ_ERR_NOMEM; }
  catch (const std::exception& e) { sq_set_error("index build failed: %s", e.what()); return SQ_ERR_STATE; }
}
static int index_build_mem_impl(const sq_index_opts* opts, uint32_t nrefs, const char* const* names,
                                  const char* const* seqs, const uint32_t* lens, uint32_t first_decoy,
                                  const char* outdir, sq_index** out) {
  if (!names || !seqs || !lens || nrefs == 0) { sq_set_error("sq_index_build_mem: bad arguments"); return SQ_ERR_ARG; }
  sq_index* idx = new sq_index();
  int rc = finish_opts(opts, idx); if (rc) { delete idx; return rc; }
  std::vector<std::string> n(nrefs), s(nrefs); std::vector<uint8_t> dec(nrefs, 0);
  for (uint32_t i = 0; i < nrefs; ++i) { n[i] = names[i]; s[i].assign(seqs[i], lens[i]); dec[i] = (i >= first_decoy); }
  std::vector<uint32_t> clen; uint32_t fd = 0;
  prep_refs(opts, n, s, clen, dec, &fd, idx->duplicates, idx);
  rc = build_core(opts, n, s, clen, fd, idx); if (rc) { delete idx; return rc; }
  if (outdir) { rc = sq_index_save(*idx, outdir); if (rc) { delete idx; return rc; } }
  if (out) *out = idx; else delete idx;
  return SQ_OK;
}

static int index_build_impl(const sq_index_opts* opts, const char* fasta_path, const char* decoys_path, const char* outdir, sq_index** out) {
  if (!fasta_path || (!outdir && !out)) { sq_set_error("sq_index_build: bad arguments"); return SQ_ERR_ARG; }
  sq_index* idx = new sq_index();
  int rc = finish_opts(opts, idx); if (rc) { delete idx; return rc; }
  std::vector<std::string> n, s;
  rc = read_fasta(fasta_path, n, s); if (rc) { delete idx; return rc; }
  std::vector<uint8_t> dec(n.size(), 0);
  if (decoys_path && decoys_path[0]) {
    FILE* f = fopen(decoys_path, "r"); if (!f) { sq_set_error("cannot open decoy list '%s'", decoys_path); delete idx; return SQ_ERR_IO; }
    std::unordered_set<std::string> ds; char line[4096];
    while (fgets(line, sizeof(line), f)) {
      std::string l(line);
      while (!l.empty() && isspace((unsigned char)l.back())) l.pop_back();
      if (!l.empty()) ds.insert(l);
    }
    fclose(f);
    for (size_t i = 0; i < n.size(); ++i) if (ds.count(n[i])) dec[i] = 1;
  }
  std::vector<uint32_t> clen; uint32_t fd = 0;
  prep_refs(opts, n, s, clen, dec, &fd, idx->duplicates, idx);
  rc = build_core(opts, n, s, clen, fd, idx);
  if (!rc && outdir) rc = sq_index_save(*idx, outdir);
  if (!rc && out) *out = idx; else delete idx;
  return rc;
}

// ------------------------------------------------------------------------------------------------
// on-disk format: index.bin = header + raw sections; info.json / versionInfo.json keep the key
// names the reference reads back (SalmonIndex.hpp:138-154, SalmonIndexVersionInfo.hpp:93-105).
namespace {
struct Hdr { uint64_t magic; uint32_t version, k, m, nrefs, first_decoy, n_parts; uint64_t num_kmers, nsec; };
template <class T> bool wvec(FILE* f, const std::vector<T>& v) {
  uint64_t n = v.size();
  return fwrite(&n, 8, 1, f) == 1 && (n == 0 || fwrite(v.data(), sizeof(T), n, f) == n);
}
template <class T> bool rvec(FILE* f, std::vector<T>& v, uint64_t file_bytes) {
  uint64_t n;
  if (fread(&n, 8, 1, f) != 1) return false;
  const long at = ftell(f);
  if (at < 0 || n > (file_bytes - (uint64_t)at) / sizeof(T)) return false;   // a section cannot be longer than what is left of the file
  v.resize(n);
  return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}
}

int sq_index_save(const sq_index& idx, const std::string& dir) {
  mkdir(dir.c_str(), 0755);
  std::string p = dir + "/index.bin";
  FILE* f = fopen(p.c_str(), "wb"); if (!f) { sq_set_error("cannot write '%s'", p.c_str()); return SQ_ERR_IO; }
  Hdr h{SQ_INDEX_MAGIC, SQ_INDEX_VERSION, idx.k, idx.m, (uint32_t)idx.names.size(), idx.first_decoy, idx.n_parts, idx.num_kmers, 0};
  bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
  std::vector<char> nm; for (auto& s : idx.names) { nm.insert(nm.end(), s.begin(), s.end()); nm.push_back('\0'); }
  ok = ok && wvec(f, nm) && wvec(f, idx.ref_len) && wvec(f, idx.ref_clen) && wvec(f, idx.ref_accum) && wvec(f, idx.refseq) &&
       wvec(f, idx.useq) && wvec(f, idx.uoff) && wvec(f, idx.ctab_off) && wvec(f, idx.ctab) && wvec(f, idx.part_slot_off) &&
       wvec(f, idx.part_bkt_off) && wvec(f, idx.pilots) && wvec(f, idx.slots) && wvec(f, idx.entries) && wvec(f, idx.skew_keys) && wvec(f,
           idx.skew_vals);
  fclose(f);
  if (!ok) { sq_set_error("short write on '%s'", p.c_str()); return SQ_ERR_IO; }
  f = fopen((dir + "/info.json").c_str(), "w");
  if (f) {
    fprintf(f,
        "{\n  \"index_version\": %u,\n  \"sampling_type\": \"sshash-hip\",\n  \"k\": %u,\n  \"m\": %u,\n  \"num_kmers\": %llu,\n  \"num_contigs\": %llu,\n  \"seq_len\": %llu,\n"
               "  \"num_refs\": %zu,\n  \"first_decoy_index\": %u,\n  \"num_minimizers\": %llu,\n  \"num_super_kmers\": %llu,\n  \"num_skew_kmers\": %llu,\n  \"max_bucket\": %llu,\n"
               "  \"keep_duplicates\": %s,\n  \"SeqHash\": \"%s\",\n  \"NameHash\": \"%s\",\n  \"SeqHash512\": \"%s\",\n  \"NameHash512\": \"%s\",\n  \"DecoySeqHash\": \"%s\",\n  \"DecoyNameHash\": \"%s\"\n}\n",
            SQ_INDEX_VERSION, idx.k, idx.m, (unsigned long long)idx.num_kmers, (unsigned long long)(idx.uoff.size() - 1),
                (unsigned long long)idx.uoff.back(),
                idx.names.size(), idx.first_decoy,
            (unsigned long long)idx.num_minimizers, (unsigned long long)idx.num_superkmers, (unsigned long long)idx.num_skew_kmers,
                (unsigned long long)idx.max_bucket, idx.keep_duplicates ? "true" : "false",
                idx.hashes[0].c_str(), idx.hashes[1].c_str(), idx.hashes[2].c_str(), idx.hashes[3].c_str(), idx.hashes[4].c_str(), idx.hashes[5].c_str());
    fclose(f);
  }
  f = fopen((dir + "/versionInfo.json").c_str(), "w");
  if (f) {
    fprintf(f,
        "{\n  \"indexVersion\": 6,\n  \"hasAuxIndex\": false,\n  \"auxKmerLength\": %u,\n  \"indexType\": 2,\n  \"salmonVersion\": \"1.11.4\"\n}\n",
        idx.k);
    fclose(f);
  }
  f = fopen((dir + "/duplicate_clusters.tsv").c_str(), "w");
  if (f) {
    fprintf(f, "RetainedRef\tDuplicateRef\n");
    for (auto& d : idx.duplicates) fprintf(f, "%s\t%s\n", d.first.c_str(), d.second.c_str());
    fclose(f);
  }
  return SQ_OK;
}

static int index_load_host_impl(const std::string& dir, sq_index** out);
int sq_index_load_host(const std::string& dir, sq_index** out) {   // never lets an exception cross the C ABI
  try { return index_load_host_impl(dir, out); }
  catch (const std::bad_alloc&) { sq_set_error("out of memory while loading the index in '%s'", dir.c_str()); return SQ_ERR_NOMEM; }
  catch (const std::exception& e) { sq_set_error("cannot load the index in '%s': %s", dir.c_str(), e.what()); return SQ_ERR_IO; }
}
static int index_load_host_impl(const std::string& dir, sq_index** out) {
  std::string p = dir + "/index.bin";
  struct stat st;
  // SalmonIndex.hpp:124-131
  if (stat((dir + "/versionInfo.json").c_str(), &st) != 0) {
    sq_set_error("index directory '%s' has no versionInfo.json", dir.c_str());
    return SQ_ERR_IO;
  }
  FILE* f = fopen(p.c_str(), "rb"); if (!f) { sq_set_error("cannot open '%s'", p.c_str()); return SQ_ERR_IO; }
  Hdr h;
  if (fread(&h, sizeof(h), 1, f) != 1 || h.magic != SQ_INDEX_MAGIC || h.version != SQ_INDEX_VERSION) {
    fclose(f);
    sq_set_error("'%s' is not a salmon-hip index of version %u", p.c_str(), SQ_INDEX_VERSION);
    return SQ_ERR_IO;
  }
  struct stat fst; if (stat(p.c_str(), &fst) != 0) { fclose(f); sq_set_error("cannot stat '%s'", p.c_str()); return SQ_ERR_IO; }
  const uint64_t FB = (uint64_t)fst.st_size;
  sq_index* idx = new sq_index();
  idx->k = h.k;
  idx->m = h.m;
  idx->first_decoy = h.first_decoy;
  idx->n_parts = h.n_parts;
  idx->num_kmers = h.num_kmers;
  std::vector<char> nm;
  bool ok = rvec(f, nm, FB) && rvec(f, idx->ref_len, FB) && rvec(f, idx->ref_clen, FB) && rvec(f, idx->ref_accum, FB) && rvec(f, idx->refseq, FB) &&
            rvec(f, idx->useq, FB) && rvec(f, idx->uoff, FB) && rvec(f, idx->ctab_off, FB) && rvec(f, idx->ctab, FB) && rvec(f, idx->part_slot_off, FB) &&
            rvec(f, idx->part_bkt_off, FB) && rvec(f, idx->pilots, FB) && rvec(f, idx->slots, FB) && rvec(f, idx->entries, FB) &&
            rvec(f, idx->skew_keys, FB) && rvec(f, idx->skew_vals, FB);
  fclose(f);
  if (!ok) { delete idx; sq_set_error("truncated index '%s'", p.c_str()); return SQ_ERR_IO; }
  if (!nm.empty() && nm.back() != '\0') { delete idx; sq_set_error("corrupt name table in '%s' (no terminator)", p.c_str()); return SQ_ERR_IO; }
  for (size_t i = 0; i < nm.size();) { idx->names.emplace_back(&nm[i]); i += idx->names.back().size() + 1; }
  if (idx->names.size() != h.nrefs) { delete idx; sq_set_error("corrupt name table in '%s'", p.c_str()); return SQ_ERR_IO; }
  // the sections must agree with the header and with each other before anything indexes them on the device
  const uint64_t U = idx->uoff.empty() ? 0 : idx->uoff.size() - 1;
  const bool sane = h.k >= 3 && h.k <= 31 && (h.k & 1) && h.m >= 1 && h.m <= h.k && h.first_decoy <= h.nrefs &&
      idx->ref_len.size() == h.nrefs && idx->ref_clen.size() == h.nrefs && idx->ref_accum.size() == (size_t)h.nrefs + 1 &&
      !idx->uoff.empty() && idx->ctab_off.size() == idx->uoff.size() && idx->ctab_off.back() == idx->ctab.size() &&
      (idx->uoff.back() + 31) / 32 + 1 <= (uint64_t)idx->useq.size() && (idx->ref_accum.back() + 31) / 32 + 1 <= (uint64_t)idx->refseq.size() &&   // one padding word: the device reads word pairs

      idx->part_slot_off.size() == (size_t)h.n_parts + 1 && idx->part_bkt_off.size() == (size_t)h.n_parts + 1 &&
      (h.n_parts == 0 || (idx->part_slot_off.back() == idx->slots.size() && idx->part_bkt_off.back() == idx->pilots.size())) &&
      idx->skew_keys.size() == idx->skew_vals.size() && (idx->skew_keys.empty() || (idx->skew_keys.size() & (idx->skew_keys.size() - 1)) == 0);
  bool mono = sane;
  for (size_t i = 0; mono && i + 1 < idx->uoff.size(); ++i) mono = idx->uoff[i] <= idx->uoff[i + 1] && idx->ctab_off[i] <= idx->ctab_off[i + 1];
  for (size_t r = 0; mono && r < h.nrefs; ++r) mono = idx->ref_accum[r + 1] - idx->ref_accum[r] == idx->ref_len[r];
  if (mono) for (uint64_t o : idx->ctab) if ((uint32_t)(o >> 32) >= h.nrefs) { mono = false; break; }
  if (!mono) { delete idx; sq_set_error("inconsistent index sections in '%s' (%llu unitigs): rebuild the index", p.c_str(), (unsigned long long)U); return SQ_ERR_IO; }
  // the digests and the duplicate policy live in info.json, where the reference keeps them (SalmonIndex.hpp:138-154); an index written
  // before they existed simply has none
  if (FILE* jf = fopen((dir + "/info.json").c_str(), "r")) {
    std::string js; char bufj[4096]; size_t got; while ((got = fread(bufj, 1, sizeof(bufj), jf)) > 0) js.append(bufj, got); fclose(jf);
    static const char* keys[6] = {"\"SeqHash\"", "\"NameHash\"", "\"SeqHash512\"", "\"NameHash512\"", "\"DecoySeqHash\"", "\"DecoyNameHash\""};
    for (int i = 0; i < 6; ++i) {
      size_t at = js.find(keys[i]); if (at == std::string::npos) continue;
      at = js.find(':', at); if (at == std::string::npos) continue;
      const size_t a = js.find('"', at), b = a == std::string::npos ? a : js.find('"', a + 1);
      if (b != std::string::npos) idx->hashes[i] = js.substr(a + 1, b - a - 1);
    }
    const size_t kd = js.find("\"keep_duplicates\""); if (kd != std::string::npos) idx->keep_duplicates = js.compare(js.find(':', kd) + 1, 5, " true") == 0;
  }
  *out = idx;
  return SQ_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" {
uint32_t sq_index_k(const sq_index* i) { return i->k; }
uint32_t sq_index_m(const sq_index* i) { return i->m; }
uint32_t sq_index_num_refs(const sq_index* i) { return (uint32_t)i->names.size(); }
uint32_t sq_index_first_decoy(const sq_index* i) { return i->first_decoy; }
const char* sq_index_hash(const sq_index* i, int which) { return (i && which >= 0 && which < 6) ? i->hashes[which].c_str() : ""; }
int sq_index_keeps_duplicates(const sq_index* i) { return i && i->keep_duplicates ? 1 : 0; }
const char* sq_index_ref_name(const sq_index* i, uint32_t t) { return t < i->names.size() ? i->names[t].c_str() : nullptr; }
uint32_t sq_index_ref_len(const sq_index* i, uint32_t t) { return t < i->ref_len.size() ? i->ref_len[t] : 0; }
uint32_t sq_index_ref_complete_len(const sq_index* i, uint32_t t) { return t < i->ref_clen.size() ? i->ref_clen[t] : 0; }
int sq_index_is_decoy(const sq_index* i, uint32_t t) { return t >= i->first_decoy; }
uint64_t sq_index_num_unitigs(const sq_index* i) { return i->uoff.empty() ? 0 : i->uoff.size() - 1; }
uint64_t sq_index_num_kmers(const sq_index* i) { return i->num_kmers; }
int sq_index_get_view(const sq_index* i, sq_index_view* v) {
  if (!i || !v) return SQ_ERR_ARG;
  v->k = i->k; v->m = i->m; v->num_refs = (uint32_t)i->names.size(); v->first_decoy = i->first_decoy;
  v->num_unitigs = i->uoff.size() - 1; v->total_unitig_nt = i->uoff.back(); v->num_kmers = i->num_kmers;
  v->total_ref_nt = i->ref_accum.back(); v->num_occ = i->ctab.size();
  v->ref_accum = i->ref_accum.data(); v->ref_len = i->ref_len.data(); v->ref_clen = i->ref_clen.data(); v->refseq = i->refseq.data();
  v->useq = i->useq.data(); v->uoff = i->uoff.data(); v->ctab_off = i->ctab_off.data(); v->ctab = i->ctab.data();
  return SQ_OK;
}
int sq_index_lookup_host(const sq_index* i, uint64_t kmer, uint64_t* unitig, uint32_t* offset, int* is_fw) {
  sq_dict_view d = i->host_view();
  return sq_dict_lookup(d, kmer & sq_kmask(i->k), unitig, offset, is_fw);
}
}
