// sq_internal.h — index layout + k-mer / dictionary primitives shared by the host builder
// (host/index_build.cpp), the host-side query used in tests, and the gfx950 kernels (hip/*.hip).
//
// The dictionary is our own HBM-first layout of the SSHash idea (Pibiri 2022; the reference reaches
// it through PufferfishIndex::getRefPos — call sites SalmonQuantify.cpp:1266-1275): unitigs are
// 2-bit packed into one string pool; every k-mer is keyed by its canonical minimizer (length m);
// a partitioned hash-and-displace MPHF (PTHash-style pilots) maps a minimizer to one 8-byte slot
// record; the record either holds the single string position of that minimizer inline (the common
// case: ONE dependent HBM read after the pilot) or points to a short list of positions; buckets
// larger than SQ_SKEW_THRESH use a k-mer-keyed skew table.  A candidate is always verified against
// the string pool, so absent k-mers can never produce false hits.
#pragma once
#include <stdint.h>
#include <math.h>
#include "../../include/sq_math.h"

#define SQ_INDEX_MAGIC 0x3158444951535153ULL /* "SQSQIDX1" */
#define SQ_INDEX_VERSION 5u
#define SQ_SKEW_THRESH 32u
#define SQ_MPHF_LAMBDA 4.0
#define SQ_MPHF_ALPHA 0.90
#define SQ_MPHF_PART_KEYS 65536u
#define SQ_POS_BITS 40
#define SQ_POS_MASK ((1ULL << SQ_POS_BITS) - 1)
#define SQ_SLOT_INLINE (1ULL << 63)
#define SQ_SLOT_EMPTY (~0ULL)

SQ_HD uint64_t sq_kmask(uint32_t k) { return (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1); }

// reverse-complement of a k-mer stored with base j at bits [2j,2j+1]
SQ_HD uint64_t sq_revcomp(uint64_t x, uint32_t k) {
  x = ~x;
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
  x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
  x = (x >> 32) | (x << 32);
  return x >> (64 - 2 * k);
}

// two consecutive 8-byte words.  On the device they come with ONE 16-byte load: a second load instruction that touches a cache line
// whose fill is still in flight parks the CU's in-order L1 until the fill arrives (every packed pool is padded by a word, and the
// offset tables are read as (x[i], x[i+1]) pairs anyway).
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint64_t sq_u64x2_t __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(8))) sq_u64x2_al8 { sq_u64x2_t v; };
SQ_HD void sq_ld_pair(const uint64_t* p, uint64_t* a, uint64_t* b) { const sq_u64x2_t v = reinterpret_cast<const sq_u64x2_al8*>(p)->v; *a = v.x; *b = v.y; }
#else
SQ_HD void sq_ld_pair(const uint64_t* p, uint64_t* a, uint64_t* b) { *a = p[0]; *b = p[1]; }
#endif

// fetch `n` (<=32) bases starting at nt position p from a packed pool
SQ_HD uint64_t sq_fetch_bases(const uint64_t* pool, uint64_t p, uint32_t n) {
  uint64_t w = p >> 5;
  uint32_t sh = (uint32_t)(p & 31) * 2;
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t a, b; sq_ld_pair(pool + w, &a, &b);
  const uint64_t lo = (a >> sh) | (sh ? (b << (64 - sh)) : 0ULL);   // bits of b beyond 2n are masked off below
#else
  uint64_t lo = pool[w] >> sh;
  if (sh != 0 && sh + 2 * n > 64) lo |= pool[w + 1] << (64 - sh);
#endif
  return lo & sq_kmask(n);
}
SQ_HD uint32_t sq_fetch_base(const uint64_t* pool, uint64_t p) {
  return (uint32_t)(pool[p >> 5] >> ((p & 31) * 2)) & 3u;
}

// G/C content through a sampled prefix: gcpre[w] = number of G/C bases in the words before word w of a 2-bit pool (A 0, C 1, G 2, T 3:
// a base is G or C iff its two bits differ).  sq_gc_before(pool, gcpre, p) = G/C among pool positions [0, p).
SQ_HD uint32_t sq_popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }   // clang lowers it to s_bcnt / v_bcnt on the device
SQ_HD uint32_t sq_gc_word(uint64_t w) { return sq_popc64((w ^ (w >> 1)) & 0x5555555555555555ULL); }
SQ_HD uint64_t sq_gc_before(const uint64_t* pool, const uint32_t* gcpre, uint64_t p) {
  const uint64_t w = p >> 5; const uint32_t r = (uint32_t)(p & 31);
  uint64_t c = gcpre[w];
  if (r) c += sq_popc64((pool[w] ^ (pool[w] >> 1)) & 0x5555555555555555ULL & ((1ULL << (2 * r)) - 1));
  return c;
}

// Transcript::gcDesc (include/salmon/internal/model/Transcript.hpp:294-341): GC percentage of the fragment [s, e] on a transcript that
// starts at pool position g and has refLen bases, and of its 5' / 3' context windows (3 bases outside + 2 inside each end).
// Returns false when there is no context.  frag / context percentages are lrint()-rounded as in the reference.
#define SQ_GC_FRAG_BINS 25   /* numFragGCBins, SalmonDefaults.hpp:105 */
#define SQ_GC_COND_BINS 3    /* numConditionalGCBins, :106 */
SQ_HD bool sq_gc_desc(const uint64_t* pool, const uint32_t* gcpre, uint64_t g, int32_t refLen, int32_t s, int32_t e, int32_t* fragFrac, int32_t* ctxFrac) {
  const uint64_t g0 = sq_gc_before(pool, gcpre, g);
  #define SQ_GCI(i) ((int64_t)(sq_gc_before(pool, gcpre, g + (uint64_t)(i) + 1) - g0))   /* GCCount_[i]: G/C in [0, i] */
  const int lastPos = refLen - 1;
  const int64_t cs = (s > 0) ? SQ_GCI(s - 1) : 0, ce = SQ_GCI(e);
  int fs = s - 4, fe = s + 1, ts = e - 2, te = e + 3;
  const bool fpL = fs >= 0, fpR = fe <= lastPos, tpL = ts >= 0, tpR = te <= lastPos;
  const int64_t fps = fpL ? SQ_GCI(fs) : 0, fpe = fpR ? SQ_GCI(fe) : ce, tps = tpL ? SQ_GCI(ts) : 0, tpe = tpR ? SQ_GCI(te) : ce;
  #undef SQ_GCI
  fs = fs < 0 ? 0 : fs; fe = fe > lastPos ? lastPos : fe; ts = ts < 0 ? 0 : ts; te = te > lastPos ? lastPos : te;
  const int fpSize = !fpL ? (fe + 1) : (fe - fs), tpSize = !tpL ? (te + 1) : (te - ts);
  const double contextSize = (double)(fpSize + tpSize);
  if (contextSize == 0) return false;
  *fragFrac = (int32_t)rint((100.0 * (double)(ce - cs)) / (double)(e - s + 1));
  *ctxFrac = (int32_t)rint(100.0 * ((double)((fpe - fps) + (tpe - tps)) / contextSize));
  return true;
}
SQ_HD int32_t sq_gc_frag_bin(int32_t fragFrac) { const double w = 100.0 / SQ_GC_FRAG_BINS; const int32_t b = (int32_t)((double)fragFrac / w); return b < SQ_GC_FRAG_BINS - 1 ? b : SQ_GC_FRAG_BINS - 1; }   // GCDesc::fragBin(n)
SQ_HD int32_t sq_gc_ctx_bin(int32_t ctxFrac) { const double w = 100.0 / SQ_GC_COND_BINS; const int32_t b = (int32_t)((double)ctxFrac / w); return b < SQ_GC_COND_BINS - 1 ? b : SQ_GC_COND_BINS - 1; }       // GCDesc::contextBin(n)

SQ_HD uint32_t sq_fastrange32(uint32_t x, uint32_t n) { return (uint32_t)(((uint64_t)x * (uint64_t)n) >> 32); }

// Minimizer order: a cheap 32-bit hash of the canonical m-mer, ties broken by the m-mer value (so a
// k-mer and its reverse complement always agree).  The probe kernel evaluates it k-m+1 times per
// lookup; with the 64-bit murmur finaliser the lookup was VALU-bound (k_seed reached ~30 % of the
// measured random-sector gather rate of the chip), this form costs ~8 integer ops.
SQ_HD uint32_t sq_mhash(uint64_t c) {
  uint32_t x = (uint32_t)c * 0x9E3779B1u + 0x7F4A7C15u;
  x ^= (uint32_t)(c >> 32) * 0x85EBCA77u;
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
  return x;
}
// canonical minimizer of a k-mer: the canonical m-mer with the smallest (sq_mhash, value); returns the value.
SQ_HD uint64_t sq_minimizer(uint64_t kmer, uint64_t rc, uint32_t k, uint32_t m) {
  const uint64_t mm = sq_kmask(m);
  uint32_t best = 0xFFFFFFFFu; uint64_t bestv = ~0ULL;
  const uint32_t w = k - m;
  for (uint32_t j = 0; j <= w; ++j) {
    uint64_t a = (kmer >> (2 * j)) & mm;
    uint64_t b = (rc >> (2 * (w - j))) & mm;
    uint64_t c = a < b ? a : b;
    uint32_t h = sq_mhash(c);
    if (h < best || (h == best && c < bestv)) { best = h; bestv = c; }
  }
  return bestv;
}

struct sq_dict_view {
  uint32_t k, m;
  uint32_t n_parts;
  const uint64_t* part_slot_off;  // [n_parts+1]
  const uint32_t* part_bkt_off;   // [n_parts+1]
  const uint16_t* pilots;         // [sum buckets]
  const uint64_t* slots;          // [sum slots]
  const uint64_t* entries;        // list entries: unitig<<33 | minimizer pool position
  const uint64_t* skew_keys;      // open addressing, canonical k-mer or ~0
  const uint64_t* skew_vals;      // unitig<<33 | k-mer start pool position
  uint64_t skew_mask;             // capacity-1 (0 if no skew table)
  const uint64_t* useq;           // string pool
  const uint64_t* uoff;           // [U+1]
  uint64_t num_unitigs;
  const uint64_t* uinfo;          // [r3] device only (nullptr on the host): [U+1][2] = {uoff[u], ctab_off[u]} interleaved — a hit needs the bounds of unitig u in both
                                  // tables, and here they are 32 contiguous bytes (one 64-byte sector three times out of four) instead of two sectors
  const uint64_t* kfilter;        // k-mer membership filter (device only; nullptr = none): word-blocked Bloom, SQ_KF_BITS_PER_KEY bits per k-mer
  uint64_t kfilter_words;
  const uint64_t* mtab;           // [r5] device only (nullptr = none): minimizer -> slot record in ONE dependent sector (sq_mtab_*, below) instead of pilot -> slot
  uint64_t mtab_buckets;
};

// k-mer membership filter in front of the dictionary.  ~80 % of the probes of a read are misses (the mismatchSeedSkip walk across a
// sequencing error looks up ~10 k-mers that contain the error); a miss used to cost the whole minimizer scan (~300 integer ops) and
// four dependent 64-byte sectors (pilot, slot record, two string-pool candidates).  One 8-byte word of this filter answers "not in
// the index" for > 99 % of them: the word is picked by the hash of the canonical k-mer, four of its 64 bits are the k-mer's
// signature.  No false negatives (every k-mer of every unitig is inserted when the index is uploaded: index_dev.hip), so results
// are unchanged; false positives (~0.6 % at 16 bits per k-mer) just take the full path.  288 GB of HBM buys this: 2 bytes per k-mer.
#define SQ_KF_BITS_PER_KEY 32
// [r3] Blocked by MINIMIZER: the filter is an array of 64-byte blocks (8 words); a k-mer's block is chosen by the hash of its minimizer, its
// word inside the block and its four signature bits by the hash of the k-mer itself.  The ~12 consecutive k-mers of a read that share a
// minimizer — in particular the run of missing k-mers the mismatchSeedSkip walk probes across a sequencing error — therefore ask the SAME
// 64-byte line: one HBM sector per run instead of one per probe (round 2 measured 1.66x the algorithmic traffic in k_seed, all of it filter
// sectors).  The price is the minimizer scan before the filter (ALU the kernel had idle) and 4 bytes per k-mer instead of 2.
#define SQ_KF_BLOCK_WORDS 8
SQ_HD uint64_t sq_kf_hash(uint64_t canonical_kmer) { return sq_mix64(canonical_kmer ^ 0xA24BAED4963EE407ULL); }
SQ_HD uint64_t sq_kf_mask(uint64_t h) { return (1ULL << (h & 63)) | (1ULL << ((h >> 6) & 63)) | (1ULL << ((h >> 12) & 63)) | (1ULL << ((h >> 18) & 63)); }
SQ_HD uint64_t sq_kf_word(uint64_t h, uint64_t nwords) {   // mulhi(h, nwords): the high bits of h choose the word, the low 24 the signature
  const uint64_t a_lo = (uint32_t)h, a_hi = h >> 32, b_lo = (uint32_t)nwords, b_hi = nwords >> 32;   // portable 64 x 64 -> high 64
  const uint64_t p0 = a_lo * b_lo, p1 = a_lo * b_hi, p2 = a_hi * b_lo, p3 = a_hi * b_hi;
  const uint64_t mid = (p0 >> 32) + (uint32_t)p1 + (uint32_t)p2;
  return p3 + (p1 >> 32) + (p2 >> 32) + (mid >> 32);
}

SQ_HD uint64_t sq_kf_word_of(uint64_t mini, uint64_t h_kmer, uint64_t nblocks) {   // word index of a k-mer with minimizer `mini` and filter hash h_kmer
  return sq_kf_word(sq_mix64(mini ^ 0x6A09E667F3BCC909ULL), nblocks) * SQ_KF_BLOCK_WORDS + ((h_kmer >> 24) & (SQ_KF_BLOCK_WORDS - 1));
}

// displacement of a key hash by its bucket's pilot (PTHash: hash(key) xor hash(pilot), then reduce); one multiply each
SQ_HD uint64_t sq_pilot_mix(uint64_t pilot) { return (pilot + 1) * 0x9E3779B97F4A7C15ULL; }
SQ_HD uint32_t sq_slot_mix(uint64_t h, uint64_t pm) { return (uint32_t)(((h ^ pm) * 0xD6E8FEB86659FD93ULL) >> 32); }

SQ_HD uint64_t sq_mphf_slot(const sq_dict_view& d, uint64_t minimizer) {
  uint64_t h = sq_mix64(minimizer ^ 0x9E3779B97F4A7C15ULL);
  uint32_t part = sq_fastrange32((uint32_t)(h >> 32), d.n_parts);
  uint32_t b0 = d.part_bkt_off[part], nb = d.part_bkt_off[part + 1] - b0;
  uint64_t s0 = d.part_slot_off[part];
  uint32_t ns = (uint32_t)(d.part_slot_off[part + 1] - s0);
  uint32_t bkt = sq_fastrange32((uint32_t)h, nb);
  uint64_t pilot = d.pilots[b0 + bkt];
  return s0 + sq_fastrange32(sq_slot_mix(h, sq_pilot_mix(pilot)), ns);
}

// A minimizer occurrence is stored as (unitig id, ABSOLUTE pool position of the minimizer):
//   entry = unitig << SQ_APOS_BITS | pool position          (63 bits)
// so after the slot record arrives, the pool words and the unitig bounds (uoff[u], uoff[u+1]) can be
// fetched in parallel: the dependent chain of a lookup is pilot -> slot -> {pool, bounds}.
#define SQ_APOS_BITS 33
#define SQ_APOS_MASK ((1ULL << SQ_APOS_BITS) - 1)
#define SQ_ENT_MASK ((1ULL << 63) - 1)
#define SQ_UOFF_BITS 30   /* unitig ids and in-unitig offsets stay below 2^30 */

// Try one candidate: k-mer starting at pool position `sp` inside unitig u. Returns 1 if the pool holds
// `kmer` (fw) or `rc` there and the k-mer lies inside the unitig.
SQ_HD int sq_dict_try(const sq_dict_view& d, uint64_t kmer, uint64_t rc, uint64_t u, int64_t sp,
                      uint64_t* unitig, uint32_t* off, int* fw) {
  if (sp < 0) return 0;
  const uint64_t s = sq_fetch_bases(d.useq, (uint64_t)sp, d.k);
  int f;
  // most failed probes end here: the unitig bounds are only fetched for a string match
  if (s == kmer) f = 1;
  else if (s == rc) f = 0;
  else return 0;
  uint64_t b, e;
  if (d.uinfo) { b = d.uinfo[2 * u]; e = d.uinfo[2 * u + 2]; } else sq_ld_pair(d.uoff + u, &b, &e);
  if ((uint64_t)sp < b || (uint64_t)sp + d.k > e) return 0;
  *unitig = u; *off = (uint32_t)((uint64_t)sp - b); *fw = f;
  return 1;
}

// Full dictionary query. kmer in read orientation; on success fw tells whether the read k-mer
// equals the unitig's forward string at (unitig, off).  KT/MT > 0 fix k and m at compile time (the
// probe kernel is instruction-issue bound: constant shifts/masks and fully unrolled window loops
// roughly halve its instruction count); 0 = take them from the view.
// one pass over the window of a k-mer: its minimizer (the canonical m-mer with the smallest (sq_mhash, value)) and the set of positions j
// that hold it (bit j of *at)
template <int KT, int MT>
SQ_HD void sq_min_scan(const sq_dict_view& d, uint64_t kmer, uint64_t rc, uint64_t* mini_out, uint32_t* at_out) {
  const uint32_t k = KT ? (uint32_t)KT : d.k, m = MT ? (uint32_t)MT : d.m, w = k - m;
  const uint64_t mm = sq_kmask(m);
  uint32_t best = 0xFFFFFFFFu; uint64_t mini = ~0ULL; uint32_t at = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (uint32_t j = 0; j <= w; ++j) {
    const uint64_t a = (kmer >> (2 * j)) & mm;
    const uint64_t b = (rc >> (2 * (w - j))) & mm;
    const uint64_t c = a < b ? a : b;
    const uint32_t h = sq_mhash(c);
    if (h < best || (h == best && c < mini)) { best = h; mini = c; at = 0; }
    if (c == mini) at |= 1u << j;
  }
  *mini_out = mini; *at_out = at;
}
// [r5] The minimizer table (device only, built at upload from the MPHF's own answers: hip/index_dev.hip).  The MPHF costs two dependent
// 64-byte sectors per hit (pilot, then slot record); the table keeps (minimizer, slot record) pairs in 64-byte buckets of four chosen by
// the minimizer's hash, so the record of a present minimizer arrives with ONE sector (linear probing over buckets for the few that
// overflow; a bucket with a free place ends the search).  Same records, so everything behind it is unchanged.
#define SQ_MTAB_BUCKET 4u            /* (key, record) pairs per 64-byte bucket */
#define SQ_MTAB_EMPTY (~0ULL)
SQ_HD uint64_t sq_mtab_bucket_of(uint64_t mini, uint64_t nbuckets) { return sq_kf_word(sq_mix64(mini ^ 0xBB67AE8584CAA73BULL), nbuckets); }
SQ_HD uint64_t sq_mtab_find(const uint64_t* tab, uint64_t nbuckets, uint64_t mini) {   // the slot record of `mini`, SQ_SLOT_EMPTY if it has none
  uint64_t b = sq_mtab_bucket_of(mini, nbuckets);
  for (;;) {
    const uint64_t* q = tab + b * (2 * SQ_MTAB_BUCKET);
    uint64_t k0, r0, k1, r1, k2, r2, k3, r3;   // four 16-byte loads of one sector, in flight together
    sq_ld_pair(q, &k0, &r0); sq_ld_pair(q + 2, &k1, &r1); sq_ld_pair(q + 4, &k2, &r2); sq_ld_pair(q + 6, &k3, &r3);
    if (k0 == mini) return r0;
    if (k1 == mini) return r1;
    if (k2 == mini) return r2;
    if (k3 == mini) return r3;
    if (k0 == SQ_MTAB_EMPTY || k1 == SQ_MTAB_EMPTY || k2 == SQ_MTAB_EMPTY || k3 == SQ_MTAB_EMPTY) return SQ_SLOT_EMPTY;
    if (++b == nbuckets) b = 0;
  }
}

// the dictionary walk for a k-mer whose minimizer scan has been done and whose minimizer's slot record is at hand: record -> {string pool, unitig bounds}
template <int KT, int MT>
SQ_HD int sq_dict_lookup_rec(const sq_dict_view& d, uint64_t kmer, uint64_t rc, uint64_t rec, uint32_t at, uint64_t* unitig, uint32_t* off, int* fw) {
  const uint32_t k = KT ? (uint32_t)KT : d.k, m = MT ? (uint32_t)MT : d.m, w = k - m;
  if (rec == SQ_SLOT_EMPTY) return 0;
  const bool inl = (rec & SQ_SLOT_INLINE) != 0;     // the common case: the record IS the single occurrence (no pointer, no scratch)
  uint64_t nent = 1; const uint64_t* ent = nullptr;
  if (!inl) {
    nent = rec >> SQ_POS_BITS; ent = d.entries + (rec & SQ_POS_MASK);
    if (nent > SQ_SKEW_THRESH) {  // heavy bucket -> skew table keyed by canonical k-mer
      if (!d.skew_mask) return 0;
      uint64_t can = kmer < rc ? kmer : rc;
      uint64_t h = sq_mix64(can) & d.skew_mask;
      for (;;) {
        uint64_t kk = d.skew_keys[h];
        if (kk == ~0ULL) return 0;
        if (kk == can) {
          uint64_t v = d.skew_vals[h] & SQ_ENT_MASK;
          return sq_dict_try(d, kmer, rc, v >> SQ_APOS_BITS, (int64_t)(v & SQ_APOS_MASK), unitig, off, fw);
        }
        h = (h + 1) & d.skew_mask;
      }
    }
  }
  // positions j (in the read-orientation k-mer) where the canonical m-mer equals the minimizer
  while (at) {
    const uint32_t j = (uint32_t)__builtin_ctz(at); at &= at - 1;
    for (uint64_t e = 0; e < nent; ++e) {
      const uint64_t ev = (inl ? rec : ent[e]) & SQ_ENT_MASK;
      uint64_t u = ev >> SQ_APOS_BITS;
      int64_t A = (int64_t)(ev & SQ_APOS_MASK);  // minimizer position in the pool
      // same strand: k-mer starts at A - j ; opposite strand: starts at A - (w - j)
      if (sq_dict_try(d, kmer, rc, u, A - (int64_t)j, unitig, off, fw)) return 1;
      if (j != w - j && sq_dict_try(d, kmer, rc, u, A - (int64_t)(w - j), unitig, off, fw)) return 1;
    }
  }
  return 0;
}
// the dictionary walk for a k-mer whose minimizer scan has been done: pilot -> slot record -> {string pool, unitig bounds}
template <int KT, int MT>
SQ_HD int sq_dict_lookup_pre(const sq_dict_view& d, uint64_t kmer, uint64_t rc, uint64_t mini, uint32_t at, uint64_t* unitig, uint32_t* off, int* fw) {
  return sq_dict_lookup_rec<KT, MT>(d, kmer, rc, d.slots[sq_mphf_slot(d, mini)], at, unitig, off, fw);
}
// Full dictionary query. kmer in read orientation; on success fw tells whether the read k-mer
// equals the unitig's forward string at (unitig, off).  KT/MT > 0 fix k and m at compile time (the
// probe kernel is instruction-issue bound: constant shifts/masks and fully unrolled window loops
// roughly halve its instruction count); 0 = take them from the view.
template <int KT, int MT>
SQ_HD int sq_dict_lookup_t(const sq_dict_view& d, uint64_t kmer, uint64_t* unitig, uint32_t* off, int* fw, bool filtered = false) {
  const uint32_t k = KT ? (uint32_t)KT : d.k;
  const uint64_t rc = sq_revcomp(kmer, k);
  uint64_t mini; uint32_t at; sq_min_scan<KT, MT>(d, kmer, rc, &mini, &at);
  if (d.kfilter && !filtered) {   // membership filter first (`filtered`: the caller has asked it already)
    const uint64_t h = sq_kf_hash(kmer < rc ? kmer : rc), msk = sq_kf_mask(h);
    if ((d.kfilter[sq_kf_word_of(mini, h, d.kfilter_words / SQ_KF_BLOCK_WORDS)] & msk) != msk) return 0;
  }
  return sq_dict_lookup_pre<KT, MT>(d, kmer, rc, mini, at, unitig, off, fw);
}
SQ_HD int sq_dict_lookup(const sq_dict_view& d, uint64_t kmer, uint64_t* unitig, uint32_t* off, int* fw) {
  return sq_dict_lookup_t<0, 0>(d, kmer, unitig, off, fw);
}
