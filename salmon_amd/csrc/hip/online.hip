// hip/online.hip — online model + equivalence-class accumulation on gfx950 (seam B2).
//
// Replaces processMiniBatch (reference src/quant/SalmonQuantify.cpp:426-1023),
// FragmentLengthDistribution (src/model/FragmentLengthDistribution.cpp:23-186), the transcript
// mass / count atomics (include/salmon/internal/model/Transcript.hpp:136-141,210-217) and
// EquivalenceClassBuilder::addGroup/finish (include/salmon/internal/quant/EquivalenceClassBuilder.hpp).
//
// The reference mutates shared state from N threads in arrival order (nondeterministic).  Here a
// mini-batch (5000 fragments, SalmonQuantify.cpp:150) is the unit of synchrony: every fragment of a
// mini-batch reads the model as of the batch start; increments are accumulated with INTEGER atomics
// (fixed-point masses, histogram counts, fixed-point class weights) and applied at the batch end,
// so results do not depend on thread order and equal the CPU checker bit for bit (SPEC §D1).
// The equivalence-class table is an HBM open-addressing table keyed by a 128-bit label hash.
#include "ctx.h"
#include "scan_kernels.h"
#include "sq_rng.h"
#include "../host/posbias.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include <chrono>
#include <cstdlib>
#include <cstdio>

struct sq_online_dev {
  uint32_t M = 0;
  // model
  sq_dbuf<double> hist, cpmf, ccmf, ambig, mass, prior_mass, log_eff_len, fm_table, cfac, tlc; sq_dbuf<double> scal;  // scal[0]=totMass
  sq_dbuf<uint32_t> touched, touched_n, tflag;   // transcripts whose mass changed in the current group of mini-batches: two lists (group parity), [2*M] + [2]; tflag[M] = already listed
  uint32_t inflight = 1;                          // W: mini-batches per model snapshot (SPEC §D1); mass_acc is [M][W], fld_cnt [W][1024]
  // ctr: [0]=numAssigned [1]=burnedIn [2]=minLen [3]=cached [4]=pending_finalize [5]=numCompatible
  sq_dbuf<unsigned long long> mass_acc, uniq, total, lib_counts;
  sq_dbuf<uint32_t> fld_cnt;
  // [r5] alignment-based input with the CIGAR error model (AlignmentModel.cpp): log-space cells [2][bins][82][82] and row sums [2][bins][82], the increments of the
  // W mini-batches of a group [W][2][bins][82][82] (fixed-point sums of exp(p)), one byte per alignment: mini-batch slot + 1 where the update was drawn
  sq_dbuf<double> err_cell, err_row; sq_dbuf<unsigned long long> err_acc; sq_dbuf<uint8_t> err_flag; uint32_t err_bins = 0;
  sq_dbuf<unsigned long long> ctr;
  // per big batch
  sq_dbuf<uint8_t> has_compat;
  struct PreAln;
  sq_dbuf<uint8_t> pre, dyn;   // dyn: DynAln per alignment (the post-burn-in split, k_frag_static -> k_frag_dynamic)
  sq_dbuf<uint8_t> gflag;   // [r4] the transcripts every group of a batch touches (TouchArgs)
  sq_dbuf<double> alp;
  sq_dbuf<uint32_t> assigned_flag;
  sq_dbuf<uint64_t> assigned_prefix;
  sq_dbuf<unsigned long long> awq;
  sq_dbuf<uint32_t> abin;
  sq_dbuf<uint64_t> rh1, rh2;
  sq_dbuf<uint32_t> rslot;
  sq_dbuf<uint8_t> scan_tmp;
  // eq table
  // [0] labels used, [1] classes, [2] overflow flag
  uint64_t tcap = 0;
  sq_dbuf<unsigned long long> tk1, tk2, tcount, tpool;
  sq_dbuf<uint32_t> tn;
  sq_dbuf<uint32_t> pool_tid, pool_bin;
  sq_dbuf<unsigned long long> pool_wq;
  sq_dbuf<unsigned long long> pool_cursor;
  uint64_t pool_cap = 0;
  // export of the table in canonical order: persistent device buffers + a pinned host staging area.  The export
  // (kernels + D2H) runs back to back with the end of the eq stage on the first sq_eq_finish call; the second call
  // (caller's arrays now allocated) is a host copy.  Touching the GPU again after the short idle gap in between
  // was measured to stall 20-35 ms on MI355X (first dispatch after heavy load + ~2 ms idle).
  struct eq_export {
    sq_dbuf<unsigned long long> keys, keys2, d_wq, d_cnt, d_h1, d_h2, d_ctr;
    sq_dbuf<uint32_t> slots, slots2, nlab, d_tid, d_bins, d_tie;
    sq_dbuf<uint64_t> d_off;
    sq_dbuf<double> d_w;
    sq_dbuf<uint8_t> tmp;
    // staged model summary (mass, uniq, total, logEffLen) at host + model_off
    uint8_t* host = nullptr;
    size_t host_cap = 0;
    uint64_t E = 0, L = 0;
    bool valid = false, model_valid = false;
    size_t model_off = 0;
    void release() { keys.free_(); keys2.free_(); d_wq.free_(); d_cnt.free_(); d_h1.free_(); d_h2.free_(); d_ctr.free_(); slots.free_(); slots2.free_(); nlab.free_(); d_tid.free_(); d_bins.free_(); d_tie.free_(); d_off.free_(); d_w.free_(); tmp.free_();
                     if (host) (void)hipHostFree(host); host = nullptr; host_cap = 0; valid = false; }
  } exp;
  sq_dbuf<uint32_t> merge_slot;
  std::vector<double> fm_host;
  uint64_t num_observed = 0, num_mapped_ub = 0, batch_no = 0, group_no = 0; bool burned_known = false;
  // `-l A` (SPEC §D8): per-format sample counts of the mini-batches seen so far while detection is active
  bool detect_active = false, detected = false; uint64_t det_counts[64] = {0}; uint64_t det_samples = 0;
  sq_dbuf<uint32_t> mb_samples;   // [mini-batches of a batch][64]
  sq_dbuf<int32_t> cmeans;   // conditional fragment-length means of the prior distribution [1001] (single-end --gcBias)
  sq_dbuf<uint16_t> posbin; sq_dbuf<unsigned long long> pos_obs; sq_dbuf<uint8_t> lenclass;   // --posBias: per alignment the 5' bin | 3' bin << 8 (class * 20 + bin, 255 = none); observed masses [2][100], fixed point 2^-32; Transcript::lengthClassIndex
  sq_dbuf<uint8_t> gcbin; sq_dbuf<unsigned long long> gc_obs;   // --gcBias: GC bin (ctx * 25 + frag bin, 255 = none) per alignment of the batch; observed masses [75], fixed point 2^-32
  sq_dbuf<uint64_t> assigned_prefix_b;   // bounds scratch after a format switch
  sq_dbuf<unsigned long long> seq_obs;   // --seqBias: observed context counts [FW 576 | RC 576] + [1152] fragments sampled so far
  sq_dbuf<uint32_t> seq_flag; sq_dbuf<uint64_t> seq_pref, seq_code;
};

namespace {
const int TB = 256;
inline uint32_t nblk(uint64_t n) { return (uint32_t)((n + TB - 1) / TB); }
#define EQ_EMPTY (~0ULL)

__device__ inline double dev_u01(uint64_t seed, uint64_t read, uint64_t aln) {
  uint64_t x = sq_mix64(seed ^ sq_mix64(read * 0x9E3779B97F4A7C15ULL + aln + 1));
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
__device__ inline uint32_t frag_len_pedantic(const sq_aln& a, uint32_t txpLen) {  // ReadPair.hpp:149-168 semantics
  if (a.mate_status != SQ_MS_PAIRED_END_PAIRED || a.fwd == a.mate_fwd) return 0;
  int32_t T = (int32_t)txpLen;
  int32_t p1 = a.fwd ? a.pos : a.mate_pos; p1 = p1 < 0 ? 0 : p1; p1 = p1 > T ? T : p1;
  int32_t p2 = a.fwd ? a.mate_pos + (int32_t)a.mate_len : a.pos + (int32_t)a.read_len; p2 = p2 < 0 ? 0 : p2; p2 = p2 > T ? T : p2;
  return (uint32_t)(p1 > p2 ? p1 - p2 : p2 - p1);
}
// library compatibility (SalmonUtils.cpp:138-298)
__device__ inline bool is_compatible(uint8_t fid, uint8_t et, uint8_t eo, uint8_t es, bool fwd, uint8_t ms) {
  if (ms != SQ_MS_PAIRED_END_PAIRED) {
    switch (ms) {
      case SQ_MS_SINGLE_END: return fwd ? (es == 4 || es == 2) : (es == 4 || es == 3);
      case SQ_MS_PAIRED_END_LEFT: if (eo == 0) return es == 4 || (es == 2 && fwd) || (es == 3 && !fwd);
      return fwd ? (es == 4 || es == 0) : (es == 4 || es == 1);
      case SQ_MS_PAIRED_END_RIGHT: if (eo == 0) return es == 4 || (es == 2 && fwd) || (es == 3 && !fwd);
      return fwd ? (es == 4 || es == 1) : (es == 4 || es == 0);
      default: return false;
    }
  }
  uint8_t ot = fid & 1, oo = (fid >> 1) & 3, os = fid >> 3;
  if (ot != 1) return false;
  if (eo != oo) return false;
  return es == 4 || es == os;
}

struct OnlineView {
  uint32_t M, W; const uint32_t* ref_len; const uint32_t* ref_clen; double* tlc;
  double* hist;
  double* cpmf;
  double* ccmf;
  const double* ambig;
  double* mass;
  const double* prior_mass;
  double* log_eff_len;
  double* scal;
  double* cfac;
  unsigned long long* mass_acc;
  unsigned long long* uniq;
  unsigned long long* total;
  unsigned long long* lib_counts;
  uint32_t* fld_cnt;
  unsigned long long* ctr;
  uint32_t* touched; uint32_t* touched_n; uint32_t* tflag;
  unsigned long long* gc_obs;   // nullptr unless --gcBias
  unsigned long long* pos_obs; const uint16_t* posbin;   // nullptr unless --posBias
  double* err_cell; double* err_row; unsigned long long* err_acc; uint8_t* err_flag; uint32_t err_bins;   // nullptr / 0 unless the error model is on
};
// observedPosBiasFwd / RC [lengthClassIndex].addMass(pos, RefLength, aln.logProb) — SalmonQuantify.cpp:895-934; the bins were chosen by k_pre_aln
__device__ inline void pos_observe(const OnlineView& V, uint64_t ai, double pr) {
  const uint32_t pb = V.posbin[ai]; const unsigned long long q = (unsigned long long)sq_to_fixed(pr, 32);
  if ((pb & 255u) != 255u) (void)__hip_atomic_fetch_add(&V.pos_obs[pb & 255u], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((pb >> 8) != 255u) (void)__hip_atomic_fetch_add(&V.pos_obs[100u + (pb >> 8)], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LibraryTypeDetector::addSample (LibraryTypeDetector.hpp:155-160): every alignment whose observed format has the library's read type
// is a sample; block b histograms the samples of mini-batch b by format id
__global__ void k_mb_samples(uint32_t n, uint32_t mb, const uint64_t* __restrict__ aln_off, const sq_aln* __restrict__ aln, uint32_t lib_type,
                             uint32_t* __restrict__ out /*[nmb][64]*/) {
  __shared__ uint32_t h[64];
  if (threadIdx.x < 64) h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t r0 = (uint64_t)blockIdx.x * mb, r1 = min((uint64_t)n, r0 + mb);
  for (uint64_t i = aln_off[r0] + threadIdx.x; i < aln_off[r1]; i += blockDim.x) {
    const uint32_t f = aln[i].format_id;
    if ((f & 1u) == lib_type) atomicAdd(&h[f & 63u], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 64) out[(size_t)blockIdx.x * 64 + threadIdx.x] = h[threadIdx.x];
}

__global__ void k_flag_compat(uint32_t n, uint32_t r_start, const uint64_t* __restrict__ aln_off, const sq_aln* __restrict__ aln, sq_quant_opts o,
    uint32_t* __restrict__ flag) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  if (r == n || r < r_start) { flag[r] = 0; return; }   // rows before r_start belong to an earlier segment of the batch (format switch, SPEC §D8)
  uint32_t f = 0;
  for (uint64_t i = aln_off[r]; i < aln_off[r + 1]; ++i) {
    const sq_aln a = aln[i];
    bool c = is_compatible(a.format_id, o.lib_type, o.lib_orientation, o.lib_strand, a.fwd, a.mate_status);
    if (c || !o.ignore_incompat) {
      f = 1;
      break;
    }
  }
  flag[r] = f;
}

__device__ inline void label_hash_step(uint64_t& a, uint64_t& b, uint32_t x) {
  a = sq_mix64(a ^ (uint64_t)x) + 0x9E3779B97F4A7C15ULL;
  b = sq_mix64(b + (uint64_t)x * 0xD6E8FEB86659FD93ULL) ^ (b >> 29);
}

// Model-independent per-alignment terms, computed once per mapped batch (thread per alignment):
// logFragCov, the paired-end start-position term, log(RefLength), the fragment lengths and the
// compatibility verdict.  Everything a mini-batch still has to evaluate per alignment is then a
// table lookup (FLD pmf/cmf, cached transcript log-mass) plus the in-order log-sum chains.
struct PreAln { double c_cov; double c_start; uint32_t flen; uint16_t fl_ped, max_fl; uint16_t tl; uint8_t flags, fmt; uint32_t tid; };  // 32 B: everything about an alignment that does not depend on the evolving model
enum { PF_KEEP = 1, PF_COMPAT = 2, PF_PE_START = 4, PF_ORPHAN_MODEL = 8, PF_UNEXP_ORPHAN = 16,
    PF_FLEN_IN_REF = 32 /* flen < refLength (refLength = max(RefLength, 1)) */ };

__global__ void k_pre_aln(uint64_t na, const sq_aln* __restrict__ aln, const uint32_t* __restrict__ ref_len,
    const uint32_t* __restrict__ ref_clen, sq_quant_opts o,
    PreAln* __restrict__ pre, const uint64_t* __restrict__ refseq, const uint32_t* __restrict__ gcpre, const uint64_t* __restrict__ ref_accum,
    uint8_t* __restrict__ gcbin, const int32_t* __restrict__ cmeans, uint16_t* __restrict__ posbin, const uint8_t* __restrict__ lenclass) {
  uint64_t ai = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ai >= na) return;
  const sq_aln a = aln[ai]; const uint32_t rl = ref_len[a.tid];
  PreAln p; p.flags = 0; p.fmt = a.format_id; p.tid = a.tid;
  const double refLength = rl > 0 ? (double)rl : 1.0;
  p.c_cov = a.est_aln_prob > 0 ? sq_log(a.est_aln_prob) : 0.0;
  uint32_t ped = frag_len_pedantic(a, rl);
  uint32_t flen = a.frag_len; if (a.mate_status == SQ_MS_PAIRED_END_PAIRED && a.fwd != a.mate_fwd) flen = ped;
  p.flen = flen; p.fl_ped = (uint16_t)(ped > 1000 ? 1000 : ped);
  if (flen < rl || (rl == 0 && flen < 1)) p.flags |= PF_FLEN_IN_REF;
  const bool isCompat = is_compatible(a.format_id, o.lib_type, o.lib_orientation, o.lib_strand, a.fwd, a.mate_status);
  if (isCompat) p.flags |= PF_COMPAT;
  if (isCompat || !o.ignore_incompat) p.flags |= PF_KEEP;
  if (a.mate_status == SQ_MS_PAIRED_END_PAIRED && !o.no_length_correction) {
    p.flags |= PF_PE_START;
    p.c_start = ((double)flen <= refLength) ? -sq_log(refLength - (double)flen + 1.0) : SQ_LOG_EPSILON;
  }
  else p.c_start = sq_log((double)rl);   // log(RefLength); the mini-batch negates it or uses the cached effective length
  const bool singleEnd = (o.lib_type == 0);
  const bool unexpectedOrphan = (o.lib_type == 1 && a.mate_status != SQ_MS_PAIRED_END_PAIRED);
  if (unexpectedOrphan) p.flags |= PF_UNEXP_ORPHAN;
  p.max_fl = 0; p.tl = 0;
  if (o.model_single_frag_prob && o.use_frag_len_dist && (singleEnd || unexpectedOrphan)) {
    p.flags |= PF_ORPHAN_MODEL;
    int32_t tl = (int32_t)ref_clen[a.tid], maxFL;
    if (a.fwd) { int32_t p1 = a.pos < 0 ? 0 : a.pos; p1 = p1 > tl ? tl : p1; maxFL = tl - p1; }
    else { int32_t p1 = a.pos + (int32_t)a.read_len; p1 = p1 < 0 ? 0 : p1; p1 = p1 > tl ? tl : p1; maxFL = p1; }
    p.max_fl = (uint16_t)(maxFL > 1000 ? 1000 : maxFL); p.tl = (uint16_t)(tl > 1000 ? 1000 : tl);   // tables saturate at 1000
  }
  pre[ai] = p;
  if (gcbin) {   // observedGCMass.inc(transcript.gcDesc(start, stop), aln.logProb) — SalmonQuantify.cpp:938-951 (paired-end observations)
    uint8_t b = 255;
    if ((a.format_id & 1u) == 1u) {
      const int32_t start = a.pos < a.mate_pos ? a.pos : a.mate_pos, stop = start + (int32_t)a.frag_len - 1;
      int32_t ff, cf;
      if (start >= 0 && stop < (int32_t)rl && stop >= start && sq_gc_desc(refseq, gcpre, ref_accum[a.tid], (int32_t)rl, start, stop, &ff, &cf))
        b = (uint8_t)(sq_gc_ctx_bin(cf) * SQ_GC_FRAG_BINS + sq_gc_frag_bin(ff));
    } else if (o.lib_type == 0) {   // :952-971: a single-end library takes every fragment to have the conditional mean length of the prior
      const int32_t cmean = cmeans[rl >= 1001 ? 1000 : rl];
      int32_t start = a.fwd ? a.pos : a.pos - cmean; if (!a.fwd && start < 0) start = 0;
      const int32_t stop = start + cmean; int32_t ff, cf;
      if (start >= 0 && stop < (int32_t)rl && sq_gc_desc(refseq, gcpre, ref_accum[a.tid], (int32_t)rl, start, stop, &ff, &cf))
        b = (uint8_t)(sq_gc_ctx_bin(cf) * SQ_GC_FRAG_BINS + sq_gc_frag_bin(ff));
    }
    gcbin[ai] = b;
  }
  if (posbin) {   // SimplePosBias::addMass(pos, length, mass) (SimplePosBias.cpp:20-28): bin = floor(pos / (length / 20)) of the clamped read start
    const uint32_t li = lenclass[a.tid]; const double step = (double)rl / 20.0;
    auto bin_of = [&](int32_t p) -> uint32_t { if (p < 0) p = 0; if (p >= (int32_t)rl) p = (int32_t)rl - 1; int b = (int)floor((double)p / step); if (b > 19) b = 19; return li * 20u + (uint32_t)b; };
    uint32_t b5 = 255u, b3 = 255u;
    if (a.mate_status == SQ_MS_PAIRED_END_PAIRED) { if (a.fwd != a.mate_fwd) { b5 = bin_of(a.fwd ? a.pos : a.mate_pos); b3 = bin_of(a.fwd ? a.mate_pos : a.pos); } }
    else if (a.fwd) b5 = bin_of(a.pos); else b3 = bin_of(a.pos);
    posbin[ai] = (uint16_t)(b5 | (b3 << 8));
  }
}

// one mini-batch: fragments [r0, r1) of the current mapped batch (thread per fragment)
// first toucher of a transcript in this mini-batch records it (one atomic per wave: the ballot sees the calling lanes only)
__device__ inline void mass_add(const OnlineView& V, uint32_t par, uint32_t w, uint32_t t, unsigned long long q) {
  if (!q) return;
  const unsigned long long old = atomicAdd(&V.mass_acc[(size_t)t * V.W + w], q);   // [r3] the W slots of a transcript share a line: k_apply reads one sector per transcript
  if (old == 0 && atomicExch(&V.tflag[t], 1u) == 0u) {   // first toucher of (mini-batch slot, transcript), and the transcript is not listed yet
    const unsigned long long m = __ballot(1);
    const int leader = __ffsll((long long)m) - 1, lane = (int)(threadIdx.x & 63);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&V.touched_n[par], (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    V.touched[(size_t)par * V.M + base + (uint32_t)__popcll(m & ((1ULL << lane) - 1))] = t;
  }
}

__device__ inline void mini_batch_fragment(const OnlineView& V, const sq_quant_opts& o, uint32_t r, uint32_t r0, uint32_t r1, uint32_t mbs,
    uint64_t read_counter0,
                             const uint64_t* __restrict__ aln_off, const sq_aln* __restrict__ aln, const PreAln* __restrict__ pre,
                                 const uint64_t* __restrict__ assigned_prefix, uint64_t assigned_base,
                             unsigned long long* __restrict__ awq, double* __restrict__ alp, uint32_t* __restrict__ abin,
                                 uint64_t* __restrict__ rh1,
                                 uint64_t* __restrict__ rh2, uint64_t* fmt_out, uint32_t par, int* compat_out, const uint8_t* __restrict__ gcbin) {
  if (r >= r1) return;
  const uint64_t a0 = aln_off[r], a1 = aln_off[r + 1];
  rh1[r] = EQ_EMPTY; rh2[r] = 0;
  if (a1 == a0) return;
  const bool burned = V.ctr[1] != 0; const bool cached = V.ctr[3] != 0;
  const uint64_t assigned_before = assigned_base + assigned_prefix[r];
  const bool useAux = assigned_before >= o.num_pre_burnin_frags;
  const bool cond = burned || useAux;
  const bool singleEnd = (o.lib_type == 0);
  const double totMass = V.scal[0];
  // pass 1: auxProb / logProb per kept alignment and their in-order log-sums
  double auxDenom = SQ_LOG_0, sumProbs = SQ_LOG_0; uint32_t nk = 0; uint64_t fmtSeen = 0; bool hasCompat = false;
  for (uint64_t ai = a0; ai < a1; ++ai) {
    const PreAln p = pre[ai];
    abin[ai] = 0xFFFFFFFFu;
    if (!(p.flags & PF_KEEP)) continue;
    if (p.flags & PF_COMPAT) hasCompat = true;                      // hasCompatibleMapping (SalmonQuantify.cpp:767-769)
    const uint32_t t = aln[ai].tid;
    double logFragProb = 0.0;
    if (p.flags & PF_ORPHAN_MODEL) {
      const bool useFLD = singleEnd || burned;
      const double* tab = useFLD ? (cached ? V.ccmf : V.ambig + 1024) : V.ambig;   // FLD::cmf live (uncached) / LogCMFCache table
      double refCM = tab[p.tl]; bool cm = !(refCM == SQ_LOG_0);
      logFragProb = cm ? (tab[p.max_fl] - refCM) : SQ_LOG_EPSILON;
    } else if (p.flags & PF_UNEXP_ORPHAN) logFragProb = SQ_LOG_EPSILON;
    if (p.flen > 0 && o.use_frag_len_dist && cond) {
      const uint32_t fi = p.flen > 1000 ? 1000 : p.flen;
      const double lenProb = cached ? V.cpmf[fi] : (V.hist[fi] - totMass);
      if (burned) {
        double cm = V.ccmf[fi];
        bool ok = (p.flen < V.ref_len[t] || (V.ref_len[t] == 0 && p.flen < 1)) && !(cm == SQ_LOG_0);
        logFragProb = ok ? (lenProb - cm) : SQ_LOG_EPSILON;
      }
      else if (useAux) logFragProb = lenProb;
    }
    const double logCompat = (p.flags & PF_COMPAT) ? 0.0 : o.incompat_prior;
    double startPosProb;
    if (p.flags & PF_PE_START) startPosProb = p.c_start;
    else {
      double logRefLength = o.no_length_correction ? 1.0 : ((o.no_eff_length_correction || !burned) ? p.c_start : V.log_eff_len[t]);
      startPosProb = -logRefLength;
    }
    fmtSeen |= 1ULL << p.fmt;
    const double auxProb = logFragProb + p.c_cov + logCompat;
    const double logProb = V.tlc[t] + auxProb + startPosProb;
    if (fabs(logProb) == SQ_LOG_0) continue;
    sumProbs = sq_log_add(sumProbs, logProb);
    auxDenom = sq_log_add(auxDenom, auxProb);
    awq[ai] = (unsigned long long)__double_as_longlong(auxProb); alp[ai] = logProb; abin[ai] = 0;
    ++nk;
  }
  if (nk == 0 || sumProbs == SQ_LOG_0) { for (uint64_t ai = a0; ai < a1; ++ai) abin[ai] = 0xFFFFFFFFu; return; }
  *compat_out = hasCompat ? 1 : 0;
  // pass 2: normalise, range-factorization bins, label hash, model increments
  const int32_t rangeCount = (int32_t)(sqrt((double)nk) + (double)o.range_factorization_bins);
  const uint32_t labLen = o.range_factorization_bins > 0 ? 2 * nk : nk;
  uint64_t ha = 0x243F6A8885A308D3ULL ^ (uint64_t)labLen, hb = 0x13198A2E03707344ULL + (uint64_t)labLen;
  for (uint64_t ai = a0; ai < a1; ++ai) if (abin[ai] == 0) label_hash_step(ha, hb, aln[ai].tid);
  const uint64_t readIdx = read_counter0 + (r - r0);
  uint32_t ki = 0; uint32_t firstTid = 0;
  for (uint64_t ai = a0; ai < a1; ++ai) {
    if (abin[ai] != 0) continue;
    const uint32_t t = aln[ai].tid;
    const double auxProb = __longlong_as_double((long long)awq[ai]);
    const double w = sq_exp(auxProb - auxDenom);
    uint32_t bin = 0;
    if (o.range_factorization_bins > 0) bin = (uint32_t)(int32_t)(w * (double)rangeCount);
    awq[ai] = sq_to_fixed(w, SQ_WFRAC_BITS);
    const double pr = sq_exp(alp[ai] - sumProbs);
    mass_add(V, par, mbs, t, (unsigned long long)sq_to_fixed(pr, SQ_MFRAC_BITS));
    atomicAdd(&V.total[t], 1ULL);
    if (gcbin && gcbin[ai] != 255) atomicAdd(&V.gc_obs[gcbin[ai]], (unsigned long long)sq_to_fixed(pr, 32));
    if (V.posbin) pos_observe(V, ai, pr);
    if (!burned) {
      double rr = dev_u01(o.seed, readIdx, ki);
      if (rr < pr) {
        if (V.err_flag) V.err_flag[ai] = (uint8_t)(mbs + 1);   // [r5] the error model learns from this alignment too (k_err_count)
        uint32_t fl = pre[ai].fl_ped;
        if (fl > 0) {
          atomicAdd(&V.fld_cnt[mbs * 1024u + fl], 1u);
          if ((unsigned long long)fl < __hip_atomic_load(&V.ctr[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&V.ctr[2],
              (unsigned long long)fl);
        }
      }
    }
    abin[ai] = bin;   // kept alignments now carry their bin id (< 0xFFFFFFFF)
    if (ki == 0) firstTid = t;
    ++ki;
  }
  if (o.range_factorization_bins > 0) for (uint64_t ai = a0; ai < a1; ++ai) if (abin[ai] != 0xFFFFFFFFu) label_hash_step(ha, hb, abin[ai]);
  uint64_t h1 = sq_mix64(ha), h2 = sq_mix64(hb);
  if (h1 == EQ_EMPTY) h1 = EQ_EMPTY - 1; if (h2 == 0) h2 = 1;
  rh1[r] = h1; rh2[r] = h2;
  if (nk == 1) atomicAdd(&V.uniq[firstTid], 1ULL);
  *fmt_out = fmtSeen;
}


// 8 lanes per fragment, two alignment slots per lane (alignments j and j+8): each lane evaluates its
// alignments (table lookups only), the in-order log-sum chains are replayed by all lanes of the group
// from shuffled values (SIMT-free), then every lane finishes its own alignments (two exps, fixed-point
// increments).  Fragments with more than 16 alignments take the sequential path on lane 0.  Same
// arithmetic, same order as the checker.  (8 rather than 16 lanes: half the workgroups per mini-batch,
// so the kernel fits the eq stage's CU partition in one round.)
#define AP_TB_ 256
#define SQ_MAX_INFLIGHT 64
struct FmArr { double v[SQ_MAX_INFLIGHT]; };   // forgetting masses of the group's mini-batches, in order
#define MB_G 8
#define MB_S 2
__global__ void k_mini_batch(OnlineView V, sq_quant_opts o, uint32_t r0, uint32_t r1, uint32_t mb, uint64_t read_counter0,
                             const uint64_t* __restrict__ aln_off, const sq_aln* __restrict__ aln, const PreAln* __restrict__ pre,
                                 const uint64_t* __restrict__ assigned_prefix, uint64_t assigned_base,
                             unsigned long long* __restrict__ awq, double* __restrict__ alp, uint32_t* __restrict__ abin,
                                 uint64_t* __restrict__ rh1,
                                 uint64_t* __restrict__ rh2, uint32_t par, const uint8_t* __restrict__ gcbin) {
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t r = r0 + gtid / MB_G; const uint32_t j = threadIdx.x & (MB_G - 1);
  uint64_t fmtSeen = 0; int compatFrag = 0;   // this lane reports an assigned fragment that has a compatible alignment
  const bool valid = r < r1;
  const uint32_t mbs = valid ? (r - r0) / mb : 0;   // mini-batch slot inside the group: increments are kept apart per mini-batch (each has its own forgetting mass)
  const uint64_t a0 = valid ? aln_off[r] : 0, a1 = valid ? aln_off[r + 1] : 0;
  const uint32_t nA = (uint32_t)(a1 - a0);
  if (valid && nA > MB_G * MB_S) {
    if (j == 0) mini_batch_fragment(V, o, r, r0, r1, mbs, read_counter0, aln_off, aln, pre, assigned_prefix, assigned_base, awq, alp, abin, rh1,
        rh2, &fmtSeen, par,
        &compatFrag, gcbin);
  } else if (valid) {
    if (j == 0) { rh1[r] = EQ_EMPTY; rh2[r] = 0; }
    if (nA > 0) {
      const bool burned = V.ctr[1] != 0; const bool cached = V.ctr[3] != 0;
      const bool useAux = (assigned_base + assigned_prefix[r]) >= o.num_pre_burnin_frags;
      const bool cond = burned || useAux; const bool singleEnd = (o.lib_type == 0);
      const double totMass = V.scal[0];
      // phase 1: lane j -> alignments a0 + j and a0 + j + 8
      bool keep[MB_S];
      double auxProb[MB_S], logProb[MB_S];
      uint32_t t[MB_S];
      uint32_t fl_ped[MB_S];
      uint64_t fmtBit[MB_S];
      bool compat[MB_S];
#pragma unroll
      for (int sl = 0; sl < MB_S; ++sl) {
        keep[sl] = false; auxProb[sl] = 0.0; logProb[sl] = 0.0; t[sl] = 0; fl_ped[sl] = 0; fmtBit[sl] = 0; compat[sl] = false;
        const uint32_t idx = j + MB_G * sl;
        if (idx < nA) {
          const uint64_t ai = a0 + idx;
          const PreAln p = pre[ai]; fl_ped[sl] = p.fl_ped;
          if (p.flags & PF_KEEP) {
            compat[sl] = (p.flags & PF_COMPAT) != 0;
            const uint32_t tt = p.tid; t[sl] = tt;
            double logFragProb = 0.0;
            if (p.flags & PF_ORPHAN_MODEL) {
              // FLD::cmf live (uncached) / LogCMFCache table
              const bool useFLD = singleEnd || burned;
              const double* tab = useFLD ? (cached ? V.ccmf : V.ambig + 1024) : V.ambig;
              double refCM = tab[p.tl]; bool cm = !(refCM == SQ_LOG_0);
              logFragProb = cm ? (tab[p.max_fl] - refCM) : SQ_LOG_EPSILON;
            } else if (p.flags & PF_UNEXP_ORPHAN) logFragProb = SQ_LOG_EPSILON;
            if (p.flen > 0 && o.use_frag_len_dist && cond) {
              const uint32_t fi = p.flen > 1000 ? 1000 : p.flen;
              const double lenProb = cached ? V.cpmf[fi] : (V.hist[fi] - totMass);
              if (burned) {
                double cm = V.ccmf[fi];
                bool ok = (p.flags & PF_FLEN_IN_REF) && !(cm == SQ_LOG_0);
                logFragProb = ok ? (lenProb - cm) : SQ_LOG_EPSILON;
              }
              else if (useAux) logFragProb = lenProb;
            }
            const double logCompat = (p.flags & PF_COMPAT) ? 0.0 : o.incompat_prior;
            double startPosProb;
            if (p.flags & PF_PE_START) startPosProb = p.c_start;
            else {
              double logRefLength = o.no_length_correction ? 1.0 : ((o.no_eff_length_correction || !burned) ? p.c_start : V.log_eff_len[tt]);
              startPosProb = -logRefLength;
            }
            fmtBit[sl] = 1ULL << p.fmt;
            auxProb[sl] = logFragProb + p.c_cov + logCompat;
            logProb[sl] = V.tlc[tt] + auxProb[sl] + startPosProb;
            keep[sl] = !(fabs(logProb[sl]) == SQ_LOG_0);
          }
        }
      }
      // chains, replayed identically by the lanes of the group (alignment i lives in lane i & 7, slot i >> 3)
      // [r2] the two log-sum chains are independent: the even lanes of the group run sumProbs, the odd lanes auxDenom (one sq_log_add
      // per step in every lane instead of two), each chain in its sequential order
      const bool odd = (j & 1) != 0;
      double chain = SQ_LOG_0; uint32_t nk = 0; uint64_t fmtAll = 0;
      for (uint32_t i = 0; i < nA; ++i) {
        const int src = (int)(i & (MB_G - 1)); const bool hi = i >= MB_G;
        const int kp = __shfl((int)(hi ? keep[1] : keep[0]), src, MB_G);
        const double xa = __shfl(hi ? auxProb[1] : auxProb[0], src, MB_G);
        const double xl = __shfl(hi ? logProb[1] : logProb[0], src, MB_G);
        const unsigned long long fm = __shfl((unsigned long long)(hi ? fmtBit[1] : fmtBit[0]), src, MB_G);
        fmtAll |= fm;
        if (kp) { chain = sq_log_add(chain, odd ? xa : xl); ++nk; }
      }
      const double sumProbs = __shfl(chain, 0, MB_G), auxDenom = __shfl(chain, 1, MB_G);
      const bool assigned = !(nk == 0 || sumProbs == SQ_LOG_0);
      // kept-index of this lane's alignments (ki of the sequential form); ballots in group-uniform code
      const unsigned long long kb0 = __ballot(keep[0]), kb1 = __ballot(keep[1]);
      const int gsh = (int)((threadIdx.x & 63) & ~(MB_G - 1));
      const uint32_t g0 = (uint32_t)((kb0 >> gsh) & ((1u << MB_G) - 1)), g1 = (uint32_t)((kb1 >> gsh) & ((1u << MB_G) - 1));
      const uint32_t kis[MB_S] = {(uint32_t)__popc(g0 & ((1u << j) - 1)), (uint32_t)(__popc(g0) + __popc(g1 & ((1u << j) - 1)))};
      // phase 2
      uint32_t bin[MB_S] = {0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
      for (int sl = 0; sl < MB_S; ++sl) {
        const uint32_t idx = j + MB_G * sl;
        if (idx < nA) {
          const uint64_t ai = a0 + idx;
          if (assigned && keep[sl]) {
            const int32_t rangeCount = (int32_t)(sqrt((double)nk) + (double)o.range_factorization_bins);
            const double w = sq_exp(auxProb[sl] - auxDenom);
            bin[sl] = (o.range_factorization_bins > 0) ? (uint32_t)(int32_t)(w * (double)rangeCount) : 0u;
            awq[ai] = sq_to_fixed(w, SQ_WFRAC_BITS);
            const double pr = sq_exp(logProb[sl] - sumProbs);
            mass_add(V, par, mbs, t[sl], (unsigned long long)sq_to_fixed(pr, SQ_MFRAC_BITS));
            atomicAdd(&V.total[t[sl]], 1ULL);
            if (gcbin && gcbin[ai] != 255) atomicAdd(&V.gc_obs[gcbin[ai]], (unsigned long long)sq_to_fixed(pr, 32));
            if (V.posbin) pos_observe(V, ai, pr);
            if (!burned) {
              double rr = dev_u01(o.seed, read_counter0 + (r - r0), kis[sl]);
              if (rr < pr && V.err_flag) V.err_flag[ai] = (uint8_t)(mbs + 1);
              if (rr < pr && fl_ped[sl] > 0) {
                atomicAdd(&V.fld_cnt[mbs * 1024u + fl_ped[sl]], 1u);
                if ((unsigned long long)fl_ped[sl] < __hip_atomic_load(&V.ctr[2], __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&V.ctr[2],
                    (unsigned long long)fl_ped[sl]);
              }
            }
          }
          abin[ai] = bin[sl];
        }
      }
      // label hash (tids of kept alignments, then their bins), replayed by all lanes; lane 0 publishes
      if (assigned) {
        const uint32_t labLen = o.range_factorization_bins > 0 ? 2 * nk : nk;
        // the two halves of the label hash are independent as well: even lanes carry `a`, odd lanes `b` of label_hash_step, with the
        // one sq_mix64 of a step shared through selects
        uint64_t hh = odd ? (0x13198A2E03707344ULL + (uint64_t)labLen) : (0x243F6A8885A308D3ULL ^ (uint64_t)labLen);
        auto half_step = [&](uint32_t x) {
          const uint64_t u = odd ? (hh + (uint64_t)x * 0xD6E8FEB86659FD93ULL) : (hh ^ (uint64_t)x);
          const uint64_t m = sq_mix64(u);
          hh = odd ? (m ^ (hh >> 29)) : (m + 0x9E3779B97F4A7C15ULL);
        };
        uint32_t firstTid = 0;
        bool gotFirst = false;
        for (uint32_t i = 0; i < nA; ++i) {
          const int src = (int)(i & (MB_G - 1)); const bool hi = i >= MB_G;
          const int kp = __shfl((int)(hi ? keep[1] : keep[0]), src, MB_G);
          const uint32_t ti = (uint32_t)__shfl((int)(hi ? t[1] : t[0]), src, MB_G);
          if (kp) { half_step(ti); if (!gotFirst) { firstTid = ti; gotFirst = true; } }
        }
        if (o.range_factorization_bins > 0) for (uint32_t i = 0; i < nA; ++i) {
          const uint32_t bi = (uint32_t)__shfl((int)(i >= MB_G ? bin[1] : bin[0]), (int)(i & (MB_G - 1)), MB_G);
          if (bi != 0xFFFFFFFFu) half_step(bi);
        }
        const uint64_t ha = (uint64_t)__shfl((unsigned long long)hh, 0, MB_G), hb = (uint64_t)__shfl((unsigned long long)hh, 1, MB_G);
        if (j == 0) {
          uint64_t h1 = sq_mix64(ha), h2 = sq_mix64(hb);
          if (h1 == EQ_EMPTY) h1 = EQ_EMPTY - 1; if (h2 == 0) h2 = 1;
          rh1[r] = h1; rh2[r] = h2;
          if (nk == 1) atomicAdd(&V.uniq[firstTid], 1ULL);
        }
      }
      fmtSeen = (j == 0 && assigned) ? fmtAll : 0;
      { const unsigned long long cb = __ballot(compat[0]) | __ballot(compat[1]);      // group-uniform code
        compatFrag = (j == 0 && assigned && ((cb >> gsh) & ((1u << MB_G) - 1))) ? 1 : 0; }
    }
  }
  // library-format counts and numCompatibleFragments (:811-815): summed per block in LDS, one atomic per (block, format) — a launch has
  // 5000 waves and the same-address atomics of one per wave were a large part of its 190 us
  __shared__ unsigned long long s_lib[64]; __shared__ unsigned long long s_cf;
  if (threadIdx.x < 64) s_lib[threadIdx.x] = 0;
  if (threadIdx.x == 64) s_cf = 0;
  __syncthreads();
  uint64_t any = fmtSeen; for (int s = 32; s >= 1; s >>= 1) any |= __shfl_xor(any, s, 64);
  while (any) {
    int f = __ffsll((long long)any) - 1;
    any &= any - 1;
    unsigned long long m = __ballot((fmtSeen >> f) & 1);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_lib[f], (unsigned long long)__popcll(m));
  }
  {
    const unsigned long long m = __ballot(compatFrag);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_cf, (unsigned long long)__popcll(m));
  }
  __syncthreads();
  if (threadIdx.x < 64 && s_lib[threadIdx.x]) atomicAdd(&V.lib_counts[threadIdx.x], s_lib[threadIdx.x]);
  if (threadIdx.x == 64 && s_cf) atomicAdd(&V.ctr[5], s_cf);
}

// ---- [r5] the CIGAR-based alignment error model of alignment-based input (src/alignment/AlignmentModel.cpp; SPEC f4) ------------------------------------
// The checker's err_like_rec / err_update_rec (oracle.cpp), walk for walk: state = refSymbol * 9 + readSymbol, 82 x 82 log transition weights per
// read-position bin and side, foreground minus background; the update walk differs from the likelihood walk where the reference's two functions do.
struct ErrReads { const uint64_t* cig_off; const uint32_t* cig; const uint64_t* seq_off; const uint8_t* seq; const int32_t* pos; const int32_t* score; };
#define ERR_NS 82u
__device__ inline int err_cig_type(uint32_t op) { return op < 9 ? (int)((0x3C1A7u >> (op << 1)) & 3u) : 0; }
__device__ inline void err_cig_states(uint32_t op, uint32_t& refB, uint32_t& readB) {
  switch (op) { case 1: refB = 4; break; case 2: readB = 4; break; case 3: readB = 8; break; case 4: refB = 5; break; case 5: refB = 6; readB = 6; break; case 6: refB = 7; readB = 7; break; default: break; }
}
__device__ inline void err_like_rec(const OnlineView& V, int side, const uint64_t* __restrict__ refseq, uint64_t g0, uint32_t tLen, int32_t pos, const uint32_t* cig, uint32_t ncig,
                                    const uint8_t* seq, int32_t len, double* fg, double* bg) {
  size_t readIdx = 0; long long tIdx = pos;
  if (tIdx < 0) { readIdx = (size_t)(-tIdx); tIdx = 0; }
  size_t uT = (size_t)tIdx;
  if (uT >= tLen) { *fg = SQ_LOG_0; *bg = 0.0; return; }
  if (ncig == 0) { *fg = SQ_LOG_EPSILON; *bg = 0.0; return; }
  if (len <= 0) { *fg = 0.0; *bg = 0.0; return; }
  const double* cell = V.err_cell + (size_t)side * V.err_bins * ERR_NS * ERR_NS; const double* row = V.err_row + (size_t)side * V.err_bins * ERR_NS;
  double ll = 0.0, bl = 0.0; uint32_t bin = 0, prev = ERR_NS - 1; const double invLen = (double)V.err_bins / (double)len;
  for (uint32_t ci = 0; ci < ncig; ++ci) {
    const uint32_t opLen = cig[ci] >> 4, op = cig[ci] & 15u; const int ty = err_cig_type(op);
    uint32_t curRead = (ty & 1) ? (readIdx < (size_t)len ? seq[readIdx] : 0u) : 0u, curRef = (ty & 2) ? (uT < tLen ? sq_fetch_base(refseq, g0 + uT) : 0u) : 0u;
    bool advRead = false, advRef = false;
    for (uint32_t i = 0; i < opLen; ++i) {
      if (advRead) { if (readIdx >= (size_t)len) { *fg = ll; *bg = bl; return; } curRead = seq[readIdx]; bin = (uint32_t)((double)readIdx * invLen); advRead = false; }
      if (advRef) { if (uT >= tLen) { *fg = ll; *bg = bl; return; } curRef = sq_fetch_base(refseq, g0 + uT); advRef = false; }
      err_cig_states(op, curRef, curRead);
      const uint32_t cur = curRef * 9 + curRead;
      ll += cell[((size_t)bin * ERR_NS + prev) * ERR_NS + cur] - row[(size_t)bin * ERR_NS + prev];
      bl += cell[((size_t)bin * ERR_NS + 0) * ERR_NS + 0] - row[(size_t)bin * ERR_NS + 0];
      prev = cur;
      if (ty & 1) { ++readIdx; advRead = true; }
      if (ty & 2) { ++uT; advRef = true; }
    }
  }
  *fg = ll; *bg = bl;
}
// one thread per fragment of [r0, r1): the conditional probability of every alignment under the matrices as they stand (the group's snapshot), written where
// the mini-batch kernels read the alignment's score term (PreAln::c_cov); LOG_1 until the auxiliary models count (useAuxParams)
__global__ void k_err_like(OnlineView V, ErrReads R, uint32_t r0, uint32_t r1, const uint64_t* __restrict__ aln_off, PreAln* __restrict__ pre,
                           const uint64_t* __restrict__ assigned_prefix, unsigned long long assigned_base, int base_from_ctr, uint32_t num_pre_burnin,
                           const uint32_t* __restrict__ ref_len, const uint64_t* __restrict__ ref_accum, const uint64_t* __restrict__ refseq) {
  const uint32_t r = r0 + blockIdx.x * blockDim.x + threadIdx.x; if (r >= r1) return;
  const unsigned long long base = base_from_ctr ? V.ctr[0] : assigned_base;
  const bool useAux = (base + assigned_prefix[r]) >= num_pre_burnin;
  for (uint64_t ai = aln_off[r]; ai < aln_off[r + 1]; ++ai) {
    double ll = 0.0, bg = 0.0;
    if (useAux) {
      const uint32_t t = pre[ai].tid;
      for (int k = 0; k < 2; ++k) {
        const uint64_t j = 2 * ai + (uint64_t)k; const uint32_t ncig = (uint32_t)(R.cig_off[j + 1] - R.cig_off[j]); const int32_t len = (int32_t)(R.seq_off[j + 1] - R.seq_off[j]);
        if (ncig == 0 && len == 0) continue;
        double f, b; err_like_rec(V, k, refseq, ref_accum[t], ref_len[t], R.pos[j], R.cig + R.cig_off[j], ncig, R.seq + R.seq_off[j], len, &f, &b); ll += f; bg += b;
      }
    }
    pre[ai].c_cov = ll - bg;
  }
}
// one thread per fragment of the group: the alignments the mini-batch kernel drew (err_flag = mini-batch slot + 1) add exp(p) to every cell their walk visits
__global__ void k_err_count(OnlineView V, ErrReads R, uint32_t r0, uint32_t r1, const uint64_t* __restrict__ aln_off, const PreAln* __restrict__ pre,
                            const uint32_t* __restrict__ ref_len, const uint64_t* __restrict__ ref_accum, const uint64_t* __restrict__ refseq) {
  const uint32_t r = r0 + blockIdx.x * blockDim.x + threadIdx.x; if (r >= r1) return;
  const size_t ncell = (size_t)V.err_bins * ERR_NS * ERR_NS;
  for (uint64_t ai = aln_off[r]; ai < aln_off[r + 1]; ++ai) {
    const uint32_t fl = V.err_flag[ai]; if (!fl) continue;
    const unsigned long long q = (unsigned long long)sq_to_fixed(sq_exp((double)R.score[ai]), SQ_MFRAC_BITS);
    const uint32_t t = pre[ai].tid; const uint64_t g0 = ref_accum[t]; const uint32_t tLen = ref_len[t];
    for (int k = 0; k < 2; ++k) {
      const uint64_t j = 2 * ai + (uint64_t)k; const uint32_t ncig = (uint32_t)(R.cig_off[j + 1] - R.cig_off[j]); const int32_t len = (int32_t)(R.seq_off[j + 1] - R.seq_off[j]);
      const uint32_t* cig = R.cig + R.cig_off[j]; const uint8_t* seq = R.seq + R.seq_off[j];
      unsigned long long* acc = V.err_acc + ((size_t)(fl - 1) * 2 + (size_t)k) * ncell;
      int32_t readIdx = 0; long long tIdx = R.pos[j];
      if (tIdx < 0) { readIdx = (int32_t)(-tIdx); tIdx = 0; }
      size_t uT = (size_t)tIdx;
      if (uT >= tLen || ncig == 0 || len <= 0) continue;
      bool advRead = false, advRef = false, stop = false; uint32_t bin = 0, prev = ERR_NS - 1; const double invLen = (double)V.err_bins / (double)len;
      for (uint32_t ci = 0; ci < ncig && !stop; ++ci) {
        const uint32_t opLen = cig[ci] >> 4, op = cig[ci] & 15u; const int ty = err_cig_type(op);
        uint32_t curRead = (ty & 1) ? ((readIdx >= 0 && readIdx < len) ? seq[readIdx] : 0u) : 0u, curRef = (ty & 2) ? (uT < tLen ? sq_fetch_base(refseq, g0 + uT) : 0u) : 0u;
        advRef = false;
        for (uint32_t i = 0; i < opLen; ++i) {
          if (advRead) { if (readIdx >= len) { stop = true; break; } curRead = seq[readIdx]; bin = (uint32_t)((double)readIdx * invLen); advRead = false; }
          if (advRef) { if (uT >= tLen) { stop = true; break; } curRef = sq_fetch_base(refseq, g0 + uT); advRef = false; }
          err_cig_states(op, curRef, curRead);
          const uint32_t cur = curRef * 9 + curRead;
          (void)__hip_atomic_fetch_add(&acc[((size_t)bin * ERR_NS + prev) * ERR_NS + cur], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          prev = cur;
          if (ty & 1) { ++readIdx; advRead = true; }
          if (ty & 2) { ++uT; advRef = true; }
        }
      }
    }
  }
}
// a block per (side, bin, row): thread `cur` folds its cell's increments in, mini-batch after mini-batch, each with its forgetting mass (AtomicMatrix::increment);
// the row's sums (exact integer sums of its cells' increments) go into the row sum the same way
__global__ void __launch_bounds__(128) k_err_apply(OnlineView V, FmArr FM, uint32_t nw) {
  __shared__ unsigned long long s_row[SQ_MAX_INFLIGHT];
  const size_t ncell = (size_t)V.err_bins * ERR_NS * ERR_NS; const uint32_t side = blockIdx.x / (V.err_bins * ERR_NS), br = blockIdx.x % (V.err_bins * ERR_NS), cur = threadIdx.x;
  if (threadIdx.x < SQ_MAX_INFLIGHT) s_row[threadIdx.x] = 0;
  __syncthreads();
  if (cur < ERR_NS) {
    const size_t cellI = (size_t)br * ERR_NS + cur; double v = V.err_cell[(size_t)side * ncell + cellI]; bool any = false;
    for (uint32_t w = 0; w < nw; ++w) {
      unsigned long long* a = V.err_acc + ((size_t)w * 2 + side) * ncell + cellI; const unsigned long long q = *a;
      if (!q) continue;
      v = sq_log_add(v, FM.v[w] + sq_log(sq_from_fixed(q, SQ_MFRAC_BITS))); *a = 0; any = true; atomicAdd(&s_row[w], q);
    }
    if (any) V.err_cell[(size_t)side * ncell + cellI] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double rv = V.err_row[(size_t)side * V.err_bins * ERR_NS + br]; bool any = false;
    for (uint32_t w = 0; w < nw; ++w) if (s_row[w]) { rv = sq_log_add(rv, FM.v[w] + sq_log(sq_from_fixed(s_row[w], SQ_MFRAC_BITS))); any = true; }
    if (any) V.err_row[(size_t)side * V.err_bins * ERR_NS + br] = rv;
  }
}

// ---- --seqBias: observed read-start context models (SalmonQuantify.cpp:1668-1747; SPEC §B2) ----------------------------------------
// One alignment of every paired-end fragment is drawn; a qualifying draw yields two 9-base contexts.  Pass 1 decides per fragment and
// keeps the two context codes, an exclusive scan ranks the successes in read order, pass 2 counts the first num_bias_samples of them.
__device__ inline uint32_t sbo_ctx(const uint64_t* refseq, uint64_t g, int32_t p) { const uint64_t w = sq_fetch_bases(refseq, g + (uint64_t)p, 9); uint32_t v = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) v = (v << 2) | (uint32_t)((w >> (2 * i)) & 3u); return v; }
__device__ inline uint32_t sbo_rc(uint32_t v) { uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) { r = (r << 2) | (3u - (v & 3u)); v >>= 2; } return r; }
__device__ inline uint32_t sbo_cell(uint32_t v, int i) { const int order = i == 0 ? 0 : (i == 1 ? 1 : 2); return (uint32_t)i * 64u + ((v >> (18 - 2 * (i + 1))) & ((1u << (2 * (order + 1))) - 1u)); }
__global__ void k_seq_pick(uint32_t n, int paired_lib, uint64_t read0, uint64_t seed, const uint64_t* __restrict__ aln_off, const sq_aln* __restrict__ aln, const uint32_t* __restrict__ ref_len,
                           const uint64_t* __restrict__ refseq, const uint64_t* __restrict__ ref_accum, uint32_t* __restrict__ flag, uint64_t* __restrict__ code) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  uint32_t ok = 0; uint64_t cd = 0;
  if (r < n) {
    const uint64_t a0 = aln_off[r], a1 = aln_off[r + 1]; const uint32_t na = (uint32_t)(a1 - a0);
    if (na) {
      const uint64_t x = sq_mix64((seed ^ 0x5EB1A5ULL) ^ sq_mix64((read0 + r) * 0x9E3779B97F4A7C15ULL + 1));
      const uint32_t pick = (uint32_t)sq_mulhi64(x, (uint64_t)na + 1);
      if (pick < na) {
        const sq_aln h = aln[a0 + pick];
        if (!paired_lib) {   // single-end library (SalmonQuantify.cpp:2211-2257): one context
          const int32_t RL = (int32_t)ref_len[h.tid]; const int32_t sp = h.fwd ? h.pos : h.pos + (int32_t)h.read_len; const bool rc = !h.fwd;
          const int32_t b = rc ? 5 : 3, a = rc ? 3 : 5;
          if (sp > 0 && sp < RL && sp >= b && sp + a < RL) {
            uint32_t cx = sbo_ctx(refseq, ref_accum[h.tid], sp - b); if (rc) cx = sbo_rc(cx);
            ok = 1; cd = (uint64_t)cx | ((uint64_t)(h.fwd ? 1 : 0) << 36) | (1ULL << 37);
          }
        } else
        if (h.mate_status == SQ_MS_PAIRED_END_PAIRED && h.fwd != h.mate_fwd) {
          const int32_t RL = (int32_t)ref_len[h.tid];
          const int32_t s1 = h.fwd ? h.pos : h.pos + (int32_t)h.read_len - 1, s2 = h.mate_fwd ? h.mate_pos : h.mate_pos + (int32_t)h.mate_len - 1;
          const bool rc1 = !h.fwd, rc2 = !h.mate_fwd;
          const int32_t b1 = rc1 ? 5 : 3, c1 = rc1 ? 3 : 5, b2 = rc2 ? 5 : 3, c2 = rc2 ? 3 : 5;
          const int32_t fwPos = h.fwd ? s1 : s2, rcPos = h.fwd ? s2 : s1;
          if (s1 > 0 && s1 < RL && s2 > 0 && s2 < RL && s1 >= b1 && s1 + c1 < RL && s2 >= b2 && s2 + c2 < RL && fwPos < rcPos) {
            const uint64_t g = ref_accum[h.tid];
            uint32_t left = sbo_ctx(refseq, g, s1 - b1), right = sbo_ctx(refseq, g, s2 - b2);
            if (rc1) left = sbo_rc(left); else right = sbo_rc(right);
            ok = 1; cd = (uint64_t)left | ((uint64_t)right << 18) | ((uint64_t)(h.fwd ? 1 : 0) << 36);
          }
        }
      }
    }
  }
  flag[r] = ok; if (r < n) code[r] = cd;
}
__global__ void k_seq_count(uint32_t n, const uint32_t* __restrict__ flag, const uint64_t* __restrict__ pref, const uint64_t* __restrict__ code, uint64_t cap,
                            unsigned long long* __restrict__ obs /* [1152] counts + [1152] sampled so far */) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || !flag[r]) return;
  const uint64_t before = obs[1152];                      // successes of earlier batches (updated by k_seq_close after this launch)
  if (before + pref[r] >= cap) return;
  const uint64_t cd = code[r]; const uint32_t left = (uint32_t)(cd & 0x3FFFFu), right = (uint32_t)((cd >> 18) & 0x3FFFFu); const bool fwd = (cd >> 36) & 1, single = (cd >> 37) & 1;
#pragma unroll
  for (int i = 0; i < 9; ++i) { atomicAdd(&obs[(fwd ? 0 : 576) + sbo_cell(left, i)], 1ULL); if (!single) atomicAdd(&obs[(fwd ? 576 : 0) + sbo_cell(right, i)], 1ULL); }
}
__global__ void k_seq_close(uint32_t n, const uint64_t* __restrict__ pref, uint64_t cap, unsigned long long* __restrict__ obs) {
  if (blockIdx.x || threadIdx.x) return;
  unsigned long long v = obs[1152] + pref[n]; obs[1152] = v > cap ? cap : v;
}

// ---- [r3] after burn-in: the model-independent half of a mini-batch, once per mapped batch -----------------------------------------
// Once the fragment-length tables are cached and the effective lengths fixed (burn-in, SalmonQuantify.cpp:1012-1018), everything a
// fragment contributes except its transcript-mass terms no longer depends on the evolving model: auxProb, the range-factorization
// weights and bins, the label and its hash, the unique / total counts, the library-format counts.  k_frag_static computes all of
// that for the WHOLE mapped batch in one launch (no mini-batch chain), leaving per kept alignment the two addends the chain still
// needs: logProb = (transcriptLogCount + auxProb) + startPosProb, in that order.  Same arithmetic, same order as k_mini_batch.
struct DynAln { double aux, start; uint32_t tid, keep; };   // 24 B
// [r4] per batch: flag[group][stride] bytes "a kept alignment of this group names this transcript" (plain byte stores, nothing waits on them);
// stride = M rounded up to AP_TB_ * 8 so that k_apply_flagged's blocks read whole 8-byte words; gsize = fragments per group (mini-batch size x W)
struct TouchArgs { uint8_t* flag; uint32_t gsize, stride; };
typedef unsigned long long sqk_u64x2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_frag_static(OnlineView V, sq_quant_opts o, uint32_t n, const uint64_t* __restrict__ aln_off,
    const PreAln* __restrict__ pre, unsigned long long* __restrict__ awq, uint32_t* __restrict__ abin, uint64_t* __restrict__ rh1, uint64_t* __restrict__ rh2,
    DynAln* __restrict__ dyn, TouchArgs TA) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t fmtSeen = 0; int compatFrag = 0;
  if (r < n) {
    const uint64_t a0 = aln_off[r], a1 = aln_off[r + 1];
    uint64_t h1o = EQ_EMPTY, h2o = 0;
    if (a1 > a0) {
      const bool singleEnd = (o.lib_type == 0);
      double auxDenom = SQ_LOG_0; uint32_t nk = 0; bool hasCompat = false; uint64_t fm = 0;
      for (uint64_t ai = a0; ai < a1; ++ai) {
        const PreAln p = pre[ai];
        DynAln d; d.aux = 0.0; d.start = 0.0; d.tid = p.tid; d.keep = 0;
        uint32_t bn = 0xFFFFFFFFu;
        if (p.flags & PF_KEEP) {
          if (p.flags & PF_COMPAT) hasCompat = true;
          double logFragProb = 0.0;
          if (p.flags & PF_ORPHAN_MODEL) {
            const double* tab = V.ccmf; (void)singleEnd;          // burned and cached: FLD::cmf from the cached table either way
            const double refCM = tab[p.tl]; const bool cm = !(refCM == SQ_LOG_0);
            logFragProb = cm ? (tab[p.max_fl] - refCM) : SQ_LOG_EPSILON;
          } else if (p.flags & PF_UNEXP_ORPHAN) logFragProb = SQ_LOG_EPSILON;
          if (p.flen > 0 && o.use_frag_len_dist) {
            const uint32_t fi = p.flen > 1000 ? 1000 : p.flen;
            const double lenProb = V.cpmf[fi], cm = V.ccmf[fi];
            const bool ok = (p.flags & PF_FLEN_IN_REF) && !(cm == SQ_LOG_0);
            logFragProb = ok ? (lenProb - cm) : SQ_LOG_EPSILON;
          }
          const double logCompat = (p.flags & PF_COMPAT) ? 0.0 : o.incompat_prior;
          double startPosProb;
          if (p.flags & PF_PE_START) startPosProb = p.c_start;
          else { const double logRefLength = o.no_length_correction ? 1.0 : (o.no_eff_length_correction ? p.c_start : V.log_eff_len[p.tid]); startPosProb = -logRefLength; }
          const double auxProb = logFragProb + p.c_cov + logCompat;
          // transcriptLogCount is finite for every transcript that can be aligned to (its prior is log(0.005 len)), so the reference's
          // `logProb == LOG_0` test reduces to the finiteness of the two model-independent addends
          fm |= 1ULL << p.fmt;
          if (!(fabs(auxProb + startPosProb) == SQ_LOG_0)) {
            auxDenom = sq_log_add(auxDenom, auxProb);
            d.aux = auxProb; d.start = startPosProb; d.keep = 1; bn = 0; ++nk;
          }
        }
        dyn[ai] = d; abin[ai] = bn;
        if (TA.flag && d.keep) TA.flag[(size_t)(r / TA.gsize) * TA.stride + p.tid] = 1;   // [r4] k_apply_flagged visits the flagged transcripts only
      }
      if (nk > 0) {
        const int32_t rangeCount = (int32_t)(sqrt((double)nk) + (double)o.range_factorization_bins);
        const uint32_t labLen = o.range_factorization_bins > 0 ? 2 * nk : nk;
        uint64_t ha = 0x243F6A8885A308D3ULL ^ (uint64_t)labLen, hb = 0x13198A2E03707344ULL + (uint64_t)labLen;
        uint32_t firstTid = 0; bool got = false;
        for (uint64_t ai = a0; ai < a1; ++ai) if (abin[ai] == 0) { const uint32_t t = dyn[ai].tid; label_hash_step(ha, hb, t); if (!got) { firstTid = t; got = true; } }
        for (uint64_t ai = a0; ai < a1; ++ai) {
          if (abin[ai] != 0) continue;
          const double w = sq_exp(dyn[ai].aux - auxDenom);
          const uint32_t bin = (o.range_factorization_bins > 0) ? (uint32_t)(int32_t)(w * (double)rangeCount) : 0u;
          awq[ai] = sq_to_fixed(w, SQ_WFRAC_BITS);
          atomicAdd(&V.total[dyn[ai].tid], 1ULL);
          abin[ai] = bin;
        }
        if (o.range_factorization_bins > 0) for (uint64_t ai = a0; ai < a1; ++ai) if (abin[ai] != 0xFFFFFFFFu) label_hash_step(ha, hb, abin[ai]);
        uint64_t h1 = sq_mix64(ha), h2 = sq_mix64(hb);
        if (h1 == EQ_EMPTY) h1 = EQ_EMPTY - 1; if (h2 == 0) h2 = 1;
        h1o = h1; h2o = h2;
        if (nk == 1) atomicAdd(&V.uniq[firstTid], 1ULL);
        fmtSeen = fm; compatFrag = hasCompat ? 1 : 0;
      }
    }
    rh1[r] = h1o; rh2[r] = h2o;
  }
  __shared__ unsigned long long s_lib[64]; __shared__ unsigned long long s_cf;
  if (threadIdx.x < 64) s_lib[threadIdx.x] = 0;
  if (threadIdx.x == 64) s_cf = 0;
  __syncthreads();
  uint64_t any = fmtSeen; for (int s = 32; s >= 1; s >>= 1) any |= __shfl_xor(any, s, 64);
  while (any) {
    const int f = __ffsll((long long)any) - 1; any &= any - 1;
    const unsigned long long m = __ballot((fmtSeen >> f) & 1);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_lib[f], (unsigned long long)__popcll(m));
  }
  { const unsigned long long m = __ballot(compatFrag); if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_cf, (unsigned long long)__popcll(m)); }
  __syncthreads();
  if (threadIdx.x < 64 && s_lib[threadIdx.x]) atomicAdd(&V.lib_counts[threadIdx.x], s_lib[threadIdx.x]);
  if (threadIdx.x == 64 && s_cf) atomicAdd(&V.ctr[5], s_cf);
}

// the model-dependent half, one launch per group of W mini-batches (fragments [r0, r1)): logProb from the current transcript masses, the
// in-order log-sum, and the fixed-point mass increments (+ the observed GC model, which is weighted by the same probabilities).
// The increments leave as fire-and-forget atomics (nothing waits for their return); the transcripts a group touched were listed by
// k_frag_static (TouchArgs).
__global__ void __launch_bounds__(256) k_frag_dynamic(OnlineView V, uint32_t r0, uint32_t r1, uint32_t mb, const uint64_t* __restrict__ aln_off,
    const DynAln* __restrict__ dyn, const uint8_t* __restrict__ gcbin) {
  const uint32_t r = r0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= r1) return;
  const uint64_t a0 = aln_off[r], a1 = aln_off[r + 1];
  if (a1 == a0) return;
  const uint32_t mbs = (r - r0) / mb;
  // [r6] a fragment has 2-3 alignments: the records of the first four are requested together, then their transcripts' log-counts together — three dependent trips
  // to memory (offsets, records, log-counts) where the loop took two per alignment.  Same sums in the same order.
  constexpr int NR = 4;
  const uint32_t na = (uint32_t)((a1 - a0) < (uint64_t)NR ? (a1 - a0) : (uint64_t)NR);
  DynAln d[NR]; double lp[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) d[i] = dyn[a0 + ((uint32_t)i < na ? (uint32_t)i : na - 1)];   // no branch around a load: a fragment with fewer reads its last record again
#pragma unroll
  for (int i = 0; i < NR; ++i) lp[i] = V.tlc[d[i].tid];                                        // (every record names a transcript, kept or not)
#pragma unroll
  for (int i = 0; i < NR; ++i) if ((uint32_t)i >= na) d[i].keep = 0;
  double sumProbs = SQ_LOG_0; uint32_t nk = 0;
#pragma unroll
  for (int i = 0; i < NR; ++i) if (d[i].keep) { lp[i] = lp[i] + d[i].aux + d[i].start; sumProbs = sq_log_add(sumProbs, lp[i]); ++nk; }
  for (uint64_t ai = a0 + NR; ai < a1; ++ai) {
    const DynAln x = dyn[ai];
    if (!x.keep) continue;
    sumProbs = sq_log_add(sumProbs, V.tlc[x.tid] + x.aux + x.start); ++nk;
  }
  if (nk == 0) return;
  auto add = [&](uint64_t ai, uint32_t tid, double logProb) {
    const double pr = sq_exp(logProb - sumProbs);
    const unsigned long long q = (unsigned long long)sq_to_fixed(pr, SQ_MFRAC_BITS);
    if (q) (void)__hip_atomic_fetch_add(&V.mass_acc[(size_t)tid * V.W + mbs], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gcbin && gcbin[ai] != 255) (void)__hip_atomic_fetch_add(&V.gc_obs[gcbin[ai]], (unsigned long long)sq_to_fixed(pr, 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (V.posbin) pos_observe(V, ai, pr);
  };
#pragma unroll
  for (int i = 0; i < NR; ++i) if (d[i].keep) add(a0 + i, d[i].tid, lp[i]);
  for (uint64_t ai = a0 + NR; ai < a1; ++ai) {
    const DynAln x = dyn[ai];
    if (!x.keep) continue;
    add(ai, x.tid, V.tlc[x.tid] + x.aux + x.start);
  }
}
// group end after burn-in: one thread per transcript reads its W mass slots (one 64-byte line at W = 8) and, where the group left
// something, folds the mini-batches' increments in order, each with its own forgetting mass (as apply_mass_part); thread 0 keeps
// the running count of assigned fragments
__device__ __forceinline__ void apply_one(const OnlineView& V, const FmArr& FM, uint32_t nw, uint32_t t) {
  unsigned long long* acc = V.mass_acc + (size_t)t * V.W;
  if (V.W == 8) {   // the default: the eight slots as four 16-byte loads, held in registers
    unsigned long long q[8]; unsigned long long any = 0;
    const sqk_u64x2* a2 = (const sqk_u64x2*)acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const sqk_u64x2 v = a2[i]; q[2 * i] = v.x; q[2 * i + 1] = v.y; any |= v.x | v.y; }
    double m = V.mass[t]; const double pm = V.prior_mass[t];   // [r6] requested with the slots, not behind them: a flagged transcript nearly always has something to fold in
    if (!any) return;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      if ((uint32_t)w < nw && q[w]) { m = sq_log_add(m, FM.v[w] + sq_log(sq_from_fixed(q[w], SQ_MFRAC_BITS))); acc[w] = 0; }
    }
    V.mass[t] = m;
    V.tlc[t] = sq_log_add(pm, m);
    return;
  }
  double m = V.mass[t]; bool any = false;
  for (uint32_t w = 0; w < nw; ++w) {
    const unsigned long long q = acc[w];
    if (!q) continue;
    m = sq_log_add(m, FM.v[w] + sq_log(sq_from_fixed(q, SQ_MFRAC_BITS)));
    acc[w] = 0; any = true;
  }
  if (!any) return;
  V.mass[t] = m;
  V.tlc[t] = sq_log_add(V.prior_mass[t], m);
}
// [r4] a group's flagged transcripts: a block reads the flags of AP_TB_ * 4 transcripts (4 per thread, one load), gathers the
// flagged ones in LDS (their order is free: every transcript is its own update) and shares them out evenly.  [r6] Four flags per thread, not eight: a quarter of the
// transcripts are flagged in a group of 40 000 fragments, so with eight a thread had two transcripts to update one after the other — a third dependent trip to memory
// in a kernel that is nothing but trips (flags -> mass slots -> done is two)
__global__ void __launch_bounds__(AP_TB_) k_apply_flagged(OnlineView V, FmArr FM, uint32_t nw, const uint64_t* __restrict__ assigned_prefix, uint32_t r0, uint32_t r1,
    const uint8_t* __restrict__ flag) {
  __shared__ uint32_t s_list[AP_TB_ * 4]; __shared__ uint32_t s_n;
  if (blockIdx.x == 0 && threadIdx.x == 0) V.ctr[0] += (unsigned long long)(assigned_prefix[r1] - assigned_prefix[r0]);
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint32_t t0 = (blockIdx.x * AP_TB_ + threadIdx.x) * 4;
  uint32_t f = ((const uint32_t*)flag)[blockIdx.x * AP_TB_ + threadIdx.x];
  if (f) {
    uint32_t at = atomicAdd(&s_n, (uint32_t)__popc(f));
    while (f) { const int i = (__ffs((int)f) - 1) >> 3; s_list[at++] = t0 + (uint32_t)i; f &= ~(0xffu << (8 * i)); }
  }
  __syncthreads();
  const uint32_t n = s_n;
  for (uint32_t i = threadIdx.x; i < n; i += AP_TB_) apply_one(V, FM, nw, s_list[i]);
}

// batch end, part 1: masses (one thread per transcript); also refreshes the cached
// transcript.mass(withPrior) = logAdd(priorMass, mass) (Transcript.hpp:214-217)
__device__ inline void apply_mass_part(const OnlineView& V, const FmArr& FM, uint32_t nw, uint64_t assigned_after, int set_ctr, uint32_t par,
    uint32_t mass_blocks) {
  // the other list is idle until the next mini-batch
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (set_ctr) V.ctr[0] = assigned_after;
    V.touched_n[par ^ 1] = 0;
  }
  const uint32_t n = V.touched_n[par]; const uint32_t* list = V.touched + (size_t)par * V.M;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += mass_blocks * blockDim.x) {
    const uint32_t t = list[i];
    double m = V.mass[t];
    for (uint32_t w = 0; w < nw; ++w) {                    // the group's mini-batches in order, each with its own forgetting mass
      const unsigned long long q = V.mass_acc[(size_t)t * V.W + w];
      if (!q) continue;
      m = sq_log_add(m, FM.v[w] + sq_log(sq_from_fixed(q, SQ_MFRAC_BITS)));
      V.mass_acc[(size_t)t * V.W + w] = 0;
    }
    V.mass[t] = m;
    V.tlc[t] = sq_log_add(V.prior_mass[t], m);
    V.tflag[t] = 0;
  }
}

// batch end, part 2 (one block of 1024): FLD histogram update + tree total + counters + burn-in trigger
// 256 threads walk the 1024 histogram bins (4 per thread): blocks of 256 threads slot in beside the
// resident waves of the mapping kernels, a 1024-thread block has to wait for a whole CU to drain.
#define AP_TB 256
__device__ inline void apply_fld_part(const OnlineView& V, const FmArr& FM, uint32_t nw, uint64_t assigned_after, uint64_t num_burnin) {
  __shared__ double v[1024]; __shared__ int any;
  const int tid = threadIdx.x;
  if (tid == 0) any = 0;
  __syncthreads();
  const bool burned = V.ctr[1] != 0;
  if (!burned) { int mine = 0; for (uint32_t b = tid; b < nw * 1024u; b += AP_TB) if (V.fld_cnt[b]) mine = 1; if (mine) any = 1; }
  __syncthreads();
  if (!burned && any) {
    const double kern[5] = {sq_log(0.0625), sq_log(0.25), sq_log(0.375), sq_log(0.25), sq_log(0.0625)};
    for (int b = tid; b < 1024; b += AP_TB) {
      double h = (b <= 1000) ? V.hist[b] : SQ_LOG_0;
      if (b >= 1 && b <= 1000) {
        for (uint32_t w = 0; w < nw; ++w) {                // a bin's updates depend on nothing but the counts: mini-batch after mini-batch
          const uint32_t* cnt = V.fld_cnt + w * 1024u;
          for (int i = 4; i >= 0; --i) {
            int len = b + 2 - i;
            if (len < 0 || len > 1000) continue;
            uint32_t c = cnt[len];
            if (!c) continue;
            h = sq_log_add(h, FM.v[w] + kern[i] + sq_log((double)c));
          }
        }
        V.hist[b] = h;
      }
      v[b] = h;
    }
    __syncthreads();
    for (int s = 512; s >= 1; s >>= 1) { for (int b = tid; b < s; b += AP_TB) v[b] = sq_log_add(v[b], v[b + s]); __syncthreads(); }
    if (tid == 0) V.scal[0] = v[0];
    for (uint32_t b = tid; b < nw * 1024u; b += AP_TB) V.fld_cnt[b] = 0;
  }
  if (tid == 0) {
    V.ctr[0] = assigned_after;
    if (assigned_after >= num_burnin && V.ctr[1] == 0 && V.ctr[4] == 0) V.ctr[4] = 1;  // burn-in finalisation pending
  }
}

// ONE launch per mini-batch end: blocks [0, mass_blocks) update the masses of the transcripts the mini-batch
// touched (a list the first toucher appends to — a few thousand entries instead of a sweep over all M, so the
// kernel needs few workgroups and squeezes in beside the mapping kernels), the extra last block (before
// burn-in only) updates the FLD — the two parts touch disjoint state, and every launch saved shortens
// the sequential mini-batch chain (the model of mini-batch i+1 depends on the end of mini-batch i).
__global__ void __launch_bounds__(AP_TB) k_apply(OnlineView V, FmArr FM, uint32_t nw, uint64_t assigned_after, uint64_t num_burnin,
    uint32_t mass_blocks, int with_fld,
    uint32_t par) {
  if (blockIdx.x < mass_blocks) apply_mass_part(V, FM, nw, assigned_after, with_fld ? 0 : 1, par, mass_blocks);
  else apply_fld_part(V, FM, nw, assigned_after, num_burnin);
}

// burn-in finalisation (FLD.cacheCMF :174-186 + updateTranscriptLengthsAtomic ReadExperiment.inl:62-94), one thread: 3 chains of 1001 logAdds, once
__global__ void k_burnin_tables(OnlineView V, int force) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (!(V.ctr[4] == 1 || force)) return;
  const double tot0 = V.scal[0];
  // effective-length correction factors use the *uncached* pmf over [minV, 1000]
  uint32_t minLen = (uint32_t)V.ctr[2]; uint32_t minV = (minLen == 1000) ? 1 : minLen;
  if (V.ctr[3] == 0) {
    double sum = SQ_LOG_0; for (uint32_t i = minV; i <= 1000; ++i) sum = sq_log_add(sum, V.hist[i] - tot0);
    double vals = 0.0, mult = 0.0;  // correctionFactorsFromMass (DistributionUtils.cpp:9-32); pmf[0] = 0 unless minV == 0 (never)
    V.cfac[0] = 0.0;
    for (uint32_t i = 1; i <= 1000; ++i) {
      double p = (i >= minV && i < 1000) ? 100.0 * sq_exp((V.hist[i] - tot0) - sum) : 0.0;
      vals = p * (double)i + vals; mult = p + mult;
      V.cfac[i] = (mult > 0) ? vals / mult : 0.0;
    }
  }
  if (!force) {
    // cacheCMF: normalised pmf then prefix log-sum
    double tot = SQ_LOG_0; for (int i = 0; i <= 1000; ++i) { double p = V.hist[i] - tot0; V.cpmf[i] = p; tot = sq_log_add(tot, p); }
    double cum = SQ_LOG_0;
    for (int i = 0; i <= 1000; ++i) {
      double p = V.cpmf[i] - tot;
      V.cpmf[i] = p;
      cum = sq_log_add(cum, p);
      V.ccmf[i] = cum;
    }
    V.ctr[3] = 1; V.ctr[1] = 1; V.ctr[4] = 2;
  } else V.ctr[4] = 3;
}
__global__ void k_burnin_efflen(OnlineView V, int stage /*2 after burn-in, 3 forced at finish*/) {
  if (V.ctr[4] != (unsigned long long)stage) return;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V.M) return;
  double ol = (double)V.ref_len[t]; uint32_t rl = V.ref_len[t];
  double c = (rl >= 1001) ? V.cfac[1000] : V.cfac[rl];
  double el = ol - c; if (el < 1.0) el = ol;
  V.log_eff_len[t] = sq_log(el);
}
__global__ void k_burnin_done(OnlineView V) { if (threadIdx.x == 0 && blockIdx.x == 0 && (V.ctr[4] == 2 || V.ctr[4] == 3)) V.ctr[4] = 4; }

// ---- eq-class table ---------------------------------------------------------------------------
struct EqView { uint64_t cap; unsigned long long* k1; unsigned long long* k2; unsigned long long* count; unsigned long long* pool; uint32_t* n; uint32_t* pool_tid; uint32_t* pool_bin; unsigned long long* pool_wq; unsigned long long* cursor; uint64_t pool_cap; };

__device__ inline uint64_t eq_find_or_insert(const EqView& T, uint64_t h1, uint64_t h2, bool* is_new) {
  uint64_t slot = h1 & (T.cap - 1); *is_new = false;
  for (uint64_t probes = 0; probes < T.cap; ++probes) {
    unsigned long long cur = __hip_atomic_load(&T.k1[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == EQ_EMPTY) {
      unsigned long long old = atomicCAS(&T.k1[slot], EQ_EMPTY, (unsigned long long)h1);
      if (old == EQ_EMPTY) {
        __hip_atomic_store(&T.k2[slot], (unsigned long long)h2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *is_new = true;
        return slot;
      }
      cur = old;
    }
    if (cur == h1) {
      unsigned long long o2 = __hip_atomic_load(&T.k2[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (o2 == 0) { --probes; continue; }  // inserter has not published h2 yet: re-poll this slot (loop re-converges every iteration)
      if (o2 == h2) return slot;
    }
    slot = (slot + 1) & (T.cap - 1);
  }
  return ~0ULL;
}

__global__ void k_eq_insert(EqView T, uint32_t n, const uint64_t* __restrict__ aln_off, const sq_aln* __restrict__ aln,
    const uint32_t* __restrict__ abin,
                            const uint64_t* __restrict__ rh1, const uint64_t* __restrict__ rh2, uint32_t* __restrict__ rslot,
                                uint32_t bins_on) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  rslot[r] = 0xFFFFFFFFu;
  const uint64_t h1 = rh1[r]; if (h1 == EQ_EMPTY) return;
  bool is_new; uint64_t slot = eq_find_or_insert(T, h1, rh2[r], &is_new);
  if (slot == ~0ULL) { T.cursor[2] = 1; return; }
  rslot[r] = (uint32_t)slot;
  if (is_new) {
    uint32_t nk = 0; for (uint64_t ai = aln_off[r]; ai < aln_off[r + 1]; ++ai) if (abin[ai] != 0xFFFFFFFFu) ++nk;
    unsigned long long off = atomicAdd(&T.cursor[0], (unsigned long long)nk); atomicAdd(&T.cursor[1], 1ULL);
    if (off + nk > T.pool_cap) { T.cursor[2] = 2; T.n[slot] = 0; T.pool[slot] = 0; return; }
    uint32_t i = 0;
    for (uint64_t ai = aln_off[r]; ai < aln_off[r + 1]; ++ai) if (abin[ai] != 0xFFFFFFFFu) {
      T.pool_tid[off + i] = aln[ai].tid;
      T.pool_bin[off + i] = bins_on ? abin[ai] : 0;
      ++i;
    }
    T.n[slot] = nk; T.pool[slot] = off;
  }
}
__global__ void k_eq_add(EqView T, uint32_t n, const uint64_t* __restrict__ aln_off, const uint32_t* __restrict__ abin,
    const unsigned long long* __restrict__ awq,
    const uint32_t* __restrict__ rslot) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  uint32_t slot = rslot[r]; if (slot == 0xFFFFFFFFu) return;
  atomicAdd(&T.count[slot], 1ULL);
  if (T.n[slot] == 0) return;
  unsigned long long off = T.pool[slot]; uint32_t i = 0;
  for (uint64_t ai = aln_off[r]; ai < aln_off[r + 1]; ++ai) if (abin[ai] != 0xFFFFFFFFu) { atomicAdd(&T.pool_wq[off + i], awq[ai]); ++i; }
}
// merge an external table (classes given as CSR) — exact integer adds, any order
__global__ void k_eq_merge_insert(EqView T, uint64_t E, const uint64_t* __restrict__ off, const uint32_t* __restrict__ tid,
    const uint32_t* __restrict__ bins,
    const uint64_t* __restrict__ h1, const uint64_t* __restrict__ h2, uint32_t* __restrict__ cslot) {
  uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= E) return;
  bool is_new; uint64_t slot = eq_find_or_insert(T, h1[c], h2[c], &is_new);
  if (slot == ~0ULL) { T.cursor[2] = 1; cslot[c] = 0xFFFFFFFFu; return; }
  cslot[c] = (uint32_t)slot;
  if (is_new) {
    uint32_t nk = (uint32_t)(off[c + 1] - off[c]);
    unsigned long long po = atomicAdd(&T.cursor[0], (unsigned long long)nk); atomicAdd(&T.cursor[1], 1ULL);
    if (po + nk > T.pool_cap) { T.cursor[2] = 2; T.n[slot] = 0; T.pool[slot] = 0; return; }
    for (uint32_t i = 0; i < nk; ++i) { T.pool_tid[po + i] = tid[off[c] + i]; T.pool_bin[po + i] = bins ? bins[off[c] + i] : 0; }
    T.n[slot] = nk; T.pool[slot] = po;
  }
}
__global__ void k_eq_merge_add(EqView T, uint64_t E, const uint64_t* __restrict__ off, const uint64_t* __restrict__ wq,
    const uint64_t* __restrict__ count,
    const uint32_t* __restrict__ cslot) {
  uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= E) return;
  uint32_t slot = cslot[c]; if (slot == 0xFFFFFFFFu) return;
  atomicAdd(&T.count[slot], (unsigned long long)count[c]);
  if (T.n[slot] == 0) return;
  unsigned long long po = T.pool[slot];
  for (uint64_t i = off[c]; i < off[c + 1]; ++i) atomicAdd(&T.pool_wq[po + (i - off[c])], (unsigned long long)wq[i]);
}

__global__ void k_gather_bounds(const uint64_t* __restrict__ prefix, uint32_t mb, uint32_t n, uint32_t nmb, uint64_t* __restrict__ out) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nmb) return;
  uint64_t r = (uint64_t)b * mb; if (r > n) r = n;
  out[b] = prefix[r];
}

// ---- device-side export of the table in canonical order --------------------------------------
// Canonical class order = ascending (first transcript id, h1, h2).  Leading with the first label
// keeps classes of one gene family adjacent, so the EM's theta[tid] / inv[class] gathers
// (em.hip k_class / k_l1) land in a few cache lines instead of being spread by the hash.
// Sort key = first_tid << 32 | h1 >> 32; a key tie that is out of (h1,h2) order is fixed on the host.
__global__ void k_eq_collect(EqView T, unsigned long long* __restrict__ keys, uint32_t* __restrict__ slots,
    unsigned long long* __restrict__ counter) {
  uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool occ = s < T.cap && T.k1[s] != EQ_EMPTY;
  const unsigned long long m = __ballot(occ);
  unsigned long long base = 0;
  if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(counter, (unsigned long long)__popcll(m));
  base = __shfl(base, 0, 64);
  if (occ) {
    uint64_t i = base + __popcll(m & ((1ULL << (threadIdx.x & 63)) - 1));
    keys[i] = ((unsigned long long)T.pool_tid[T.pool[s]] << 32) | (T.k1[s] >> 32);
    slots[i] = (uint32_t)s;
  }
}
__global__ void k_eq_sizes(EqView T, uint64_t E, const uint32_t* __restrict__ slots, const unsigned long long* __restrict__ keys,
    uint32_t* __restrict__ nlab,
    uint32_t* __restrict__ tie) {
  uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > E) return;
  if (c == E) { nlab[E] = 0; return; }
  nlab[c] = T.n[slots[c]];
  if (c + 1 < E && keys[c] == keys[c + 1]) {   // equal (first tid, h1 hi): order by the full (h1, h2) (fixed up on the host, rare)
    const unsigned long long a1 = T.k1[slots[c]], b1 = T.k1[slots[c + 1]];
    if (a1 > b1 || (a1 == b1 && T.k2[slots[c]] > T.k2[slots[c + 1]])) *tie = 1;
  }
}
__global__ void k_eq_gather(EqView T, uint64_t E, const uint32_t* __restrict__ slots, const uint64_t* __restrict__ off,
    uint32_t* __restrict__ tid,
    double* __restrict__ w, unsigned long long* __restrict__ wq,
                            unsigned long long* __restrict__ count, uint32_t* __restrict__ bins, unsigned long long* __restrict__ h1,
                                unsigned long long* __restrict__ h2) {
  uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= E) return;
  const uint32_t s = slots[c]; const uint32_t n = T.n[s]; const unsigned long long po = T.pool[s]; const uint64_t p = off[c];
  count[c] = T.count[s]; h1[c] = T.k1[s]; h2[c] = T.k2[s];
  double sum = 0.0; for (uint32_t i = 0; i < n; ++i) sum += sq_from_fixed(T.pool_wq[po + i], SQ_WFRAC_BITS);
  const double norm = 1.0 / sum;  // TGValue::normalizeAux (EquivalenceClassBuilder.hpp:116-125)
  for (uint32_t i = 0; i < n; ++i) {
    unsigned long long q = T.pool_wq[po + i];
    tid[p + i] = T.pool_tid[po + i];
    bins[p + i] = T.pool_bin[po + i];
    wq[p + i] = q;
    w[p + i] = sq_from_fixed(q, SQ_WFRAC_BITS) * norm;
  }
}

OnlineView make_view(sq_ctx* c) {
  sq_online_dev* o = c->online; OnlineView V;
  V.M = o->M; V.W = o->inflight;
  V.ref_len = c->di->ref_len;
  V.ref_clen = c->di->ref_clen;
  V.tlc = o->tlc.p;
  V.hist = o->hist.p;
  V.cpmf = o->cpmf.p;
  V.ccmf = o->ccmf.p;
  V.ambig = o->ambig.p;
  V.mass = o->mass.p;
  V.prior_mass = o->prior_mass.p;
  V.log_eff_len = o->log_eff_len.p;
  V.scal = o->scal.p;
  V.cfac = o->cfac.p;
  V.mass_acc = o->mass_acc.p;
  V.uniq = o->uniq.p;
  V.total = o->total.p;
  V.lib_counts = o->lib_counts.p;
  V.fld_cnt = o->fld_cnt.p;
  V.ctr = o->ctr.p;
  V.touched = o->touched.p;
  V.touched_n = o->touched_n.p;
  V.tflag = o->tflag.p;
  V.gc_obs = c->opts.gc_bias ? o->gc_obs.p : nullptr;
  V.pos_obs = c->opts.pos_bias ? o->pos_obs.p : nullptr; V.posbin = c->opts.pos_bias ? o->posbin.p : nullptr;
  V.err_cell = o->err_cell.p; V.err_row = o->err_row.p; V.err_acc = o->err_acc.p; V.err_flag = o->err_bins ? o->err_flag.p : nullptr; V.err_bins = o->err_bins;
  return V;
}
EqView make_eq_view(sq_online_dev* o) {
  EqView T;
  T.cap = o->tcap;
  T.k1 = o->tk1.p;
  T.k2 = o->tk2.p;
  T.count = o->tcount.p;
  T.pool = o->tpool.p;
  T.n = o->tn.p;
  T.pool_tid = o->pool_tid.p;
  T.pool_bin = o->pool_bin.p;
  T.pool_wq = o->pool_wq.p;
  T.cursor = o->pool_cursor.p;
  T.pool_cap = o->pool_cap;
  return T;
}
double phi(double x) { return 0.5 * std::erfc(-x * 0.70710678118654752440); }
}  // namespace

#include <cstring>
#include <rocprim/rocprim.hpp>

int sq_online_create(sq_ctx* c) {
  sq_online_dev* o = new sq_online_dev(); c->online = o;
  const uint32_t M = (uint32_t)c->idx->names.size(); o->M = M;
  const uint32_t W = std::max(1u, std::min<uint32_t>(c->opts.mini_batches_in_flight ? c->opts.mini_batches_in_flight : 1u, SQ_MAX_INFLIGHT)); o->inflight = W;
  // eq table capacity: 2^22 slots per million batch reads, min 2^20, max 2^26
  uint64_t cap = 1ull << 22; o->tcap = cap; o->pool_cap = cap * 4;
  bool bad = o->hist.ensure(1024) || o->cpmf.ensure(1024) || o->ccmf.ensure(1024) || o->ambig.ensure(2048) || o->mass.ensure(M) ||
      o->prior_mass.ensure(M) ||
      o->log_eff_len.ensure(M) || o->tlc.ensure(M) || o->scal.ensure(8) || o->cfac.ensure(1024) ||
             o->touched.ensure((size_t)2 * M) || o->touched_n.ensure(2) || o->tflag.ensure(M) || o->mass_acc.ensure((size_t)W * M) || o->uniq.ensure(M) ||
                 o->total.ensure(M) ||
                 o->lib_counts.ensure(64) || o->gc_obs.ensure(SQ_GC_COND_BINS * SQ_GC_FRAG_BINS + 8) || o->pos_obs.ensure(208) || o->fld_cnt.ensure((size_t)W * 1024) || o->ctr.ensure(8) ||
             o->assigned_flag.ensure((size_t)c->max_reads + 2) || o->assigned_prefix.ensure((size_t)c->max_reads + 2) ||
                 o->rh1.ensure(c->max_reads) ||
                 o->rh2.ensure(c->max_reads) || o->rslot.ensure(c->max_reads) ||
             o->tk1.ensure(cap) || o->tk2.ensure(cap) || o->tcount.ensure(cap) || o->tpool.ensure(cap) || o->tn.ensure(cap) ||
                 o->pool_tid.ensure(o->pool_cap) ||
                 o->pool_bin.ensure(o->pool_cap) || o->pool_wq.ensure(o->pool_cap) || o->pool_cursor.ensure(4) || o->seq_obs.ensure(1160);
  if (bad) { sq_set_error("device allocation failed (online model / eq table)"); return SQ_ERR_NOMEM; }
  const sq_quant_opts& q = c->opts;
  o->detect_active = q.lib_autodetect != 0;
  std::vector<double> hist(1024, SQ_LOG_0), ambig(2048, 0.0), pm(M), le(M), mass(M, SQ_LOG_0); double tot0 = 0.0;
  for (int i = 0; i <= 1000; ++i) {  // FragmentLengthDistribution.cpp:38-55 (alpha = 1)
    double nm = phi((i + 0.5 - q.fld_mean) / q.fld_sd) - phi((i - 0.5 - q.fld_mean) / q.fld_sd);
    hist[i] = (nm != 0) ? sq_log(nm) : SQ_LOG_EPSILON;
  }
  {   // total mass of the prior histogram: the canonical strided-halving sum (SPEC D2)
    std::vector<double> v(1024, SQ_LOG_0);
    for (int i = 0; i <= 1000; ++i) v[i] = hist[i];
    for (int s = 512; s >= 1; s >>= 1)
      for (int i = 0; i < s; ++i) v[i] = sq_log_add(v[i], v[i + s]);
    tot0 = v[0];
    double scal[8] = {v[0], 0, 0, 0, 0, 0, 0, 0};
    SQ_HIP_CHECK(hipMemcpy(o->scal.p, scal, sizeof(scal), hipMemcpyHostToDevice));
  }
  // evaluateLogCMF as written (DistributionUtils.cpp:104-118)
  {
    double cum = SQ_LOG_0;
    for (int j = 0; j <= 1000; ++j) {
      cum = sq_log_add(cum, SQ_LOG_EPSILON);
      ambig[j] = cum;
    }
  }
  // ambig[1024..]: FragmentLengthDistribution::cmf(len) before cacheCMF (:143-158) — the sequential log-sum prefix of the
  // histogram minus its total mass.  Only single-end libraries read it (useFLD before burn-in), and they never add
  // fragment lengths to the histogram, so the prior's table is the live one.
  { double cum = SQ_LOG_0; for (int j = 0; j <= 1000; ++j) { cum = sq_log_add(cum, hist[j]); ambig[1024 + j] = cum - tot0; } }
  // Transcript.hpp:48-56; alpha = 0.005 (ReadExperiment.inl:114)
  for (uint32_t t = 0; t < M; ++t) {
    double len = (double)c->idx->ref_len[t];
    pm[t] = sq_log(0.005 * len);
    le[t] = sq_log(len);
  }
  {   // conditional means of the prior (ReadExperiment.inl:25-43: correctionFactorsFromMass over its normalised pmf x 100, lengths 1..999)
    if (o->cmeans.ensure(1024)) { sq_set_error("device allocation failed (conditional means)"); return SQ_ERR_NOMEM; }
    double sum = SQ_LOG_0; for (int j = 1; j <= 1000; ++j) sum = sq_log_add(sum, hist[j] - tot0);
    std::vector<int32_t> cm(1024, 0); double vals = 0.0, mult = 0.0;
    for (int j = 1; j <= 1000; ++j) { const double p = j < 1000 ? 100.0 * sq_exp((hist[j] - tot0) - sum) : 0.0; vals = p * (double)j + vals; mult = p + mult; cm[j] = (int32_t)(mult > 0 ? vals / mult : 0.0); }
    SQ_HIP_CHECK(hipMemcpy(o->cmeans.p, cm.data(), 1024 * 4, hipMemcpyHostToDevice));
  }
  if (c->opts.pos_bias) {   // Transcript::lengthClassIndex of every reference (host/posbias.cpp)
    std::vector<uint8_t> cls; uint32_t quant[SQ_POS_CLASSES]; sq_pos_length_classes(c->idx, quant, cls);
    if (o->lenclass.ensure(M + 8)) { sq_set_error("device allocation failed (length classes)"); return SQ_ERR_NOMEM; }
    SQ_HIP_CHECK(hipMemcpy(o->lenclass.p, cls.data(), M, hipMemcpyHostToDevice));
  }
  SQ_HIP_CHECK(hipMemcpy(o->hist.p, hist.data(), 1024 * 8, hipMemcpyHostToDevice));
  SQ_HIP_CHECK(hipMemcpy(o->ambig.p, ambig.data(), 2048 * 8, hipMemcpyHostToDevice));
  SQ_HIP_CHECK(hipMemcpy(o->prior_mass.p, pm.data(), (size_t)M * 8, hipMemcpyHostToDevice));
  SQ_HIP_CHECK(hipMemcpy(o->log_eff_len.p, le.data(), (size_t)M * 8, hipMemcpyHostToDevice));
  SQ_HIP_CHECK(hipMemcpy(o->mass.p, mass.data(), (size_t)M * 8, hipMemcpyHostToDevice));
  SQ_HIP_CHECK(hipMemcpy(o->tlc.p, pm.data(), (size_t)M * 8, hipMemcpyHostToDevice));  // logAdd(prior, LOG_0) = prior
  SQ_HIP_CHECK(hipMemset(o->touched_n.p, 0, 8));
  SQ_HIP_CHECK(hipMemset(o->mass_acc.p, 0, (size_t)W * M * 8));
  SQ_HIP_CHECK(hipMemset(o->tflag.p, 0, (size_t)M * 4));
  SQ_HIP_CHECK(hipMemset(o->uniq.p, 0, (size_t)M * 8));
  SQ_HIP_CHECK(hipMemset(o->total.p, 0, (size_t)M * 8));
  SQ_HIP_CHECK(hipMemset(o->lib_counts.p, 0, 64 * 8));
  SQ_HIP_CHECK(hipMemset(o->gc_obs.p, 0, (SQ_GC_COND_BINS * SQ_GC_FRAG_BINS + 8) * 8));
  SQ_HIP_CHECK(hipMemset(o->pos_obs.p, 0, 208 * 8));
  SQ_HIP_CHECK(hipMemset(o->fld_cnt.p, 0, (size_t)W * 1024 * 4));
  if (c->opts.error_model) {   // [r5] AlignmentModel(1.0, numErrorBins): every cell log(alpha) = log(1), every row sum log(82 alpha) (AtomicMatrix.hpp:19-33)
    const uint32_t bins = c->opts.num_error_bins ? c->opts.num_error_bins : 6; const size_t ncell = (size_t)bins * 82 * 82, nrow = (size_t)bins * 82;
    if (o->err_cell.ensure(2 * ncell) || o->err_row.ensure(2 * nrow) || o->err_acc.ensure((size_t)W * 2 * ncell)) { sq_set_error("device allocation failed (error model)"); return SQ_ERR_NOMEM; }
    std::vector<double> c0(2 * ncell, sq_log(1.0)), r0(2 * nrow, sq_log(82.0 * 1.0));
    SQ_HIP_CHECK(hipMemcpy(o->err_cell.p, c0.data(), c0.size() * 8, hipMemcpyHostToDevice)); SQ_HIP_CHECK(hipMemcpy(o->err_row.p, r0.data(), r0.size() * 8, hipMemcpyHostToDevice));
    SQ_HIP_CHECK(hipMemset(o->err_acc.p, 0, (size_t)W * 2 * ncell * 8)); o->err_bins = bins;
  }
  SQ_HIP_CHECK(hipMemset(o->seq_obs.p, 0, 1160 * 8));
  SQ_HIP_CHECK(hipMemset(o->cfac.p, 0, 1024 * 8));
  unsigned long long ctr[8] = {0, 0, 1000, 0, 0, 0, 0, 0}; SQ_HIP_CHECK(hipMemcpy(o->ctr.p, ctr, sizeof(ctr), hipMemcpyHostToDevice));
  SQ_HIP_CHECK(hipMemset(o->tk1.p, 0xFF, cap * 8));
  SQ_HIP_CHECK(hipMemset(o->tk2.p, 0, cap * 8));
  SQ_HIP_CHECK(hipMemset(o->tcount.p, 0, cap * 8));
  SQ_HIP_CHECK(hipMemset(o->tn.p, 0, cap * 4));
  SQ_HIP_CHECK(hipMemset(o->pool_wq.p, 0, o->pool_cap * 8)); SQ_HIP_CHECK(hipMemset(o->pool_cursor.p, 0, 4 * 8));
  return SQ_OK;
}

void sq_online_free(sq_ctx* c) {
  sq_online_dev* o = c->online; if (!o) return;
  o->hist.free_();
  o->cpmf.free_();
  o->ccmf.free_();
  o->ambig.free_();
  o->mass.free_();
  o->prior_mass.free_();
  o->log_eff_len.free_();
  o->tlc.free_();
  o->pre.free_();
  o->alp.free_(); o->dyn.free_(); o->gflag.free_(); o->cmeans.free_(); o->seq_obs.free_(); o->seq_flag.free_(); o->seq_pref.free_(); o->seq_code.free_();
  o->fm_table.free_();
  o->cfac.free_();
  o->scal.free_();
  o->exp.release();
  o->merge_slot.free_();
  o->touched.free_();
  o->touched_n.free_();
  o->tflag.free_();
  o->mb_samples.free_();
  o->gcbin.free_();
  o->gc_obs.free_(); o->posbin.free_(); o->pos_obs.free_(); o->lenclass.free_();
  o->assigned_prefix_b.free_();
  o->mass_acc.free_();
  o->uniq.free_();
  o->total.free_();
  o->lib_counts.free_();
  o->fld_cnt.free_(); o->err_cell.free_(); o->err_row.free_(); o->err_acc.free_(); o->err_flag.free_();
  o->ctr.free_();
  o->has_compat.free_();
  o->assigned_flag.free_();
  o->assigned_prefix.free_();
  o->awq.free_();
  o->abin.free_();
  o->rh1.free_();
  o->rh2.free_();
  o->rslot.free_();
  o->scan_tmp.free_();
  o->tk1.free_();
  o->tk2.free_();
  o->tcount.free_();
  o->tpool.free_();
  o->tn.free_();
  o->pool_tid.free_();
  o->pool_bin.free_();
  o->pool_wq.free_();
  o->pool_cursor.free_();
  delete o; c->online = nullptr;
}

static double forgetting_mass(sq_online_dev* o, double ff, uint64_t b) {  // ForgettingMassCalculator.hpp:30-40 — the schedule of sq_forgetting_masses (host/opts.cpp)
  if (o->fm_host.size() <= b) { const size_t n = std::max<size_t>(b + 1, o->fm_host.size() * 2 + 4096); o->fm_host.resize(n); (void)sq_forgetting_masses(ff, n, o->fm_host.data()); }
  return o->fm_host[b];
}

static int check_eq_overflow(sq_ctx* c) {
  unsigned long long cur[4];
  const bool timing = getenv("SQ_TIMING") != nullptr; auto tm0 = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[sq-timing] ovf %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - tm0).count());
    tm0 = t1;
  };
  SQ_HIP_CHECK(hipMemcpyAsync(cur, c->online->pool_cursor.p, sizeof(cur), hipMemcpyDeviceToHost, c->stream));
  mark("memcpyAsync");
  SQ_HIP_CHECK(hipStreamSynchronize(c->stream));   // not the null stream (device-wide implicit sync), not the eq streams (the runtime may still be retiring their thousands of launches)
  mark("streamSync");
  if (cur[2]) {
    sq_set_error("equivalence-class table overflow (%s): %llu classes, %llu labels", cur[2] == 1 ? "slots" : "label pool", cur[1], cur[0]);
    return SQ_ERR_OVERFLOW;
  }
  if (cur[1] * 10 > c->online->tcap * 7) {
    sq_set_error("equivalence-class table over 70%% full (%llu classes of %llu slots): call sq_ctx_reserve with the expected number of classes before the first batch",
        cur[1], (unsigned long long)c->online->tcap);
    return SQ_ERR_OVERFLOW;
  }
  return SQ_OK;
}

// wait until the worker has enqueued job `id` (ids count from 1)
void sq_eq_wait_enqueued(sq_ctx* c, uint64_t id) {
  std::unique_lock<std::mutex> lk(c->eq_mu);
  c->eq_cv_done.wait(lk, [&] { return c->eq_enqueued >= id; });
}
int sq_eq_sync(sq_ctx* c) {
  if (!c || !c->stream2) return SQ_OK;
  const bool timing = getenv("SQ_TIMING") != nullptr; auto tm0 = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[sq-timing] eq_sync %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - tm0).count());
    tm0 = t1;
  };
  { std::unique_lock<std::mutex> lk(c->eq_mu); c->eq_cv_done.wait(lk, [&] { return c->eq_enqueued >= c->eq_submitted; }); }
  mark("worker");
  SQ_HIP_CHECK(hipSetDevice(c->device));
  SQ_HIP_CHECK(hipStreamSynchronize(c->stream2));
  mark("stream2");
  if (c->stream3) SQ_HIP_CHECK(hipStreamSynchronize(c->stream3));
  mark("stream3");
  c->eq_pending[0] = c->eq_pending[1] = false;
  for (sq_ctx* sh : c->shadows) sh->eq_pending[0] = sh->eq_pending[1] = false;
  sq_prof_end(c, 1);
  mark("prof");
  {
    std::lock_guard<std::mutex> lk(c->eq_mu);
    if (c->eq_err) {
      int e = c->eq_err;
      sq_set_error("%s", c->eq_errmsg.c_str());
      c->eq_err = 0;
      c->eq_errmsg.clear();
      return e;
    }
  }
  int rc = check_eq_overflow(c);
  mark("overflow-check");
  return rc;
}

static int eq_accumulate_job(sq_ctx* c, const sq_ctx::eq_job& J);
static void eq_worker(sq_ctx* c) {
  (void)hipSetDevice(c->device);
  for (;;) {
    sq_ctx::eq_job J;
    { std::unique_lock<std::mutex> lk(c->eq_mu); c->eq_cv.wait(lk, [&] { return c->eq_stop || !c->eq_q.empty(); });
      if (c->eq_q.empty()) return;
      J = c->eq_q.front(); c->eq_q.pop_front(); }
    int rc = 0;
    { std::lock_guard<std::mutex> lk(c->eq_mu); rc = c->eq_err; }
    if (!rc) {
      rc = eq_accumulate_job(c, J);
      if (rc) {
        std::lock_guard<std::mutex> lk(c->eq_mu);
        if (!c->eq_err) {
          c->eq_err = rc;
          c->eq_errmsg = sq_last_error();
        }
      }
    }
    { std::lock_guard<std::mutex> lk(c->eq_mu); c->eq_enqueued++; }
    c->eq_cv_done.notify_all();
  }
}
void sq_eq_worker_stop(sq_ctx* c) {
  if (!c->eq_thread.joinable()) return;
  { std::lock_guard<std::mutex> lk(c->eq_mu); c->eq_stop = true; }
  c->eq_cv.notify_all(); c->eq_thread.join(); c->eq_stop = false;
}

extern "C" int sq_eq_accumulate(sq_ctx* c) {
  if (!c || c->owner || !c->api_have) { sq_set_error("sq_eq_accumulate: call sq_map_batch (or sq_map_wait) first"); return SQ_ERR_STATE; }
  c->api_have = false;
  sq_ctx* src = c->last_src ? c->last_src : c;   // the lane that mapped the batch
  if (c->acc_n == 0) return SQ_OK;
  c->online->exp.valid = false;
  sq_ctx::eq_job J; J.n = c->acc_n; J.buf = c->acc_buf; J.total_aln = c->acc_total_aln; J.joint = c->acc_joint; J.src = src;
  { std::lock_guard<std::mutex> lk(c->eq_mu);
    if (c->eq_err) { int e = c->eq_err; sq_set_error("%s", c->eq_errmsg.c_str()); return e; }   // an earlier batch failed
    if (!c->eq_thread.joinable()) c->eq_thread = std::thread(eq_worker, c);
    c->eq_q.push_back(J); src->eq_job_of_buf[J.buf] = ++c->eq_submitted; }
  src->eq_pending[J.buf] = true;
  c->eq_cv.notify_one();
  return SQ_OK;   // asynchronous: sq_eq_sync() (called by finish / fetch / merge / reset) waits and reports errors
}

// runs on the worker thread
static int eq_accumulate_job(sq_ctx* c, const sq_ctx::eq_job& J) {
  sq_online_dev* o = c->online; hipStream_t st = c->stream2; const uint32_t n = J.n; const sq_quant_opts& q = c->opts;
  const bool timing = getenv("SQ_TIMING") != nullptr; auto tm0 = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[sq-timing] eq_job %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - tm0).count());
    tm0 = t1;
  };
  if (c->stream3) {   // CU partition on: use the CU-masked stream only while a mapping batch is (about to be) in flight
    // give the caller ~200 us to enter the next sq_map_batch (sleep_for has ~50 us granularity: poll instead)
    for (auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(200); !c->map_active.load() &&
        std::chrono::steady_clock::now() < t_end;) std::this_thread::yield();
    if (!c->map_active.load()) st = c->stream3;
  }
  hipStream_t sq = st;
  c->eq_stream_cur = sq;
  if (c->ev_eq_last) SQ_HIP_CHECK(hipStreamWaitEvent(sq, c->ev_eq_last, 0));   // eq jobs run in order even when they change streams
  sq_ctx* src = J.src ? J.src : c;
  const int buf = J.buf; const sq_aln* d_aln = src->aln_ptr(buf); const uint64_t* d_aln_off = src->aln_off_ptr(buf);
  const uint64_t last_total_aln = J.total_aln;
  SQ_HIP_CHECK(hipStreamWaitEvent(sq, src->ev_map_done[buf], 0));   // alignments of this batch are complete
  const size_t A = (size_t)last_total_aln + 8;
  if (o->awq.ensure(A) || o->abin.ensure(A) || o->alp.ensure(A) || o->pre.ensure(A * sizeof(PreAln)) || (q.gc_bias && o->gcbin.ensure(A)) || (q.pos_bias && o->posbin.ensure(A))) {
    sq_set_error("device allocation failed (online scratch)");
    return SQ_ERR_NOMEM;
  }
  // [r5] the CIGAR error model: this batch must have come with its reads (sq_aln_inject_reads)
  const bool err_on = o->err_bins != 0; ErrReads ER{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (err_on) {
    if (!src->rd_have[buf]) { sq_set_error("the error model is on (sq_quant_opts.error_model) but the batch came without its reads: use sq_aln_inject_reads"); return SQ_ERR_STATE; }
    if (o->err_flag.ensure(A)) { sq_set_error("device allocation failed (error model flags)"); return SQ_ERR_NOMEM; }
    ER = ErrReads{src->rd_cig_off[buf].p, src->rd_cig[buf].p, src->rd_seq_off[buf].p, src->rd_seq[buf].p, src->rd_pos[buf].p, src->rd_score[buf].p};
  }
  OnlineView V = make_view(c);
  if (err_on) SQ_HIP_CHECK(hipMemsetAsync(o->err_flag.p, 0, A, sq));
  mark("pick-stream+ensure");
  sq_prof_begin(c, 1);
  uint8_t* d_gcbin = q.gc_bias ? o->gcbin.p : nullptr;
  if (last_total_aln) k_pre_aln<<<nblk(last_total_aln), TB, 0, sq>>>(last_total_aln, d_aln, c->di->ref_len, c->di->ref_clen, q,
      (PreAln*)o->pre.p, c->di->refseq, c->di->gcpre, c->di->ref_accum, d_gcbin, o->cmeans.p, q.pos_bias ? o->posbin.p : nullptr, o->lenclass.p);
  // assigned flags + exclusive prefix over the batch (model-independent: SPEC §D1)
  k_flag_compat<<<nblk(n + 1), TB, 0, sq>>>(n, 0, d_aln_off, d_aln, q, o->assigned_flag.p);
  if (o->scan_tmp.ensure((size_t)sqk::scan_tiles(n) * 8 + 256)) { sq_set_error("scan spine allocation failed"); return SQ_ERR_NOMEM; }
  sqk::exclusive_scan_u32_u64(o->assigned_flag.p, o->assigned_prefix.p, n, (uint64_t*)o->scan_tmp.p, sq);
  if (q.seq_bias) {   // observed read-start contexts of this batch (order-free integer counts, capped in read order)
    if (o->seq_flag.ensure((size_t)n + 2) || o->seq_pref.ensure((size_t)n + 2) || o->seq_code.ensure((size_t)n + 2)) { sq_set_error("device allocation failed (sequence-bias samples)"); return SQ_ERR_NOMEM; }
    k_seq_pick<<<nblk(n + 1), TB, 0, sq>>>(n, q.lib_type == 1 ? 1 : 0, c->reads_seen, q.seed, d_aln_off, d_aln, c->di->ref_len, c->di->refseq, c->di->ref_accum, o->seq_flag.p, o->seq_code.p);
    sqk::exclusive_scan_u32_u64(o->seq_flag.p, o->seq_pref.p, n, (uint64_t*)o->scan_tmp.p, sq);
    k_seq_count<<<nblk(n), TB, 0, sq>>>(n, o->seq_flag.p, o->seq_pref.p, o->seq_code.p, (uint64_t)q.num_bias_samples, o->seq_obs.p);
    k_seq_close<<<1, 64, 0, sq>>>(n, o->seq_pref.p, (uint64_t)q.num_bias_samples, o->seq_obs.p);
  }
  sq_prof_mark(c, SG_EQ_FLAGS, 1);
  const uint32_t mb = q.mini_batch_size ? q.mini_batch_size : 5000;
  const uint32_t nmb = (n + mb - 1) / mb;
  if (o->burned_known && !o->detect_active) {
    // [r3] burned in: one model-independent launch over the whole batch, then per group of W mini-batches only the mass terms
    // (k_frag_dynamic) and their application (k_apply_flagged).  Nothing comes back to the host.
    if (o->dyn.ensure(A * sizeof(DynAln))) { sq_set_error("device allocation failed (online scratch)"); return SQ_ERR_NOMEM; }
    TouchArgs TA{nullptr, mb * o->inflight, (o->M + AP_TB_ * 8 - 1) / (AP_TB_ * 8) * (AP_TB_ * 8)};
    {
      const size_t bytes = (size_t)((n + TA.gsize - 1) / TA.gsize) * TA.stride;
      if (o->gflag.ensure(bytes + 64)) { sq_set_error("device allocation failed (touched flags)"); return SQ_ERR_NOMEM; }
      SQ_HIP_CHECK(hipMemsetAsync(o->gflag.p, 0, bytes, sq));
      TA.flag = o->gflag.p;
    }
    if (err_on && n) k_err_like<<<nblk(n), TB, 0, sq>>>(V, ER, 0, n, d_aln_off, (PreAln*)o->pre.p, o->assigned_prefix.p, 0ull, 1, q.num_pre_burnin_frags, c->di->ref_len, c->di->ref_accum, c->di->refseq);   // the matrices are final after burn-in: once per batch
    k_frag_static<<<nblk(n), 256, 0, sq>>>(V, q, n, d_aln_off, (const PreAln*)o->pre.p, o->awq.p, o->abin.p, o->rh1.p, o->rh2.p, (DynAln*)o->dyn.p, TA);
    sq_prof_mark(c, SG_EQ_STATIC, 1);
    const uint32_t W = o->inflight;
    for (uint32_t b = 0; b < nmb;) {
      FmArr FM; uint32_t nw = 0; const uint32_t b0 = b;
      while (b < nmb && nw < W) { FM.v[nw++] = forgetting_mass(o, q.forgetting_factor, o->batch_no++); ++b; }
      for (uint32_t i = nw; i < SQ_MAX_INFLIGHT; ++i) FM.v[i] = 0.0;
      const uint32_t r0 = b0 * mb, r1 = (uint32_t)std::min<uint64_t>((uint64_t)b * mb, n);
      k_frag_dynamic<<<(r1 - r0 + 255) / 256, 256, 0, st>>>(V, r0, r1, mb, d_aln_off, (const DynAln*)o->dyn.p, d_gcbin);
      k_apply_flagged<<<TA.stride / (AP_TB_ * 4), AP_TB_, 0, st>>>(V, FM, nw, o->assigned_prefix.p, r0, r1, TA.flag + (size_t)(b0 / W) * TA.stride);
      o->group_no++; if (c->prof_on) c->eq_groups++;
    }
  } else {
  std::vector<uint64_t> prefix_host;  // host needs assigned totals per mini-batch boundary: copy the prefix at the boundaries only
  std::vector<uint64_t> bound(nmb + 1);
  if (o->rh2.n < nmb + 2) { sq_set_error("internal: bounds scratch too small"); return SQ_ERR_STATE; }
  k_gather_bounds<<<nblk(nmb + 1), TB, 0, st>>>(o->assigned_prefix.p, mb, n, nmb, o->rh2.p);   // rh2 is rewritten by the mini-batches below
  SQ_HIP_CHECK(hipMemcpyAsync(bound.data(), o->rh2.p, (size_t)(nmb + 1) * 8, hipMemcpyDeviceToHost, st));
  std::vector<uint32_t> mbs;   // `-l A`: per mini-batch, samples by observed format
  if (o->detect_active) {
    if (o->mb_samples.ensure((size_t)nmb * 64)) { sq_set_error("device allocation failed (library-type samples)"); return SQ_ERR_NOMEM; }
    k_mb_samples<<<nmb, TB, 0, st>>>(n, mb, d_aln_off, d_aln, q.lib_type, o->mb_samples.p);
    mbs.resize((size_t)nmb * 64);
    SQ_HIP_CHECK(hipMemcpyAsync(mbs.data(), o->mb_samples.p, (size_t)nmb * 64 * 4, hipMemcpyDeviceToHost, st));
  }
  unsigned long long hctr[8];
  SQ_HIP_CHECK(hipMemcpyAsync(hctr, o->ctr.p, sizeof(hctr), hipMemcpyDeviceToHost, st));
  mark("pre-launches");
  SQ_HIP_CHECK(hipStreamSynchronize(st));
  mark("bounds-sync");
  uint64_t assigned_base = hctr[0];
  bool burned_host = hctr[1] != 0;
  // Groups of up to W consecutive mini-batches share one model snapshot (SPEC §D1: the reference's W = numThreads workers read a
  // shared, slightly stale model, SalmonQuantify.cpp:2390-2403): ONE k_mini_batch over the group's fragments + ONE k_apply that
  // folds the mini-batches' increments in order, each with its own forgetting mass.  A group also ends at the mini-batch that
  // reaches numBurninFrags (the burn-in tables are made right there, as with W = 1) and at the end of the mapped batch.
  const uint32_t W = o->inflight;
  for (uint32_t b = 0; b < nmb;) {
    FmArr FM; uint32_t nw = 0; bool burn_now = false, detect_now = false; const uint32_t b0 = b;
    while (b < nmb && nw < W && !burn_now && !detect_now) {
      FM.v[nw++] = forgetting_mass(o, q.forgetting_factor, o->batch_no++);
      burn_now = !burned_host && assigned_base + bound[b + 1] >= q.num_burnin_frags;
      if (o->detect_active) {   // the group also ends at the mini-batch that completes the detector's 50 000 samples
        for (int f = 0; f < 64; ++f) { o->det_counts[f] += mbs[(size_t)b * 64 + f]; o->det_samples += mbs[(size_t)b * 64 + f]; }
        detect_now = o->det_samples >= 50000;
      }
      ++b;
    }
    for (uint32_t i = nw; i < SQ_MAX_INFLIGHT; ++i) FM.v[i] = 0.0;
    const uint32_t r0 = b0 * mb, r1 = (uint32_t)std::min<uint64_t>((uint64_t)b * mb, n);
    const uint64_t assigned_after = assigned_base + bound[b];
    const uint32_t par = (uint32_t)(o->group_no & 1);
    if (err_on && r1 > r0) k_err_like<<<nblk(r1 - r0), TB, 0, st>>>(V, ER, r0, r1, d_aln_off, (PreAln*)o->pre.p, o->assigned_prefix.p, (unsigned long long)assigned_base, 0, q.num_pre_burnin_frags, c->di->ref_len, c->di->ref_accum, c->di->refseq);
    k_mini_batch<<<(uint32_t)(((uint64_t)(r1 - r0) * MB_G + 255) / 256), 256, 0, st>>>(V, q, r0, r1, mb, c->reads_seen + r0, d_aln_off, d_aln,
        (const PreAln*)o->pre.p,
        o->assigned_prefix.p, assigned_base, o->awq.p, o->alp.p, o->abin.p, o->rh1.p, o->rh2.p, par, d_gcbin);
    { const uint32_t mass_blocks = std::min<uint32_t>((o->M + AP_TB - 1) / AP_TB, 256u); const int with_fld = burned_host ? 0 : 1;
      if (err_on && !burned_host && r1 > r0) {   // what the group's drawn alignments teach the matrices (AlignmentModel::update), folded in mini-batch by mini-batch
        k_err_count<<<nblk(r1 - r0), TB, 0, st>>>(V, ER, r0, r1, d_aln_off, (const PreAln*)o->pre.p, c->di->ref_len, c->di->ref_accum, c->di->refseq);
        k_err_apply<<<2 * o->err_bins * 82, 128, 0, st>>>(V, FM, nw);
      }
      k_apply<<<mass_blocks + (uint32_t)with_fld, AP_TB, 0, st>>>(V, FM, nw, assigned_after, q.num_burnin_frags, mass_blocks, with_fld, par); }
    if (burn_now) {
      k_burnin_tables<<<1, 64, 0, st>>>(V, 0);
      k_burnin_efflen<<<nblk(o->M), TB, 0, st>>>(V, 2);
      k_burnin_done<<<1, 64, 0, st>>>(V);
      burned_host = true;
    }
    o->group_no++; if (c->prof_on) c->eq_groups++;
    if (detect_now) {
      // mostLikelyType (LibraryTypeDetector.hpp:33-152): from here on the online model expects the detected format.  What depended
      // on the format — the per-alignment compatibility flags, the assigned flags and their prefix — is made again for the rest of
      // the batch; the rows before r1 keep what they were given (their groups have run).
      uint8_t nt, no_, ns; sq_detect_lib_format(q.lib_type, o->det_counts, &nt, &no_, &ns);
      c->opts.lib_type = nt; c->opts.lib_orientation = no_; c->opts.lib_strand = ns;
      o->detect_active = false; o->detected = true;
      if (b < nmb) {
        const uint32_t rs = r1;
        assigned_base = assigned_after;
        if (last_total_aln) k_pre_aln<<<nblk(last_total_aln), TB, 0, st>>>(last_total_aln, d_aln, c->di->ref_len, c->di->ref_clen, q, (PreAln*)o->pre.p,
            c->di->refseq, c->di->gcpre, c->di->ref_accum, d_gcbin, o->cmeans.p, q.pos_bias ? o->posbin.p : nullptr, o->lenclass.p);
        k_flag_compat<<<nblk(n + 1), TB, 0, st>>>(n, rs, d_aln_off, d_aln, q, o->assigned_flag.p);
        sqk::exclusive_scan_u32_u64(o->assigned_flag.p, o->assigned_prefix.p, n, (uint64_t*)o->scan_tmp.p, st);
        // the mini-batches still to come read rh1/rh2 only after writing them: rh2 doubles as the bounds scratch again
        sq_dbuf<uint64_t>& scratch = o->assigned_prefix_b;
        if (scratch.ensure(nmb + 2)) { sq_set_error("device allocation failed (bounds scratch)"); return SQ_ERR_NOMEM; }
        k_gather_bounds<<<nblk(nmb + 1), TB, 0, st>>>(o->assigned_prefix.p, mb, n, nmb, scratch.p);
        SQ_HIP_CHECK(hipMemcpyAsync(bound.data(), scratch.p, (size_t)(nmb + 1) * 8, hipMemcpyDeviceToHost, st));
        SQ_HIP_CHECK(hipStreamSynchronize(st));
      }
    }
  }
  if (burned_host) o->burned_known = true;
  }
  sq_prof_mark(c, SG_EQ_MINIBATCH, 1);
  // eq-class table: insert labels, then add counts / fixed-point weights (labels, bins and weights are the static stage's)
  EqView T = make_eq_view(o);
  k_eq_insert<<<nblk(n), TB, 0, sq>>>(T, n, d_aln_off, d_aln, o->abin.p, o->rh1.p, o->rh2.p, o->rslot.p, q.range_factorization_bins > 0);
  k_eq_add<<<nblk(n), TB, 0, sq>>>(T, n, d_aln_off, o->abin.p, o->awq.p, o->rslot.p);
  sq_prof_mark(c, SG_EQ_TABLE, 1);
  SQ_HIP_CHECK(hipEventRecord(src->ev_eq_done[buf], st)); c->ev_eq_last = src->ev_eq_done[buf];
  mark("chain-launches");
  o->num_observed += n; o->num_mapped_ub += J.joint; c->reads_seen += n;
  return SQ_OK;
}

extern "C" int sq_ctx_reset(sq_ctx* c) {
  if (!c) return SQ_ERR_ARG;
  (void)sq_eq_sync(c);
  SQ_HIP_CHECK(hipSetDevice(c->device));
  // the export buffers and their page-locked staging area are work buffers (sized by earlier jobs / sq_ctx_reserve): they survive
  auto keep = c->online->exp; c->online->exp = decltype(keep)();
  sq_online_free(c);
  c->reads_seen = 0; c->have_batch = false; c->api_have = false;
  int rc = sq_online_create(c);
  if (rc == SQ_OK) { keep.valid = false; keep.model_valid = false; keep.E = keep.L = 0; c->online->exp = keep; }
  else { keep.release(); }
  return rc;
}

// [r4] SPEC MG, the shared burn-in prefix: every rank runs the batches up to the end of the burn-in itself, so every rank holds the same model; ranks other
// than 0 then forget what the prefix ADDED — classes, counts, observed bias masses — and keep what it TAUGHT: the fragment-length tables, the effective
// lengths, the burned-in flag, and the transcript masses, which move into the prior term (logAdd(prior, mass) stays what it was, bit for bit, while
// `mass` restarts at LOG_0 and ends the run as this rank's own increments — what the rank-ordered merge of the masses wants to add).
__global__ void k_fold_mass(uint32_t M, double* __restrict__ prior, double* __restrict__ mass) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; if (t >= M) return;
  prior[t] = sq_log_add(prior[t], mass[t]); mass[t] = SQ_LOG_0;
}
extern "C" int sq_model_drop_counts(sq_ctx* c) {
  if (!c || c->owner || !c->online) { sq_set_error("sq_model_drop_counts: bad context"); return SQ_ERR_ARG; }
  { int rs = sq_eq_sync(c); if (rs) return rs; }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  sq_online_dev* o = c->online; hipStream_t st = c->stream2; const uint32_t M = o->M;
  k_fold_mass<<<nblk(M), TB, 0, st>>>(M, o->prior_mass.p, o->mass.p);
  SQ_HIP_CHECK(hipMemsetAsync(o->uniq.p, 0, (size_t)M * 8, st)); SQ_HIP_CHECK(hipMemsetAsync(o->total.p, 0, (size_t)M * 8, st));
  SQ_HIP_CHECK(hipMemsetAsync(o->lib_counts.p, 0, 64 * 8, st));
  SQ_HIP_CHECK(hipMemsetAsync(o->gc_obs.p, 0, (SQ_GC_COND_BINS * SQ_GC_FRAG_BINS + 8) * 8, st)); SQ_HIP_CHECK(hipMemsetAsync(o->pos_obs.p, 0, 208 * 8, st));
  SQ_HIP_CHECK(hipMemsetAsync(o->seq_obs.p, 0, 1160 * 8, st));
  SQ_HIP_CHECK(hipMemsetAsync(o->ctr.p + 0, 0, 8, st)); SQ_HIP_CHECK(hipMemsetAsync(o->ctr.p + 5, 0, 8, st));          // numAssigned, numCompatible (the burned-in flag and the FLD's bookkeeping stay)
  SQ_HIP_CHECK(hipMemsetAsync(o->tk1.p, 0xFF, o->tcap * 8, st)); SQ_HIP_CHECK(hipMemsetAsync(o->tk2.p, 0, o->tcap * 8, st));
  SQ_HIP_CHECK(hipMemsetAsync(o->tcount.p, 0, o->tcap * 8, st)); SQ_HIP_CHECK(hipMemsetAsync(o->tn.p, 0, o->tcap * 4, st));
  SQ_HIP_CHECK(hipMemsetAsync(o->pool_wq.p, 0, o->pool_cap * 8, st)); SQ_HIP_CHECK(hipMemsetAsync(o->pool_cursor.p, 0, 4 * 8, st));
  SQ_HIP_CHECK(hipStreamSynchronize(st));
  o->num_observed = 0; o->num_mapped_ub = 0; o->exp.valid = false; o->exp.model_valid = false;
  return SQ_OK;
}

extern "C" int sq_model_summary_get(sq_ctx* c, sq_model_summary* out) {
  if (!c || !out) return SQ_ERR_ARG;
  { int rs = sq_eq_sync(c); if (rs) return rs; }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  unsigned long long hctr[8]; SQ_HIP_CHECK(hipMemcpy(hctr, c->online->ctr.p, sizeof(hctr), hipMemcpyDeviceToHost));
  out->num_observed = c->online->num_observed;
  out->num_assigned = hctr[0];
  out->num_mapped_ub = c->online->num_mapped_ub;
  out->burned_in = hctr[1] != 0;
  out->num_compatible = hctr[5];
  out->lib_format_id = (uint32_t)(c->opts.lib_type | (c->opts.lib_orientation << 1) | (c->opts.lib_strand << 3));
  out->lib_detected = c->online->detected ? 1u : 0u;
  return SQ_OK;
}

// finalisation when burn-in was never reached (SalmonQuantify.cpp:2734-2745): effective lengths from the observed FLD
static int finish_efflen(sq_ctx* c) {
  unsigned long long hctr[8]; SQ_HIP_CHECK(hipMemcpy(hctr, c->online->ctr.p, sizeof(hctr), hipMemcpyDeviceToHost));
  if (hctr[1] != 0 || hctr[4] == 4) return SQ_OK;
  OnlineView V = make_view(c);
  k_burnin_tables<<<1, 64, 0, c->stream>>>(V, 1);
  k_burnin_efflen<<<nblk(c->online->M), TB, 0, c->stream>>>(V, 3);
  k_burnin_done<<<1, 64, 0, c->stream>>>(V);
  SQ_HIP_CHECK(hipStreamSynchronize(c->stream));
  return SQ_OK;
}

extern "C" int sq_model_fetch(sq_ctx* c, double* log_mass, uint64_t* unique_count, uint64_t* total_count, double* log_eff_len) {
  if (!c) return SQ_ERR_ARG;
  if (c->online->exp.valid && c->online->exp.model_valid) {   // staged together with the eq-class export: no GPU round trip
    const auto& X = c->online->exp; const size_t M = c->online->M; const uint8_t* h = X.host + X.model_off;
    if (log_mass) memcpy(log_mass, h, M * 8); if (unique_count) memcpy(unique_count, h + M * 8, M * 8);
    if (total_count) memcpy(total_count, h + 2 * M * 8, M * 8); if (log_eff_len) memcpy(log_eff_len, h + 3 * M * 8, M * 8);
    return SQ_OK;
  }
  { int rs = sq_eq_sync(c); if (rs) return rs; }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  int rc = finish_efflen(c); if (rc) return rc;
  sq_online_dev* o = c->online; size_t M = o->M;
  if (log_mass) SQ_HIP_CHECK(hipMemcpy(log_mass, o->mass.p, M * 8, hipMemcpyDeviceToHost));
  if (unique_count) SQ_HIP_CHECK(hipMemcpy(unique_count, o->uniq.p, M * 8, hipMemcpyDeviceToHost));
  if (total_count) SQ_HIP_CHECK(hipMemcpy(total_count, o->total.p, M * 8, hipMemcpyDeviceToHost));
  if (log_eff_len) SQ_HIP_CHECK(hipMemcpy(log_eff_len, o->log_eff_len.p, M * 8, hipMemcpyDeviceToHost));
  return SQ_OK;
}

extern "C" int sq_model_fetch_lib_counts(sq_ctx* c, uint64_t* out64) {
  if (!c || !out64) return SQ_ERR_ARG;
  { int rs = sq_eq_sync(c); if (rs) return rs; }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  SQ_HIP_CHECK(hipMemcpy(out64, c->online->lib_counts.p, 64 * 8, hipMemcpyDeviceToHost));
  return SQ_OK;
}

// observed fragment-GC masses (observedGCMass, SalmonQuantify.cpp:938-972): [SQ_GC_COND_BINS][SQ_GC_FRAG_BINS] sums of the normalised
// alignment probabilities, linear space
extern "C" int sq_model_fetch_gc_observed(sq_ctx* c, double* out75) {
  if (!c || !out75) return SQ_ERR_ARG;
  if (!c->opts.gc_bias) { sq_set_error("sq_model_fetch_gc_observed: the context was created without gc_bias"); return SQ_ERR_STATE; }
  { int rs = sq_eq_sync(c); if (rs) return rs; }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  unsigned long long h[SQ_GC_COND_BINS * SQ_GC_FRAG_BINS];
  SQ_HIP_CHECK(hipMemcpy(h, c->online->gc_obs.p, sizeof(h), hipMemcpyDeviceToHost));
  for (int i = 0; i < SQ_GC_COND_BINS * SQ_GC_FRAG_BINS; ++i) out75[i] = sq_from_fixed(h[i], 32);
  return SQ_OK;
}

// observed read-start masses by length class (observedPosBiasFwd / RC, SalmonQuantify.cpp:895-934): [5', 3'][class][bin] sums of the
// normalised alignment probabilities, linear space, WITHOUT the initial mass of the reference's models (sq_bias_eff_lengths adds it)
extern "C" int sq_model_fetch_pos_observed(sq_ctx* c, double* out200) {
  if (!c || !out200) return SQ_ERR_ARG;
  if (!c->opts.pos_bias) { sq_set_error("sq_model_fetch_pos_observed: the context was created without pos_bias"); return SQ_ERR_STATE; }
  { int rs = sq_eq_sync(c); if (rs) return rs; }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  unsigned long long h[200];
  SQ_HIP_CHECK(hipMemcpy(h, c->online->pos_obs.p, sizeof(h), hipMemcpyDeviceToHost));
  for (int i = 0; i < 200; ++i) out200[i] = sq_from_fixed(h[i], 32);
  return SQ_OK;
}

// [r5] the error model's matrices (log space): cells [2][bins][82][82], row sums [2][bins][82]; *bins_out = the number of read-position bins
extern "C" int sq_model_fetch_error_model(sq_ctx* c, double* cells, double* rows, uint32_t* bins_out) {
  if (!c) return SQ_ERR_ARG;
  if (!c->online->err_bins) { sq_set_error("sq_model_fetch_error_model: the context was created without error_model"); return SQ_ERR_STATE; }
  { int rs = sq_eq_sync(c); if (rs) return rs; }
  SQ_HIP_CHECK(hipSetDevice(c->device)); const size_t b = c->online->err_bins;
  if (bins_out) *bins_out = (uint32_t)b;
  if (cells) SQ_HIP_CHECK(hipMemcpy(cells, c->online->err_cell.p, 2 * b * 82 * 82 * 8, hipMemcpyDeviceToHost));
  if (rows) SQ_HIP_CHECK(hipMemcpy(rows, c->online->err_row.p, 2 * b * 82 * 8, hipMemcpyDeviceToHost));
  return SQ_OK;
}

extern "C" int sq_model_fetch_seq_observed(sq_ctx* c, uint64_t* fw576, uint64_t* rc576, uint64_t* nsamples) {
  if (!c || !fw576 || !rc576) return SQ_ERR_ARG;
  if (!c->opts.seq_bias) { sq_set_error("sq_model_fetch_seq_observed: the context was created without seq_bias"); return SQ_ERR_STATE; }
  { int rs = sq_eq_sync(c); if (rs) return rs; }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  unsigned long long h[1153]; SQ_HIP_CHECK(hipMemcpy(h, c->online->seq_obs.p, sizeof(h), hipMemcpyDeviceToHost));
  memcpy(fw576, h, 576 * 8); memcpy(rc576, h + 576, 576 * 8); if (nsamples) *nsamples = h[1152];
  return SQ_OK;
}

extern "C" int sq_model_fld_min(sq_ctx* c, uint32_t* min_len) {
  if (!c || !min_len) return SQ_ERR_ARG;
  { int rs = sq_eq_sync(c); if (rs) return rs; }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  unsigned long long hctr[8]; SQ_HIP_CHECK(hipMemcpy(hctr, c->online->ctr.p, sizeof(hctr), hipMemcpyDeviceToHost));
  *min_len = hctr[2] >= 1000 ? 1u : (uint32_t)hctr[2];   // min_ == hist_.size() - 1 means nothing was added
  return SQ_OK;
}
extern "C" int sq_model_fetch_fld(sq_ctx* c, double* out) {
  if (!c || !out) return SQ_ERR_ARG;
  { int rs = sq_eq_sync(c); if (rs) return rs; }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  sq_online_dev* o = c->online; unsigned long long hctr[8]; double scal[8]; std::vector<double> h(1024);
  SQ_HIP_CHECK(hipMemcpy(hctr, o->ctr.p, sizeof(hctr), hipMemcpyDeviceToHost));
  SQ_HIP_CHECK(hipMemcpy(scal, o->scal.p, sizeof(scal), hipMemcpyDeviceToHost));
  if (hctr[3]) {
    SQ_HIP_CHECK(hipMemcpy(h.data(), o->cpmf.p, 1024 * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i <= 1000; ++i) out[i] = h[i];
  }
  else {
    SQ_HIP_CHECK(hipMemcpy(h.data(), o->hist.p, 1024 * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i <= 1000; ++i) out[i] = h[i] - scal[0];
  }
  return SQ_OK;
}

// export in canonical order (ascending (h1,h2)); sizes first when arrays are NULL.  Compaction, sort
// (rocPRIM radix sort on h1), offsets (scan) and the gather all run on the device; only the compact
// CSR crosses PCIe.
// device-side export into o->exp (see sq_online_dev::eq_export)
static int eq_export_run(sq_ctx* c) {
  const bool timing = getenv("SQ_TIMING") != nullptr; auto tm0 = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[sq-timing] eq_export %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - tm0).count());
    tm0 = t1;
  };
  { int rs = sq_eq_sync(c); if (rs) return rs; }
  mark("sync");
  SQ_HIP_CHECK(hipSetDevice(c->device));
  sq_online_dev* o = c->online; hipStream_t st = c->stream; auto& X = o->exp;
  unsigned long long cur[4], hctr[8]; SQ_HIP_CHECK(hipMemcpyAsync(cur, o->pool_cursor.p, sizeof(cur), hipMemcpyDeviceToHost, st));
  SQ_HIP_CHECK(hipMemcpyAsync(hctr, o->ctr.p, sizeof(hctr), hipMemcpyDeviceToHost, st)); SQ_HIP_CHECK(hipStreamSynchronize(st));
  mark("cursor-read");
  const uint64_t E = cur[1], L = cur[0];
  const bool model_final = hctr[1] != 0 || hctr[4] == 4;   // effective lengths already final (burned in / finalised): the model summary can ride along
  X.E = E; X.L = L; X.valid = false;
  X.model_valid = false;
  if (E == 0) return SQ_OK;   // nothing staged; fetches fall back to direct copies
  EqView T = make_eq_view(o);
  if (X.keys.ensure(E) || X.keys2.ensure(E) || X.slots.ensure(E) || X.slots2.ensure(E) || X.nlab.ensure(E + 1) || X.d_off.ensure(E + 1) ||
      X.d_tid.ensure(L) ||
      X.d_bins.ensure(L) || X.d_wq.ensure(L) || X.d_w.ensure(L) ||
      X.d_cnt.ensure(E) || X.d_h1.ensure(E) || X.d_h2.ensure(E) || X.d_ctr.ensure(1) ||
          X.d_tie.ensure(1)) { sq_set_error("device allocation failed (eq export)"); return SQ_ERR_NOMEM; }
  const size_t M = o->M; const size_t need = 32 * E + 24 * L + 64 + 32 * M;
  if (X.host_cap < need) { if (X.host) (void)hipHostFree(X.host); X.host = nullptr; X.host_cap = 0; const size_t cap = need + need / 4;
    if (hipHostMalloc((void**)&X.host, cap,
        hipHostMallocDefault) != hipSuccess) { sq_set_error("pinned staging allocation failed (eq export, %zu bytes)",
        cap); return SQ_ERR_NOMEM; } X.host_cap = cap; }
  mark("alloc");
  SQ_HIP_CHECK(hipMemsetAsync(X.d_ctr.p, 0, 8, st)); SQ_HIP_CHECK(hipMemsetAsync(X.d_tie.p, 0, 4, st));
  k_eq_collect<<<nblk(T.cap), TB, 0, st>>>(T, X.keys.p, X.slots.p, X.d_ctr.p);
  size_t tb = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tb, X.keys.p, X.keys2.p, X.slots.p, X.slots2.p, (size_t)E, 0u, 64u, st);
  size_t tb2 = 0; (void)rocprim::exclusive_scan(nullptr, tb2, X.nlab.p, X.d_off.p, (uint64_t)0, (size_t)E + 1, rocprim::plus<uint64_t>(), st);
  if (X.tmp.ensure(std::max(tb, tb2) + 256)) { sq_set_error("device allocation failed (eq export temp)"); return SQ_ERR_NOMEM; }
  tb = X.tmp.n;
  SQ_HIP_CHECK(rocprim::radix_sort_pairs(X.tmp.p, tb, X.keys.p, X.keys2.p, X.slots.p, X.slots2.p, (size_t)E, 0u, 64u, st));
  k_eq_sizes<<<nblk(E + 1), TB, 0, st>>>(T, E, X.slots2.p, X.keys2.p, X.nlab.p, X.d_tie.p);
  tb2 = X.tmp.n; SQ_HIP_CHECK(rocprim::exclusive_scan(X.tmp.p, tb2, X.nlab.p, X.d_off.p, (uint64_t)0, (size_t)E + 1, rocprim::plus<uint64_t>(), st));
  uint32_t tie = 0; SQ_HIP_CHECK(hipMemcpyAsync(&tie, X.d_tie.p, 4, hipMemcpyDeviceToHost, st)); SQ_HIP_CHECK(hipStreamSynchronize(st));
  if (tie) {  // two classes share a sort key and arrived out of (h1,h2) order: re-sort the slot list on the host (rare)
    std::vector<uint32_t> hs(E), ord(E); std::vector<unsigned long long> hk(E), k1(o->tcap), k2(o->tcap);
    SQ_HIP_CHECK(hipMemcpy(hs.data(), X.slots2.p, E * 4, hipMemcpyDeviceToHost));
    SQ_HIP_CHECK(hipMemcpy(hk.data(), X.keys2.p, E * 8, hipMemcpyDeviceToHost));
    SQ_HIP_CHECK(hipMemcpy(k1.data(), o->tk1.p, o->tcap * 8, hipMemcpyDeviceToHost));
    SQ_HIP_CHECK(hipMemcpy(k2.data(), o->tk2.p, o->tcap * 8, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < E; ++i) ord[i] = (uint32_t)i;
    std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { if (hk[x] != hk[y]) return hk[x] < hk[y]; const uint32_t a = hs[x],
        b = hs[y]; return k1[a] < k1[b] || (k1[a] == k1[b] && k2[a] < k2[b]); });
    std::vector<uint32_t> hs2(E); for (uint64_t i = 0; i < E; ++i) hs2[i] = hs[ord[i]];
    SQ_HIP_CHECK(hipMemcpy(X.slots2.p, hs2.data(), E * 4, hipMemcpyHostToDevice));
    k_eq_sizes<<<nblk(E + 1), TB, 0, st>>>(T, E, X.slots2.p, X.keys2.p, X.nlab.p, X.d_tie.p);
    SQ_HIP_CHECK(rocprim::exclusive_scan(X.tmp.p, tb2, X.nlab.p, X.d_off.p, (uint64_t)0, (size_t)E + 1, rocprim::plus<uint64_t>(), st));
  }
  k_eq_gather<<<nblk(E), TB, 0, st>>>(T, E, X.slots2.p, X.d_off.p, X.d_tid.p, X.d_w.p, X.d_wq.p, X.d_cnt.p, X.d_bins.p, X.d_h1.p, X.d_h2.p);
  // staging layout: off[E+1] | count[E] | h1[E] | h2[E] | wq[L] | w[L] | tid[L] | bins[L]
  uint8_t* h = X.host;
  SQ_HIP_CHECK(hipMemcpyAsync(h, X.d_off.p, (E + 1) * 8, hipMemcpyDeviceToHost, st)); h += (E + 1) * 8;
  SQ_HIP_CHECK(hipMemcpyAsync(h, X.d_cnt.p, E * 8, hipMemcpyDeviceToHost, st)); h += E * 8;
  SQ_HIP_CHECK(hipMemcpyAsync(h, X.d_h1.p, E * 8, hipMemcpyDeviceToHost, st)); h += E * 8;
  SQ_HIP_CHECK(hipMemcpyAsync(h, X.d_h2.p, E * 8, hipMemcpyDeviceToHost, st)); h += E * 8;
  SQ_HIP_CHECK(hipMemcpyAsync(h, X.d_wq.p, L * 8, hipMemcpyDeviceToHost, st)); h += L * 8;
  SQ_HIP_CHECK(hipMemcpyAsync(h, X.d_w.p, L * 8, hipMemcpyDeviceToHost, st)); h += L * 8;
  SQ_HIP_CHECK(hipMemcpyAsync(h, X.d_tid.p, L * 4, hipMemcpyDeviceToHost, st)); h += L * 4;
  SQ_HIP_CHECK(hipMemcpyAsync(h, X.d_bins.p, L * 4, hipMemcpyDeviceToHost, st)); h += L * 4;
  h = X.host + (((size_t)(h - X.host) + 7) & ~(size_t)7); X.model_off = (size_t)(h - X.host);
  if (model_final) {
    SQ_HIP_CHECK(hipMemcpyAsync(h, o->mass.p, M * 8, hipMemcpyDeviceToHost, st)); h += M * 8;
    SQ_HIP_CHECK(hipMemcpyAsync(h, o->uniq.p, M * 8, hipMemcpyDeviceToHost, st)); h += M * 8;
    SQ_HIP_CHECK(hipMemcpyAsync(h, o->total.p, M * 8, hipMemcpyDeviceToHost, st)); h += M * 8;
    SQ_HIP_CHECK(hipMemcpyAsync(h, o->log_eff_len.p, M * 8, hipMemcpyDeviceToHost, st));
  }
  SQ_HIP_CHECK(hipStreamSynchronize(st));
  X.model_valid = model_final;
  mark("kernels+d2h");
  X.valid = true;
  return SQ_OK;
}

extern "C" int sq_eq_finish(sq_ctx* c, sq_eq_table* out) {
  if (!c || !out) return SQ_ERR_ARG;
  sq_online_dev* o = c->online; auto& X = o->exp;
  if (!X.valid) { int rc = eq_export_run(c); if (rc) return rc; }
  const uint64_t E = X.E, L = X.L;
  out->num_classes = E; out->num_labels = L;
  if (!out->off) return SQ_OK;          // size query; the table is already staged for the fetch that follows
  if (E == 0) { out->off[0] = 0; return SQ_OK; }
  if (!out->tid || !out->w || !out->count) { sq_set_error("sq_eq_finish: output arrays missing"); return SQ_ERR_ARG; }
  const bool timing = getenv("SQ_TIMING") != nullptr; auto tm0 = std::chrono::steady_clock::now();
  // host copy out of the pinned staging area, a few threads wide
  struct Seg { uint8_t* dst; const uint8_t* src; size_t n; };
  std::vector<Seg> segs; const uint8_t* h = X.host;
  auto add = [&](void* dst, size_t n) { if (dst && n) segs.push_back({(uint8_t*)dst, h, n}); h += n; };
  add(out->off, (E + 1) * 8);
  add(out->count, E * 8);
  add(out->h1, E * 8);
  add(out->h2, E * 8);
  add(out->wq, L * 8);
  add(out->w, L * 8);
  add(out->tid, L * 4);
  add(out->bins, L * 4);
  std::vector<Seg> chunks; const size_t CH = 4u << 20;
  for (auto& sg : segs) for (size_t p = 0; p < sg.n; p += CH) chunks.push_back({sg.dst + p, sg.src + p, std::min(CH, sg.n - p)});
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= chunks.size()) break;
      memcpy(chunks[i].dst, chunks[i].src, chunks[i].n);
    }
  };
  const unsigned nth = (unsigned)std::min<size_t>(8, std::max<size_t>(1, chunks.size() / 2));
  std::vector<std::thread> th; for (unsigned i = 1; i < nth; ++i) th.emplace_back(work);
  work(); for (auto& t : th) t.join();
  if (timing) fprintf(stderr, "[sq-timing] eq_finish host-copy %.3f ms (%u threads)\n", std::chrono::duration<double,
      std::milli>(std::chrono::steady_clock::now() - tm0).count(), nth);
  return SQ_OK;
}

// merge a table whose arrays already live on this ctx's device (e.g. gathered over RCCL straight into HBM)
extern "C" int sq_eq_merge_device(sq_ctx* c, const sq_eq_table* t) {
  if (!c || !t || !t->off || !t->tid || !t->wq || !t->count || !t->h1 || !t->h2) {
    sq_set_error("sq_eq_merge_device: table must carry off/tid/wq/count/h1/h2");
    return SQ_ERR_ARG;
  }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  const uint64_t E = t->num_classes; if (E == 0) return SQ_OK;
  { int rs = sq_eq_sync(c); if (rs) return rs; }
  c->online->exp.valid = false;
  sq_dbuf<uint32_t>& d_slot = c->online->merge_slot;
  if (d_slot.ensure(E)) { sq_set_error("device allocation failed (eq merge)"); return SQ_ERR_NOMEM; }
  EqView T = make_eq_view(c->online);
  k_eq_merge_insert<<<nblk(E), TB, 0, c->stream>>>(T, E, t->off, t->tid, t->bins, (const uint64_t*)t->h1, (const uint64_t*)t->h2, d_slot.p);
  k_eq_merge_add<<<nblk(E), TB, 0, c->stream>>>(T, E, t->off, t->wq, t->count, d_slot.p);
  SQ_HIP_CHECK(hipStreamSynchronize(c->stream));
  return check_eq_overflow(c);
}

extern "C" int sq_eq_merge(sq_ctx* c, const sq_eq_table* t) {
  if (!c || !t || !t->off || !t->tid || !t->wq || !t->count || !t->h1 || !t->h2) {
    sq_set_error("sq_eq_merge: table must carry off/tid/wq/count/h1/h2");
    return SQ_ERR_ARG;
  }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  const uint64_t E = t->num_classes, L = t->num_labels; if (E == 0) return SQ_OK;
  sq_dbuf<uint64_t> d_off, d_wq, d_cnt, d_h1, d_h2; sq_dbuf<uint32_t> d_tid, d_bins;
  if (d_off.ensure(E + 1) || d_wq.ensure(L) || d_cnt.ensure(E) || d_h1.ensure(E) || d_h2.ensure(E) || d_tid.ensure(L) || d_bins.ensure(L)) {
    sq_set_error("device allocation failed (eq merge)");
    return SQ_ERR_NOMEM;
  }
  SQ_HIP_CHECK(hipMemcpy(d_off.p, t->off, (E + 1) * 8, hipMemcpyHostToDevice));
  SQ_HIP_CHECK(hipMemcpy(d_wq.p, t->wq, L * 8, hipMemcpyHostToDevice));
  SQ_HIP_CHECK(hipMemcpy(d_cnt.p, t->count, E * 8, hipMemcpyHostToDevice));
  SQ_HIP_CHECK(hipMemcpy(d_h1.p, t->h1, E * 8, hipMemcpyHostToDevice));
  SQ_HIP_CHECK(hipMemcpy(d_h2.p, t->h2, E * 8, hipMemcpyHostToDevice));
  SQ_HIP_CHECK(hipMemcpy(d_tid.p, t->tid, L * 4, hipMemcpyHostToDevice));
  if (t->bins) SQ_HIP_CHECK(hipMemcpy(d_bins.p, t->bins, L * 4, hipMemcpyHostToDevice));
  sq_eq_table dt = *t;
  dt.off = d_off.p;
  dt.wq = d_wq.p;
  dt.count = d_cnt.p;
  dt.h1 = d_h1.p;
  dt.h2 = d_h2.p;
  dt.tid = d_tid.p;
  dt.bins = t->bins ? d_bins.p : nullptr;
  dt.w = nullptr;
  int rc = sq_eq_merge_device(c, &dt);
  d_off.free_(); d_wq.free_(); d_cnt.free_(); d_h1.free_(); d_h2.free_(); d_tid.free_(); d_bins.free_();
  return rc;
}

// the canonical-order export as DEVICE pointers (valid until the next accumulate / merge / reset of this ctx)
extern "C" int sq_eq_export_device(sq_ctx* c, sq_eq_table* out) {
  if (!c || !out) return SQ_ERR_ARG;
  auto& X = c->online->exp;
  if (!X.valid) { int rc = eq_export_run(c); if (rc) return rc; }
  memset(out, 0, sizeof(*out)); out->num_classes = X.E; out->num_labels = X.L;
  if (X.E) {
    out->off = X.d_off.p;
    out->tid = X.d_tid.p;
    out->w = X.d_w.p;
    out->wq = (uint64_t*)X.d_wq.p;
    out->count = (uint64_t*)X.d_cnt.p;
    out->bins = X.d_bins.p;
    out->h1 = (uint64_t*)X.d_h1.p;
    out->h2 = (uint64_t*)X.d_h2.p;
  }
  return SQ_OK;
}

int sq_eq_export_dev(sq_ctx* c, sq_eq_dev_csr* out) {
  auto& X = c->online->exp;
  if (!X.valid) { int rc = eq_export_run(c); if (rc) return rc; }
  out->E = X.E; out->L = X.L; out->off = X.d_off.p; out->tid = X.d_tid.p; out->w = X.d_w.p; out->cnt = X.d_cnt.p;
  return SQ_OK;
}

// eq == NULL: optimise over the ctx's own accumulated classes, straight from the export that already sits in HBM
extern "C" int sq_em_optimize(sq_ctx* c, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, double* alpha_out,
    sq_em_report* rep) {
  if (!c) { sq_set_error("sq_em_optimize: null ctx"); return SQ_ERR_ARG; }
  if (eq) return sq_em_optimize_dev(c->device, eq, txp, o, alpha_out, rep);
  if (!txp || !o || !alpha_out || !txp->eff_len) { sq_set_error("sq_em_optimize: bad arguments"); return SQ_ERR_ARG; }
  sq_eq_dev_csr dv; int rc = sq_eq_export_dev(c, &dv); if (rc) return rc;
  if (dv.E == 0) { sq_set_error("sq_em_optimize: the ctx holds no equivalence classes"); return SQ_ERR_STATE; }
  return sq_em_optimize_impl(c->device, nullptr, &dv, txp, o, alpha_out, rep, &c->em_arena, (void*)(c->stream3 ? c->stream3 : c->stream2));   // the eq stage's streams are idle after the export
}

extern "C" int sq_em_optimize_bias(sq_ctx* c, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, sq_efflen_cb cb, void* user,
    double* alpha_out, double* eff_len_out, sq_em_report* rep) {
  if (!c) { sq_set_error("sq_em_optimize_bias: null ctx"); return SQ_ERR_ARG; }
  if (eq) return sq_em_optimize_bias_impl(c->device, eq, nullptr, txp, o, cb, user, alpha_out, eff_len_out, rep, nullptr, nullptr);
  sq_eq_dev_csr dv; int rc = sq_eq_export_dev(c, &dv); if (rc) return rc;
  if (dv.E == 0) { sq_set_error("sq_em_optimize_bias: the ctx holds no equivalence classes"); return SQ_ERR_STATE; }
  return sq_em_optimize_bias_impl(c->device, nullptr, &dv, txp, o, cb, user, alpha_out, eff_len_out, rep, &c->em_arena, (void*)(c->stream3 ? c->stream3 : c->stream2));
}

// Pre-size everything the end of a job allocates — the device buffers and the page-locked staging area of the eq-class
// export, and the EM workspace — for up to max_classes classes with max_labels label entries (0, 0: the reference's own
// initial map size, 10^6 classes, with 6 labels each).  Without it the first export / EM of a job pays ≈ 10 ms of hipMalloc / hipHostMalloc; larger
// jobs than reserved still work (the buffers grow).  
extern "C" int sq_ctx_reserve(sq_ctx* c, uint64_t max_classes, uint64_t max_labels) {
  if (!c || c->owner) { sq_set_error("sq_ctx_reserve: bad context"); return SQ_ERR_ARG; }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  sq_online_dev* o = c->online; auto& X = o->exp; const size_t M = o->M;
  uint64_t E = max_classes ? max_classes : 1000000;   // countMap_.reserve(1000000), EquivalenceClassBuilder.hpp:140
  uint64_t L = max_labels ? max_labels : 6 * E;
  // the class table itself (open addressing, at most 70 % full; 2^22 slots by default = 2.9 M classes) is sized here too: it can
  // only be re-made while it is empty — there is no rehash — so a job that expects more classes reserves before its first batch
  if (E * 10 > o->tcap * 7 || L > o->pool_cap) {
    { int rs = sq_eq_sync(c); if (rs) return rs; }
    unsigned long long cur[4]; SQ_HIP_CHECK(hipMemcpy(cur, o->pool_cursor.p, sizeof(cur), hipMemcpyDeviceToHost));
    if (cur[1]) {
      sq_set_error("sq_ctx_reserve: the class table already holds %llu classes; reserve %llu classes before the first sq_eq_accumulate (or after sq_ctx_reset)",
          cur[1],
          (unsigned long long)E);
      return SQ_ERR_STATE;
    }
    uint64_t cap = o->tcap; while (E * 10 > cap * 7) cap <<= 1;
    const uint64_t pcap = std::max<uint64_t>(std::max<uint64_t>(o->pool_cap, cap * 4), L);
    if (cap >= (1ull << 32) || pcap >= (1ull << 40)) {
      sq_set_error("sq_ctx_reserve: %llu classes / %llu labels are beyond the table's addressing", (unsigned long long)E,
          (unsigned long long)L);
      return SQ_ERR_ARG;
    }
    if (o->tk1.ensure(cap) || o->tk2.ensure(cap) || o->tcount.ensure(cap) || o->tpool.ensure(cap) || o->tn.ensure(cap) ||
        o->pool_tid.ensure(pcap) ||
        o->pool_bin.ensure(pcap) || o->pool_wq.ensure(pcap)) {
      sq_set_error("device allocation failed (class table for %llu classes)", (unsigned long long)E);
      return SQ_ERR_NOMEM;
    }
    o->tcap = cap; o->pool_cap = pcap;
    SQ_HIP_CHECK(hipMemset(o->tk1.p, 0xFF, cap * 8));
    SQ_HIP_CHECK(hipMemset(o->tk2.p, 0, cap * 8));
    SQ_HIP_CHECK(hipMemset(o->tcount.p, 0, cap * 8));
    SQ_HIP_CHECK(hipMemset(o->tn.p, 0, cap * 4));
    SQ_HIP_CHECK(hipMemset(o->pool_wq.p, 0, pcap * 8));
  }
  if (X.keys.ensure(E) || X.keys2.ensure(E) || X.slots.ensure(E) || X.slots2.ensure(E) || X.nlab.ensure(E + 1) || X.d_off.ensure(E + 1) ||
      X.d_tid.ensure(L) ||
      X.d_bins.ensure(L) || X.d_wq.ensure(L) || X.d_w.ensure(L) ||
      X.d_cnt.ensure(E) || X.d_h1.ensure(E) || X.d_h2.ensure(E) || X.d_ctr.ensure(1) || X.d_tie.ensure(1) ||
          X.tmp.ensure((size_t)32 << 20)) { sq_set_error("device allocation failed (sq_ctx_reserve)"); return SQ_ERR_NOMEM; }
  X.valid = false; X.model_valid = false;   // buffers may have moved: a staged export is made again on its next use
  const size_t need = 32 * E + 24 * L + 64 + 32 * M;
  if (X.host_cap < need) { if (X.host) (void)hipHostFree(X.host); X.host = nullptr; X.host_cap = 0;
    if (hipHostMalloc((void**)&X.host, need,
        hipHostMallocDefault) != hipSuccess) { sq_set_error("pinned staging allocation failed (sq_ctx_reserve, %zu bytes)",
        need); return SQ_ERR_NOMEM; } X.host_cap = need; }
  return sq_em_arena_reserve(&c->em_arena, sq_em_workspace_bytes(E, L, M), (size_t)3 * M * 8,
      (size_t)(std::max<uint64_t>(E, L / 64 + M) + 1) * 4);
}
