// hip/em.hip — collapsed EM / VBEM over equivalence classes on gfx950 (seam B3).
//
// Replaces CollapsedEMOptimizer::optimize + EMUpdate_/VBEMUpdate_ (reference
// src/inference/CollapsedEMOptimizer.cpp:178-328,732-1035).  The reference scatters
// count*theta_t*w_ct/denom_c into alphaOut with CAS-loop atomic<double> adds under TBB
// (SalmonUtils.hpp:166-172): arbitrary summation order.  Here one iteration is
//   (1) digamma/exp pass        theta_t = exp(psi(alpha_t+prior_t) - psi(sum))      [M threads]
//   (2) class pass              inv_c   = count_c / sum_t theta_t*w_ct              [one wave-lane per class, label-major CSR]
//   (3) transcript pass         alpha'_t = sum_{c∋t} theta_t*w_ct*inv_c             [transcript-major CSC; blocked-64 sums, SPEC §D4:
//                               level 1 sums runs of 64 CSC entries, level 2 runs of 64 partials, ... so no thread ever adds
//                               more than 64 terms and highly expressed transcripts do not serialise the iteration]
// with NO floating-point atomics: the transcript-major pass makes every sum order-defined, so the
// GPU result is bit-identical run to run and to the CPU checker.  Memory-bound gather/stream:
// per iteration 36·L + 16·E + 64·M algorithmic bytes.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <vector>
#include <algorithm>
#include <cmath>
#include "device_index.h"
#include "../../../include/sq_rng.h"
#include <chrono>
#include <cstdlib>
#include <cstdio>

static inline hipError_t sq_em_wait(hipStream_t st) { return hipStreamSynchronize(st); }
namespace {

// SQ_TIMING=1: host-side phase timings on stderr (diagnostics only)
struct PhaseTimer {
  bool on; std::chrono::steady_clock::time_point t0; const char* tag;
  explicit PhaseTimer(const char* tg) : on(getenv("SQ_TIMING") != nullptr), t0(std::chrono::steady_clock::now()), tag(tg) {}
  void mark(const char* what) {
    if (!on) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[sq-timing] %s %s %.3f ms\n", tag, what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

// ---- SPEC §D2 canonical sum: 64-wide strided-halving tree, applied level by level -------------
__device__ inline double wave_halving_sum(double v) {
  for (int s = 32; s >= 1; s >>= 1) v = v + __shfl_down(v, s, 64);
  return v;  // valid in lane 0
}

// level kernel: out[g] = tree(in[64g .. 64g+63]) ; if prior != nullptr the leaf is in[i] + prior[i]
__global__ void k_sum_level(const double* __restrict__ in, const double* __restrict__ prior, uint32_t n, double* __restrict__ out) {
  uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  double v = 0.0;
  if (gid < n) v = prior ? (in[gid] + prior[gid]) : in[gid];
  v = wave_halving_sum(v);
  if ((threadIdx.x & 63) == 0 && (gid >> 6) < ((n + 63) >> 6)) out[gid >> 6] = v;
}

struct EmDev {
  uint32_t M; uint32_t E; uint64_t L;
  const uint64_t* off; const uint32_t* tid; const double* cw; const double* cnt;      // label-major
  const uint64_t* t_off; const uint32_t* t_cls; const double* t_cw;                    // transcript-major
  // blocked-64 reduction plan: level l has nseg[l] segments; segment g sums src[lo .. lo+cnt) and writes dst
  const uint32_t* seg_lo[4]; const uint8_t* seg_cnt[4]; const uint32_t* seg_txp[4]; uint32_t nseg[4]; double* part[4]; int nlevels;
  const double* prior;
  double* theta; double* inv;
  double* partial;   // scratch for the sum levels
  uint32_t n_last;   // number of partials at the last level (<= 64)
  uint32_t* flags;   // [0] = done (iteration count at convergence, 0 = running), [1] = not-converged marker, [2] = iters executed
  unsigned long long* maxrel;  // bit pattern of max relDiff (non-negative doubles order like integers)
  double tol; int use_vbem; uint32_t min_iter;
  double first_add;  // [r5] 1.0 in the first iteration of optimize() without VBEM, else 0: the reference leaves alphasPrime at 1.0 after its initialisation
                     // (CollapsedEMOptimizer.cpp:797-823) and its EMUpdate_ (:178-234) adds into it without clearing it first — VBEMUpdate_ (:265-283) clears —
                     // so every transcript starts the second iteration one count up.  Found by the pin against the compiled optimiser (tests/test_vbem_pin.py)
  // k_l1 work plan: block b stages the CSC entries of level-1 segments [chunk_seg[b], chunk_seg[b+1]) through LDS
  const uint32_t* cchunk; uint32_t ncchunks;   // k_class work plan: block b owns classes [cchunk[b], cchunk[b+1])
  const uint32_t* chunk_seg; uint32_t nchunks; const uint8_t* t_seg8;   // t_seg8[p] = index of entry p's run within its block
  // level 2 folded into k_fin (plans with exactly two levels): transcript t sums part[0][l2_lo[t] .. +l2_cnt[t])
  const uint32_t* l2_lo; const uint8_t* l2_cnt;
  // [r4] three launches per iteration: k_fin also leaves psi[t] = digamma(alpha'_t + prior_t) (-inf where the VBEM rule zeroes theta), its last
  // block to finish closes the iteration and publishes logNorm; k_class / k_l1 form theta = exp(psi - logNorm) where they gather it
  int psi_mode; const double* psi; const double* log_norm; double* psi_out; double* log_norm_out;
  // [r6] three launches per iteration (k_class3, k_l13, k_fin4): see EmIterState / FinRec below
  struct EmIterState* state; struct FinRec* rec; uint32_t nrec;   // nrec = 4 * ceil(ceil(M / 64) / 64) records
  const uint4* cplan; const uint4* lplan;                          // per block of k_class3 / k_l13: {first unit, end unit, first entry, end entry}
};
// [r6] The bookkeeping of an iteration lives in two alternating slots: the class pass of iteration `it` reads slot it & 1 (written by the class pass before it — an
// earlier kernel) and its block 0 writes slot (it + 1) & 1, which k_l13 / k_fin4 of the same iteration and the next class pass read.  No kernel reads what it writes.
struct EmIterState { uint32_t done; uint32_t closed; unsigned long long maxrel; };   // done: iteration count at convergence (0 = running); closed: iterations closed so far
// What a block of k_fin4 leaves for the next class pass: a QUARTER of a level-2 tree of the canonical sum (SPEC D2) and what the block saw of the convergence test.
// Block b = 4 G + i owns the level-1 groups 64 G + i + 4 k (k = 0..15: its 16 waves) of level-2 group G; the strided-halving tree over the 64 leaves of G adds
// leaf j to leaf j + 32, + 16, + 8, + 4 — within this block's leaves — and only then across blocks: sum(G) = (q0 + q2) + (q1 + q3), the tree's last two steps.
struct FinRec { double q; unsigned long long relbad; };   // relbad: bits 0..62 = bit pattern of the block's max relDiff (>= 0), bit 63 = "a transcript moved more than the tolerance"

__device__ inline bool em_close(EmDev& d, uint32_t it_index, unsigned long long* maxrel_log) {
  uint32_t it = it_index + 1;
  d.flags[2] = it;
  bool conv = (__hip_atomic_load(&d.flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0);   // written by other blocks: bypass the CU-local L1
  maxrel_log[0] = __hip_atomic_load(d.maxrel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool done = conv && it >= d.min_iter;
  if (done) d.flags[0] = it;
  d.flags[1] = 0;
  *d.maxrel = 0ULL;
  return done;
}

// First kernel of an iteration (ONE block): closes the previous iteration's convergence bookkeeping,
// then finishes the canonical sum of (alpha + prior) from the level-1 partials (SPEC §D2: 64-leaf
// strided-halving trees, level by level) and publishes logNorm = digamma(sum).  Being a single
// block, the `done` flag it may set is visible to every later kernel of the iteration.
__global__ void __launch_bounds__(1024) k_top(EmDev d, const double* __restrict__ partials, uint32_t n1, int close_prev, uint32_t prev_it,
    unsigned long long* maxrel_log, double* __restrict__ log_norm) {
  __shared__ double buf[2][4096];
  if (threadIdx.x == 0 && close_prev && !d.flags[0]) em_close(d, prev_it, maxrel_log);
  __shared__ int s_stop;
  if (threadIdx.x == 0) s_stop = __hip_atomic_load(&d.flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // what this thread may just have written
  __syncthreads();
  if (s_stop || n1 == 0) return;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (uint32_t i = threadIdx.x; i < n1; i += blockDim.x) buf[0][i] = partials[i];
  __syncthreads();
  int cur = 0; uint32_t n = n1;
  for (;;) {
    const uint32_t g = (n + 63) / 64;
    for (uint32_t j = wave; j < g; j += nw) {
      uint32_t i = j * 64 + lane;
      double v = (i < n) ? buf[cur][i] : 0.0;
      v = wave_halving_sum(v);
      if (lane == 0) buf[cur ^ 1][j] = v;
    }
    __syncthreads();
    cur ^= 1; n = g;
    if (n == 1) break;
  }
  if (threadIdx.x == 0) *log_norm = sq_digamma(buf[cur][0]);
}

__global__ void k_theta(EmDev d, const double* __restrict__ alpha, const double* __restrict__ log_norm) {
  if (d.flags[0]) return;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.M) {
    const double logNorm = *log_norm;
    double ap = alpha[i] + d.prior[i];
    d.theta[i] = (ap > 1e-10) ? sq_exp(sq_digamma(ap) - logNorm) : 0.0;  // digammaMin (:43)
  }
}

// theta_t where a kernel gathers it: the stored value (EM: alpha; five-launch VBEM: k_theta's table) or, in psi mode, exp(psi_t - logNorm) —
// the same two operations k_theta performs, on the same operands
__device__ inline double theta_from(const EmDev& d, double x, double ln) { return d.psi_mode ? (x == -HUGE_VAL ? 0.0 : sq_exp(x - ln)) : x; }
// class pass: inv_c = count_c / sum_t theta_t * w_ct.  A block owns a run of consecutive classes whose
// label entries (<= CL_CHUNK) are staged through LDS: coalesced loads of (tid, weight), all theta[]
// gathers in flight at once, the products parked in LDS, then one thread per class adds its terms in
// label order.  A skipped term (VBEM, theta = 0) is stored as +0.0, which leaves the non-negative
// sum unchanged bit for bit.  A single class larger than the chunk (rare) is walked by one thread.
#define CL_CHUNK_5 1024
#define CL_TB 256
__global__ void __launch_bounds__(CL_TB) k_class(EmDev d, const double* __restrict__ theta) {
  if (d.flags[0]) return;
  __shared__ double s_term[CL_CHUNK_5];
  const uint32_t c0 = d.cchunk[blockIdx.x], c1 = d.cchunk[blockIdx.x + 1];
  const uint64_t e0 = d.off[c0], e1 = d.off[c1];
  const double ln = d.psi_mode ? *d.log_norm : 0.0;
  if (e1 - e0 > CL_CHUNK_5) {   // c1 == c0 + 1
    if (threadIdx.x == 0) {
      double denom = 0.0;
      for (uint64_t i = e0; i < e1; ++i) { const double th = theta_from(d, theta[d.tid[i]], ln); if (!d.use_vbem || th > 0.0) denom += th * d.cw[i]; }
      d.inv[c0] = (denom <= 2.2250738585072014e-308) ? 0.0 : d.cnt[c0] / denom;
    }
    return;
  }
  if (e1 > e0) {
    uint32_t t[CL_CHUNK_5 / CL_TB]; double w[CL_CHUNK_5 / CL_TB], h[CL_CHUNK_5 / CL_TB];
#pragma unroll
    for (int j = 0; j < CL_CHUNK_5 / CL_TB; ++j) {
      const uint64_t p = e0 + j * CL_TB + threadIdx.x;
      const uint64_t q = p < e1 ? p : e1 - 1;
      t[j] = d.tid[q];
      w[j] = d.cw[q];
    }
#pragma unroll
    for (int j = 0; j < CL_CHUNK_5 / CL_TB; ++j) h[j] = theta[t[j]];
    if (d.psi_mode) {
#pragma unroll
      for (int j = 0; j < CL_CHUNK_5 / CL_TB; ++j) h[j] = theta_from(d, h[j], ln);
    }
#pragma unroll
    for (int j = 0; j < CL_CHUNK_5 / CL_TB; ++j) {
      const uint64_t p = e0 + j * CL_TB + threadIdx.x;
      if (p < e1) s_term[p - e0] = (!d.use_vbem || h[j] > 0.0) ? h[j] * w[j] : 0.0;
    }
  }
  __syncthreads();
  for (uint32_t c = c0 + threadIdx.x; c < c1; c += CL_TB) {
    const uint64_t a = d.off[c], b = d.off[c + 1];
    if (b - a <= 1) { d.inv[c] = (b - a == 1) ? -d.cnt[c] : 0.0; continue; }  // single-transcript class gets the full count (:316-318)
    double denom = 0.0;
    const uint32_t lo = (uint32_t)(a - e0), n = (uint32_t)(b - a);
    for (uint32_t i = 0; i < n; ++i) denom += s_term[lo + i];
    d.inv[c] = (denom <= 2.2250738585072014e-308) ? 0.0 : d.cnt[c] / denom;  // minEQClassWeight (:40)
  }
}

#define SEG_TOP 0x80000000u
// level 1: one thread per run of <= 64 consecutive CSC entries of one transcript.  A block first
// evaluates the terms of all its runs with coalesced loads and every inv[] gather in flight at once
// (theta of the owning run comes from LDS via a 1-byte run index per entry), parks them in LDS, then
// each thread adds up its own run left to right (SPEC §D4) — no chain of dependent global gathers.
// A skipped term is stored as +0.0, which leaves the non-negative running sum unchanged bit for bit.
#define L1_CHUNK_5 1024
#define L1_TB 256
__global__ void __launch_bounds__(L1_TB) k_l1(EmDev d, const double* __restrict__ theta, double* __restrict__ alpha_out) {
  if (d.flags[0]) return;
  __shared__ double s_term[L1_CHUNK_5]; __shared__ double s_th[L1_TB];
  const uint32_t s0 = d.chunk_seg[blockIdx.x], s1 = d.chunk_seg[blockIdx.x + 1];
  const uint32_t e0 = d.seg_lo[0][s0], e1 = d.seg_lo[0][s1 - 1] + d.seg_cnt[0][s1 - 1];
  const uint32_t g = s0 + threadIdx.x; const bool has = g < s1;
  uint32_t tt = 0, lo = 0, n = 0;
  uint32_t c[L1_CHUNK_5 / L1_TB]; double w[L1_CHUNK_5 / L1_TB], iv[L1_CHUNK_5 / L1_TB]; uint8_t sg[L1_CHUNK_5 / L1_TB];
#pragma unroll
  for (int j = 0; j < L1_CHUNK_5 / L1_TB; ++j) {
    const uint32_t p = e0 + j * L1_TB + threadIdx.x;
    const uint32_t q = p < e1 ? p : e1 - 1;
    c[j] = d.t_cls[q];
    w[j] = d.t_cw[q];
    sg[j] = d.t_seg8[q];
  }
  if (has) { tt = d.seg_txp[0][g]; lo = d.seg_lo[0][g] - e0; n = d.seg_cnt[0][g]; s_th[threadIdx.x] = theta_from(d, theta[tt & ~SEG_TOP], d.psi_mode ? *d.log_norm : 0.0); }
#pragma unroll
  for (int j = 0; j < L1_CHUNK_5 / L1_TB; ++j) iv[j] = d.inv[c[j]];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < L1_CHUNK_5 / L1_TB; ++j) {
    const uint32_t p = e0 + j * L1_TB + threadIdx.x;
    if (p < e1) {
      const double th = s_th[sg[j]]; double term = 0.0;
      if (iv[j] < 0.0) term = -iv[j];                                                   // single-transcript class: the full count
      else if (iv[j] != 0.0 && (!d.use_vbem || th > 0.0)) { const double v = th * w[j]; term = v * iv[j]; }
      s_term[p - e0] = term;
    }
  }
  __syncthreads();
  if (!has) return;
  double acc = 0.0;
  for (uint32_t i = 0; i < n; ++i) acc += s_term[lo + i];
  if (tt & SEG_TOP) alpha_out[tt & ~SEG_TOP] = acc; else d.part[0][g] = acc;
}
// levels >= 2 (runs of <= 64 partial sums of the previous level): few segments, so ONE block walks
// the levels with a barrier between them instead of one launch per level
__global__ void k_level(EmDev d, int lvl, double* __restrict__ alpha_out) {   // grid version for levels with many segments
  if (d.flags[0]) return;
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= d.nseg[lvl]) return;
  const uint32_t tt = d.seg_txp[lvl][g];
  const double* src = d.part[lvl - 1] + d.seg_lo[lvl][g]; const uint32_t n = d.seg_cnt[lvl][g];
  double acc = 0.0;
  for (uint32_t i = 0; i < n; ++i) acc += src[i];
  if (tt & SEG_TOP) alpha_out[tt & ~SEG_TOP] = acc; else d.part[lvl][g] = acc;
}
__global__ void __launch_bounds__(1024) k_upper(EmDev d, int first_lvl, double* __restrict__ alpha_out) {
  if (d.flags[0]) return;
  for (int lvl = first_lvl; lvl < d.nlevels; ++lvl) {
    for (uint32_t g = threadIdx.x; g < d.nseg[lvl]; g += blockDim.x) {
      const uint32_t tt = d.seg_txp[lvl][g];
      const double* src = d.part[lvl - 1] + d.seg_lo[lvl][g]; const uint32_t n = d.seg_cnt[lvl][g];
      double acc = 0.0;
      for (uint32_t i = 0; i < n; ++i) acc += src[i];
      if (tt & SEG_TOP) alpha_out[tt & ~SEG_TOP] = acc; else d.part[lvl][g] = acc;
    }
    __syncthreads();
  }
}
// convergence scan (CollapsedEMOptimizer.cpp:945-957); also level 1 of the next iteration's canonical
// sum and, for two-level plans, level 2 of the blocked-64 transcript sums.
// (A "last block finishes the sum" variant was measured and dropped: the device-scope fence it needs
// writes back the XCD's L2 in every block — 53 us vs 9 us per launch on MI355X.)
__global__ void __launch_bounds__(256) k_fin(EmDev d, const double* __restrict__ alpha, double* __restrict__ alpha_out,
    double* __restrict__ partials) {
  if (d.flags[0]) return;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  double rel = -1.0; int bad = 0; double leaf = 0.0;
  if (t < d.M) {
    const bool empty = d.t_off[t + 1] == d.t_off[t];
    double acc;
    uint32_t n2 = d.l2_cnt ? d.l2_cnt[t] : 0;
    if (n2) {   // level 2 of the blocked-64 sum, folded in: <= 64 level-1 partials, left to right
      const double* src = d.part[0] + d.l2_lo[t];
      acc = 0.0;
      // eight partials are requested together, then added left to right (a missing one is +0.0: x + 0.0 = x for these non-negative sums):
      // the additions keep their order, the loads no longer wait for one another
      for (uint32_t i = 0; i < n2; i += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (i + k < n2) ? src[i + k] : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
      }
      alpha_out[t] = acc;
    } else if (empty) { acc = 0.0; alpha_out[t] = 0.0; }
    else acc = alpha_out[t];
    if (d.first_add != 0.0) { acc += d.first_add; alpha_out[t] = acc; }
    leaf = acc + d.prior[t];
    if (acc > 1e-2) {  // alphaCheckCutoff (:884)
      rel = fabs(alpha[t] - acc) / acc;
      if (rel > d.tol) bad = 1;
    }
  }
  // level 1 of next iteration's canonical sum
  if (partials) {
    double ls = wave_halving_sum(leaf);
    if ((threadIdx.x & 63) == 0 && (t >> 6) < ((d.M + 63) >> 6)) partials[t >> 6] = ls;
  }
  for (int s = 32; s >= 1; s >>= 1) {
    double o = __shfl_down(rel, s, 64);
    int ob = __shfl_down(bad, s, 64);
    rel = o > rel ? o : rel;
    bad |= ob;
  }
  // block-level combine, then one atomic per block only when it can raise the running maximum
  __shared__ double srel[16]; __shared__ int sbad[16];
  if ((threadIdx.x & 63) == 0) { srel[threadIdx.x >> 6] = rel; sbad[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) { for (uint32_t w = 1; w < (blockDim.x >> 6); ++w) { if (srel[w] > rel) rel = srel[w]; bad |= sbad[w]; } }
  if (threadIdx.x == 0) {
    if (rel >= 0.0) {
      unsigned long long b = (unsigned long long)__double_as_longlong(rel);
      if (b > __hip_atomic_load(d.maxrel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(d.maxrel, b);
    }
    if (bad && __hip_atomic_load(&d.flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __hip_atomic_store(&d.flags[1], 1u,
        __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT);
  }
}

// one thread: close the iteration (convergence bookkeeping lives on the device so the host never
// has to synchronise inside the loop)
__global__ void k_close(EmDev d, uint32_t it_index /* 0-based index of the iteration just run */, unsigned long long* maxrel_log) {
  if (d.flags[0]) return;
  em_close(d, it_index, maxrel_log);
}

// ---- [r6] the three-launch iteration ------------------------------------------------------------------------------------------------------------------------
// What k_top did in a launch of its own — close the iteration before (convergence test over k_fin's per-block results), finish the canonical sum of alpha + prior and
// take its digamma — is the prologue of EVERY block of the class pass: one wave reads the <= 256 records k_fin4 left (64 bytes per lane, one request), while the
// block's other waves already have the loads of their class chunk in flight.  Every block derives the same values from the same inputs; block 0 publishes them.
// Returns true when the iteration must not run (converged before, or just now).
__device__ inline bool em_prologue(const EmDev& d, uint32_t it, int close_prev, double* s_ln, uint32_t* s_done) {
  if (threadIdx.x < 64) {
    const uint32_t lane = threadIdx.x, ng = d.nrec >> 2;
    const EmIterState sin = d.state[it & 1];
    double p2 = 0.0; unsigned long long rb = 0;
    if (lane < ng) {
      const FinRec* r = d.rec + 4 * lane; const FinRec r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
      p2 = (r0.q + r2.q) + (r1.q + r3.q);                       // the last two steps of the level-2 tree
      rb = r0.relbad; rb = r1.relbad > rb ? r1.relbad : rb; rb = r2.relbad > rb ? r2.relbad : rb; rb = r3.relbad > rb ? r3.relbad : rb;   // (bit 63 dominates: max is also the OR of the flags)
    }
    for (int s = 32; s >= 1; s >>= 1) { const unsigned long long o = __shfl_down(rb, s, 64); rb = o > rb ? o : rb; }
    const double sum = wave_halving_sum(p2);                    // level 3: the tree over the <= 64 level-2 sums
    if (lane == 0) {
      EmIterState so = sin;
      if (!sin.done && close_prev) {
        so.closed = it; so.maxrel = rb & 0x7FFFFFFFFFFFFFFFull;
        if (!(rb >> 63) && it >= d.min_iter) so.done = it;
      }
      if (blockIdx.x == 0) d.state[(it + 1) & 1] = so;
      *s_done = so.done;
      if (!so.done && d.use_vbem) { const double ln = sq_digamma(sum); *s_ln = ln; if (blockIdx.x == 0) *d.log_norm_out = ln; }
    }
  }
  __syncthreads();
  return *s_done != 0;
}
// closes the last iteration of a chunk before the host looks (the class pass that would have done it has not been launched yet); what it writes is what that class
// pass will write again from the same inputs
__global__ void __launch_bounds__(64) k_close3(EmDev d, uint32_t it) { __shared__ double s_ln; __shared__ uint32_t s_done; EmDev e = d; e.use_vbem = 0; (void)em_prologue(e, it, 1, &s_ln, &s_done); }

template <int CL_CHUNK>
__global__ void __launch_bounds__(CL_TB) k_class3(EmDev d, const double* __restrict__ theta, uint32_t it, int close_prev) {
  __shared__ double s_term[CL_CHUNK]; __shared__ uint16_t s_off[CL_CHUNK + 2]; __shared__ double s_ln; __shared__ uint32_t s_done;
  const uint4 pl = d.cplan[blockIdx.x]; const uint32_t c0 = pl.x, c1 = pl.y; const uint64_t e0 = pl.z, e1 = pl.w;
  const bool big = e1 - e0 > CL_CHUNK;   // one class larger than the chunk (c1 == c0 + 1): walked by one thread
  uint32_t t[CL_CHUNK / CL_TB]; double w[CL_CHUNK / CL_TB], h[CL_CHUNK / CL_TB];
  if (!big && e1 > e0) {
#pragma unroll
    for (int j = 0; j < CL_CHUNK / CL_TB; ++j) { const uint64_t p = e0 + j * CL_TB + threadIdx.x; const uint64_t q = p < e1 ? p : e1 - 1; t[j] = d.tid[q]; w[j] = d.cw[q]; }
    for (uint32_t i = threadIdx.x; i <= c1 - c0; i += CL_TB) s_off[i] = (uint16_t)(d.off[c0 + i] - e0);     // class bounds inside the chunk (<= 2048): requested with the entries, not after them
  }
  double cn[CL_CHUNK / CL_TB];      // the counts of this thread's classes, requested now (a chunk holds at most CL_CHUNK classes)
#pragma unroll
  for (int r = 0; r < CL_CHUNK / CL_TB; ++r) { const uint32_t c = c0 + r * CL_TB + threadIdx.x; cn[r] = (!big && c < c1) ? d.cnt[c] : 0.0; }
  if (em_prologue(d, it, close_prev, &s_ln, &s_done)) return;
  const double ln = d.psi_mode ? s_ln : 0.0;
  if (big) {
    if (threadIdx.x == 0) {
      double denom = 0.0;
      for (uint64_t i = e0; i < e1; ++i) { const double th = theta_from(d, theta[d.tid[i]], ln); if (!d.use_vbem || th > 0.0) denom += th * d.cw[i]; }
      d.inv[c0] = (denom <= 2.2250738585072014e-308) ? 0.0 : d.cnt[c0] / denom;
    }
    return;
  }
  if (e1 > e0) {
#pragma unroll
    for (int j = 0; j < CL_CHUNK / CL_TB; ++j) h[j] = theta[t[j]];
    if (d.psi_mode) {
#pragma unroll
      for (int j = 0; j < CL_CHUNK / CL_TB; ++j) h[j] = theta_from(d, h[j], ln);
    }
#pragma unroll
    for (int j = 0; j < CL_CHUNK / CL_TB; ++j) { const uint64_t p = e0 + j * CL_TB + threadIdx.x; if (p < e1) s_term[p - e0] = (!d.use_vbem || h[j] > 0.0) ? h[j] * w[j] : 0.0; }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < CL_CHUNK / CL_TB; ++r) {
    const uint32_t c = c0 + r * CL_TB + threadIdx.x; if (c >= c1) break;
    const uint32_t lo = e1 > e0 ? s_off[c - c0] : 0, n = e1 > e0 ? (uint32_t)s_off[c - c0 + 1] - lo : 0;
    if (n <= 1) { d.inv[c] = (n == 1) ? -cn[r] : 0.0; continue; }  // single-transcript class gets the full count (:316-318)
    double denom = 0.0;
    for (uint32_t i = 0; i < n; ++i) denom += s_term[lo + i];
    d.inv[c] = (denom <= 2.2250738585072014e-308) ? 0.0 : cn[r] / denom;  // minEQClassWeight (:40)
  }
}
template <int L1_CHUNK>
__global__ void __launch_bounds__(L1_TB) k_l13(EmDev d, const double* __restrict__ theta, double* __restrict__ alpha_out, uint32_t it) {
  __shared__ double s_term[L1_CHUNK]; __shared__ double s_th[L1_TB];
  const uint4 pl = d.lplan[blockIdx.x]; const uint32_t s0 = pl.x, s1 = pl.y, e0 = pl.z, e1 = pl.w;
  const uint32_t g = s0 + threadIdx.x; const bool has = g < s1;
  uint32_t tt = 0, lo = 0, n = 0;
  uint32_t c[L1_CHUNK / L1_TB]; double w[L1_CHUNK / L1_TB], iv[L1_CHUNK / L1_TB]; uint8_t sg[L1_CHUNK / L1_TB];
#pragma unroll
  for (int j = 0; j < L1_CHUNK / L1_TB; ++j) { const uint32_t p = e0 + j * L1_TB + threadIdx.x; const uint32_t q = p < e1 ? p : e1 - 1; c[j] = d.t_cls[q]; w[j] = d.t_cw[q]; sg[j] = d.t_seg8[q]; }
  if (has) { tt = d.seg_txp[0][g]; lo = d.seg_lo[0][g] - e0; n = d.seg_cnt[0][g]; }
  if (d.state[(it + 1) & 1].done) return;      // (uniform; behind the requests above so that it does not stand in front of them)
  if (has) s_th[threadIdx.x] = theta_from(d, theta[tt & ~SEG_TOP], d.psi_mode ? *d.log_norm : 0.0);
#pragma unroll
  for (int j = 0; j < L1_CHUNK / L1_TB; ++j) iv[j] = d.inv[c[j]];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < L1_CHUNK / L1_TB; ++j) {
    const uint32_t p = e0 + j * L1_TB + threadIdx.x;
    if (p < e1) {
      const double th = s_th[sg[j]]; double term = 0.0;
      if (iv[j] < 0.0) term = -iv[j];                                                   // single-transcript class: the full count
      else if (iv[j] != 0.0 && (!d.use_vbem || th > 0.0)) { const double v = th * w[j]; term = v * iv[j]; }
      s_term[p - e0] = term;
    }
  }
  __syncthreads();
  if (!has) return;
  double acc = 0.0;
  for (uint32_t i = 0; i < n; ++i) acc += s_term[lo + i];
  if (tt & SEG_TOP) alpha_out[tt & ~SEG_TOP] = acc; else d.part[0][g] = acc;
}
// the transcript a thread of k_fin4 / k_leaf0 owns: block 4 G + i, wave k, lane l -> level-1 group 64 G + i + 4 k (FinRec)
__device__ inline uint32_t fin4_txp() { const uint32_t G = blockIdx.x >> 2, i = blockIdx.x & 3, k = threadIdx.x >> 6; return ((64u * G + i + 4u * k) << 6) + (threadIdx.x & 63); }
// the block's quarter of its level-2 tree from the 16 wave sums, and its record
__device__ inline void fin4_record(const EmDev& d, double wave_sum, double rel, int bad) {
  __shared__ double s_v[16]; __shared__ double s_rel[16]; __shared__ int s_bad[16];
  for (int s = 32; s >= 1; s >>= 1) { const double o = __shfl_down(rel, s, 64); const int ob = __shfl_down(bad, s, 64); rel = o > rel ? o : rel; bad |= ob; }
  if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = wave_sum; s_rel[threadIdx.x >> 6] = rel; s_bad[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double v[16]; for (int k = 0; k < 16; ++k) { v[k] = s_v[k]; if (s_rel[k] > rel) rel = s_rel[k]; bad |= s_bad[k]; }
    for (int st = 8; st >= 1; st >>= 1) for (int k = 0; k < st; ++k) v[k] = v[k] + v[k + st];          // leaves j and j + 32, + 16, + 8, + 4 of the level-2 tree
    FinRec r; r.q = v[0]; r.relbad = (rel >= 0.0 ? (unsigned long long)__double_as_longlong(rel) : 0ULL) | (bad ? 1ULL << 63 : 0ULL);
    d.rec[blockIdx.x] = r;
  }
}
__global__ void __launch_bounds__(1024) k_fin4(EmDev d, const double* __restrict__ alpha, double* __restrict__ alpha_out, uint32_t it) {
  const uint32_t t = fin4_txp();
  double rel = -1.0; int bad = 0; double leaf = 0.0;
  uint32_t n2 = 0; bool empty = false; double a_old = 0.0, pr = 0.0; const double* src = nullptr;
  double a_new = 0.0;
  if (t < d.M) { empty = d.t_off[t + 1] == d.t_off[t]; n2 = d.l2_cnt ? d.l2_cnt[t] : 0; a_old = alpha[t]; pr = d.prior[t]; a_new = alpha_out[t]; if (n2) src = d.part[0] + d.l2_lo[t]; }
  if (d.state[(it + 1) & 1].done) return;
  if (t < d.M) {
    double acc;
    if (n2) {
      acc = 0.0;
      for (uint32_t i = 0; i < n2; i += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (i + k < n2) ? src[i + k] : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
      }
      alpha_out[t] = acc;
    } else if (empty) { acc = 0.0; alpha_out[t] = 0.0; }
    else acc = a_new;        // (what k_l13 stored: requested with everything else above)
    if (d.first_add != 0.0) { acc += d.first_add; alpha_out[t] = acc; }
    leaf = acc + pr;
    if (acc > 1e-2) {  // alphaCheckCutoff (:884)
      rel = fabs(a_old - acc) / acc;
      if (rel > d.tol) bad = 1;
    }
    if (d.psi_out) d.psi_out[t] = (leaf > 1e-10) ? sq_digamma(leaf) : -HUGE_VAL;   // digammaMin (:43): what k_theta computes from the same sum
  }
  fin4_record(d, wave_halving_sum(leaf), rel, bad);
}
// before the first iteration: the records (canonical sum of the initial alphas + prior) and psi of the initial alphas
__global__ void __launch_bounds__(1024) k_leaf0(EmDev d, const double* __restrict__ alpha, double* __restrict__ psi) {
  const uint32_t t = fin4_txp(); double leaf = 0.0;
  if (t < d.M) { leaf = alpha[t] + d.prior[t]; if (psi) psi[t] = (leaf > 1e-10) ? sq_digamma(leaf) : -HUGE_VAL; }
  fin4_record(d, wave_halving_sum(leaf), -1.0, 0);
}
__global__ void k_pack_cplan(uint32_t n, const uint32_t* __restrict__ cchunk, const uint64_t* __restrict__ off, uint4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { const uint32_t c0 = cchunk[i], c1 = cchunk[i + 1]; out[i] = make_uint4(c0, c1, (uint32_t)off[c0], (uint32_t)off[c1]); }
}
__global__ void k_pack_lplan(uint32_t n, const uint32_t* __restrict__ chunk_seg, const uint32_t* __restrict__ seg_lo, const uint8_t* __restrict__ seg_cnt, uint4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { const uint32_t s0 = chunk_seg[i], s1 = chunk_seg[i + 1]; out[i] = make_uint4(s0, s1, seg_lo[s0], seg_lo[s1 - 1] + seg_cnt[s1 - 1]); }
}
__global__ void k_copy_f64(uint32_t n, const double* __restrict__ src, double* __restrict__ dst) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i]; }

// Workspace arena: an EM session needs ~50 device buffers; hipMalloc + hipFree of each costs more (≈ 6 ms per
// session) than the 1.3 ms the device-side preparation takes.  While a session sets itself up its buffers are
// bump-allocated from a few large chunks that outlive it: a ctx keeps its arena across calls (sq_em_optimize(ctx, …),
// pre-sized by sq_ctx_reserve), a ctx-less call owns a private one.  A request that does not fit adds a chunk.
struct EmArena {
  struct Chunk { char* base; size_t cap, used; };
  std::vector<Chunk> chunks;
  // page-locked staging for the MB-sized host<->device copies of a session (slot 0: eff_len / alphas, slot 1: the block-plan
  // jump tables); pageable copies of 1-2 MB into fresh host pages were seen to take 3-17 ms
  void* pin[2] = {nullptr, nullptr}; size_t pin_cap[2] = {0, 0};
  ~EmArena() { for (auto& c : chunks) (void)hipFree(c.base); for (void* p : pin) if (p) (void)hipHostFree(p); }
  void* pinned(int slot, size_t bytes) {
    if (bytes <= pin_cap[slot]) return pin[slot];
    if (pin[slot]) (void)hipHostFree(pin[slot]); pin[slot] = nullptr; pin_cap[slot] = 0;
    if (hipHostMalloc(&pin[slot], bytes, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      pin[slot] = nullptr;
      return nullptr;
    }
    pin_cap[slot] = bytes; return pin[slot];
  }
  void reset() { for (auto& c : chunks) c.used = 0; }
  size_t capacity() const { size_t t = 0; for (auto& c : chunks) t += c.cap; return t; }
  int add_chunk(size_t bytes) {
    Chunk c{nullptr, bytes, 0};
    if (hipMalloc((void**)&c.base, bytes) != hipSuccess) return -1;
    if (getenv("SQ_POISON")) (void)hipMemset(c.base, 0xA5, bytes);   // tests: no reliance on zeroed fresh memory (hip/ctx.h)
    chunks.push_back(c);
    return 0;
  }
  void* take(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (auto& c : chunks) if (c.cap - c.used >= bytes) { void* p = c.base + c.used; c.used += bytes; return p; }
    if (add_chunk(std::max<size_t>(bytes, (size_t)64 << 20))) return nullptr;
    Chunk& c = chunks.back(); c.used = bytes; return c.base;
  }
};
thread_local EmArena* tl_arena = nullptr;   // set while an EmSession sets itself up

template <class T>
struct DBuf {
  T* p = nullptr; bool owned = false;
  ~DBuf() { if (p && owned) (void)hipFree(p); }
  int alloc(size_t n) {
    const size_t bytes = (n ? n : 1) * sizeof(T);
    if (tl_arena) { p = (T*)tl_arena->take(bytes); owned = false; return p ? 0 : -1; }
    owned = true; if (hipMalloc((void**)&p, bytes) != hipSuccess) return -1;
    if (getenv("SQ_POISON")) (void)hipMemset(p, 0xA5, bytes);
    return 0;
  }
  int upload(const std::vector<T>& v) {
    if (alloc(v.size())) return -1;
    return v.empty() || hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
  }
};

double canonical_sum_host(std::vector<double> x) {  // SPEC §D2 (host copy used for init / final sum)
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

// ---- problem preparation on the device (CollapsedEMOptimizer.cpp:760-873) ---------------------------
// combined weights, transcript-major CSC (radix sort of (tid, class) keys), the blocked-64 reduction
// plan (SPEC §D4) and the block plans of k_class / k_l1 are all built in HBM from the label-major
// CSR; the host only follows two short "next block" chains.
__global__ void k_prep_cw(uint32_t E, uint32_t M, const uint64_t* __restrict__ off, const uint32_t* __restrict__ tid,
    const double* __restrict__ w,
    const uint64_t* __restrict__ cnt_u,
                          const double* __restrict__ eff, int no_rich, int eq_mode, double* __restrict__ cw, double* __restrict__ cnt_f,
                              uint32_t* __restrict__ err) {
  uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; if (c >= E) return;
  const uint64_t a = off[c], b = off[c + 1]; const double cn = (double)cnt_u[c];
  if (cnt_f) cnt_f[c] = cn;
  double wsum = 0.0;
  for (uint64_t i = a; i < b; ++i) {
    const uint32_t t = tid[i]; if (t >= M) { atomicMax(err, t + 1); return; }
    double el = eff[t]; if (el <= 1.0) el = 1.0;                 // :841-844
    const double ww = no_rich ? 1.0 : w[i];                      // :845-848
    const double wt = eq_mode ? ww : cn * ww * (1.0 / el);       // :850-853
    cw[i] = wt; wsum += wt;
  }
  const double wn = 1.0 / wsum;
  for (uint64_t i = a; i < b; ++i) cw[i] = cw[i] * wn;
}
// markDegenerateClasses (CollapsedEMOptimizer.cpp:330-394): a class whose sum_i alpha0[tid_i] * combinedWeight_i (NaN terms skipped, class
// order) is <= minEQClassWeight is set invalid; the update rules then skip it (:197, :289) — the same as a zero count here
__global__ void k_mark_degenerate(uint32_t E, const uint64_t* __restrict__ off, const uint32_t* __restrict__ tid, double* __restrict__ cw,
                                  const double* __restrict__ alpha0, double* __restrict__ cnt_f, uint32_t* __restrict__ ndrop) {   // cw is written: the dropped class loses its (possibly NaN) weights
  uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; if (c >= E) return;
  double denom = 0.0;
  for (uint64_t i = off[c]; i < off[c + 1]; ++i) { const double v = alpha0[tid[i]] * cw[i]; if (!(v != v)) denom += v; }
  if (denom <= 2.2250738585072014e-308) {   // count 0 and weights 0: denom = 0 in every update -> inv = 0 -> no term (k_class, k_l1)
    cnt_f[c] = 0.0; for (uint64_t i = off[c]; i < off[c + 1]; ++i) cw[i] = 0.0; atomicAdd(ndrop, 1u);
  }
}
__global__ void k_zero_dropped(uint32_t E, const uint64_t* __restrict__ off, const double* __restrict__ cnt_f, double* __restrict__ cw) {
  uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; if (c >= E) return;
  if (cnt_f[c] == 0.0) for (uint64_t i = off[c]; i < off[c + 1]; ++i) cw[i] = 0.0;
}
__global__ void k_refresh_tcw(uint64_t L, const uint32_t* __restrict__ cscpos, const double* __restrict__ cw, double* __restrict__ t_cw) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < L) t_cw[i] = cw[cscpos[i]];
}
__global__ void k_prep_prior(uint32_t M, const double* __restrict__ eff, double vb_prior, int per_txp, double* __restrict__ prior) {   // populatePriorAlphas_ :82-99
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; if (t < M) prior[t] = per_txp ? vb_prior : vb_prior * eff[t];
}
__global__ void k_prep_keys(uint32_t E, const uint64_t* __restrict__ off, const uint32_t* __restrict__ tid,
    unsigned long long* __restrict__ key,
    uint32_t* __restrict__ val) {
  uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; if (c >= E) return;
  for (uint64_t i = off[c]; i < off[c + 1]; ++i) { key[i] = ((unsigned long long)tid[i] << 32) | c; val[i] = (uint32_t)i; }
}
__global__ void k_prep_csc(uint64_t L, uint32_t M, const unsigned long long* __restrict__ key, const uint32_t* __restrict__ val,
    const double* __restrict__ cw,
                           uint32_t* __restrict__ t_cls, double* __restrict__ t_cw, uint64_t* __restrict__ t_off) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i >= L) return;
  const unsigned long long k = key[i]; const uint32_t t = (uint32_t)(k >> 32);
  t_cls[i] = (uint32_t)k; t_cw[i] = cw[val[i]];
  const int64_t tp = i ? (int64_t)(key[i - 1] >> 32) : -1;
  for (int64_t x = tp + 1; x <= (int64_t)t; ++x) t_off[x] = i;          // transcripts (tp, t] start here
  if (i + 1 == L) for (uint64_t x = (uint64_t)t + 1; x <= M; ++x) t_off[x] = L;
}
// number of blocked-64 segments of every transcript at each level (0 where the level is not needed)
__global__ void k_plan_counts(uint32_t M, const uint64_t* __restrict__ t_off, uint32_t* __restrict__ ns0, uint32_t* __restrict__ ns1,
    uint32_t* __restrict__ ns2,
    uint32_t* __restrict__ ns3, uint32_t* __restrict__ err) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; if (t > M) return;
  if (t == M) { ns0[M] = ns1[M] = ns2[M] = ns3[M] = 0; return; }
  const uint64_t n = t_off[t + 1] - t_off[t];
  const uint32_t a = (uint32_t)((n + 63) / 64);
  const uint32_t b = a > 1 ? (a + 63) / 64 : 0, c = b > 1 ? (b + 63) / 64 : 0, dd = c > 1 ? (c + 63) / 64 : 0;
  if (dd > 1) atomicMax(err, 0xFFFFFFFFu);
  ns0[t] = a; ns1[t] = b; ns2[t] = c; ns3[t] = dd;
}
// segments of level `lvl` for transcript t: runs of 64 items of the previous level (CSC entries for level 0)
__global__ void k_plan_fill(int lvl, uint32_t M, const uint64_t* __restrict__ t_off, const uint32_t* __restrict__ ns_prev,
    const uint32_t* __restrict__ base_prev,
                            const uint32_t* __restrict__ ns, const uint32_t* __restrict__ base, uint32_t* __restrict__ seg_lo,
                                uint8_t* __restrict__ seg_cnt,
                                uint32_t* __restrict__ seg_txp) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; if (t >= M) return;
  const uint32_t k = ns[t]; if (!k) return;
  const uint32_t n = lvl == 0 ? (uint32_t)(t_off[t + 1] - t_off[t]) : ns_prev[t];
  const uint32_t lo = lvl == 0 ? (uint32_t)t_off[t] : base_prev[t];
  const uint32_t first = base[t];
  for (uint32_t j = 0; j < k; ++j) {
    seg_lo[first + j] = lo + 64 * j;
    seg_cnt[first + j] = (uint8_t)min(64u, n - 64 * j);
    seg_txp[first + j] = t | (k == 1 ? SEG_TOP : 0u);
  }
}
// greedy block packing as a jump table: nxt[g] = first unit of the block after the one starting at g
// (a block takes units while its entries stay <= cap and, optionally, its unit count <= maxu)
template <class T>
__global__ void k_next_block(uint32_t n, const T* __restrict__ lo /* [n+1], lo[n] = total */, uint32_t cap, uint32_t maxu,
    uint32_t* __restrict__ nxt) {
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; if (g >= n) return;
  const T lim = lo[g] + cap;
  uint32_t a = g + 1, b = n;           // largest k in [g+1, n] with lo[k] <= lim; at least g+1
  while (a < b) { uint32_t m = (a + b + 1) >> 1; if (lo[m] <= lim) a = m; else b = m - 1; }
  if (maxu && a > g + maxu) a = g + maxu;
  nxt[g] = a;
}
__global__ void k_plan_seg8(uint32_t S0, const uint32_t* __restrict__ chunk_seg, uint32_t nchunks, const uint32_t* __restrict__ seg_lo,
    const uint8_t* __restrict__ seg_cnt, uint8_t* __restrict__ t_seg8) {
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; if (g >= S0) return;
  uint32_t a = 0, b = nchunks - 1;     // last block with chunk_seg[block] <= g
  while (a < b) { uint32_t m = (a + b + 1) >> 1; if (chunk_seg[m] <= g) a = m; else b = m - 1; }
  const uint8_t idx = (uint8_t)(g - chunk_seg[a]); const uint32_t lo = seg_lo[g], n = seg_cnt[g];
  for (uint32_t i = 0; i < n; ++i) t_seg8[lo + i] = idx;
}
__global__ void k_plan_l2(uint32_t S1, const uint32_t* __restrict__ seg_lo1, const uint8_t* __restrict__ seg_cnt1,
    const uint32_t* __restrict__ seg_txp1,
    uint32_t* __restrict__ l2_lo, uint8_t* __restrict__ l2_cnt) {
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; if (g >= S1) return;
  const uint32_t t = seg_txp1[g] & ~SEG_TOP; l2_lo[t] = seg_lo1[g]; l2_cnt[t] = seg_cnt1[g];
}

// Runs the iteration loop. mode 0: optimise to convergence; mode 1: exactly `fixed_iters` steps.
// One uploaded problem (CSR + CSC + reduction plan); `run` can be called repeatedly, e.g. once per
// bootstrap replicate with resampled class counts (the combined weights stay those of the original
// counts, as in doBootstrap — CollapsedEMOptimizer.cpp:398-552).
struct EmSession {
  typedef sq_eq_dev_csr EqDevCsr;
  uint32_t M = 0, E = 0; uint64_t L = 0; uint32_t g1 = 0; const sq_em_opts* o = nullptr; EmDev d;
  DBuf<uint64_t> d_off, d_toff, d_cntu;
  DBuf<uint32_t> d_tid, d_tcls, d_flags;
  DBuf<double> d_w, d_eff, d_cw, d_cnt, d_tcw, d_prior, d_theta, d_inv, d_a0, d_a1, d_part;
  DBuf<unsigned long long> d_maxrel, d_log;
  DBuf<double> d_lognorm, d_psi0, d_psi1;
  // entries a block of the class pass / of level 1 stages through LDS.  Measured on MI355X (c2 table of 40 M pairs: 0.52 M classes, 1.33 M labels; tools/em_sweep.py): 4096 / 2048 /
  // 1024 / 512 entries per class block 32.3 / 28.3 / 26.6 / 26.7 us per iteration, per level-1 block 29.4 / 28.3 / 28.1 / 28.8: the kernels are chains of dependent requests, and
  // more, smaller blocks hide them better than longer unrolled ones
  static constexpr int cl_chunk = CL_CHUNK_5, l1_chunk = L1_CHUNK_5;
  DBuf<EmIterState> d_state; DBuf<FinRec> d_rec; DBuf<uint4> d_cplan, d_lplan; uint32_t nrec = 0;
  DBuf<uint32_t> d_slo[4], d_stx[4]; DBuf<uint8_t> d_scn[4]; DBuf<double> d_lpart[4];
  DBuf<uint32_t> d_chunk, d_l2lo, d_cchunk; DBuf<uint8_t> d_l2cnt, d_seg8;
  hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
  EmArena* arena = nullptr; EmArena own_arena;   // arena: borrowed from a ctx (set before setup), else own_arena
  bool own_stream = true;   // false: `st` was lent by a ctx (set before setup): no stream / hardware queue is created per session
  ~EmSession() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (hipEvent_t e : ev_look) if (e) (void)hipEventDestroy(e);
    if (st && own_stream) (void)hipStreamDestroy(st);
    if (arena && arena != &own_arena) arena->reset();
  }

  // eq: host table (uploaded) — or dv: a CSR that already lives on this device
  int setup(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* opts, const EqDevCsr* dv = nullptr) {
    if (!arena) arena = &own_arena;
    struct Scope { Scope(EmArena* a) { tl_arena = a; } ~Scope() { tl_arena = nullptr; } } scope(arena);   // every DBuf::alloc below draws from the arena
    return setup_impl(device, eq, txp, opts, dv);
  }
  int setup_impl(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* opts, const EqDevCsr* dv) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
      sq_set_error("no HIP device %d (found %d): EM has no CPU fallback", device, ndev);
      return SQ_ERR_DEVICE;
    }
    SQ_HIP_CHECK(hipSetDevice(device));
    o = opts;
    PhaseTimer pt("em.setup");
    const uint64_t E64 = dv ? dv->E : eq->num_classes; L = dv ? dv->L : eq->num_labels; M = txp->num_txp; g1 = (M + 63) / 64;
    if (E64 >= 0xFFFFFFFFull) { sq_set_error("too many equivalence classes"); return SQ_ERR_OVERFLOW; }
    if (L >= 0x7FFFFFFFull) { sq_set_error("too many label entries for the EM reduction plan"); return SQ_ERR_OVERFLOW; }
    E = (uint32_t)E64;
    if (!st) { SQ_HIP_CHECK(hipStreamCreate(&st)); own_stream = true; }
    SQ_HIP_CHECK(hipEventCreate(&e0)); SQ_HIP_CHECK(hipEventCreate(&e1));
    pt.mark("stream+events");
    const int TB = 256; auto nb = [&](uint64_t n) { return (uint32_t)((n + TB - 1) / TB); };
    // inputs
    const uint64_t* p_off; const uint32_t* p_tid; const double* p_w; const unsigned long long* p_cnt;
    if (dv) { p_off = dv->off; p_tid = dv->tid; p_w = dv->w; p_cnt = dv->cnt; }
    else {
      if (d_off.alloc((size_t)E + 1) || d_tid.alloc(L) || d_w.alloc(L) || d_cntu.alloc(E)) {
        sq_set_error("device allocation failed in EM (inputs)");
        return SQ_ERR_NOMEM;
      }
      SQ_HIP_CHECK(hipMemcpyAsync(d_off.p, eq->off, ((size_t)E + 1) * 8, hipMemcpyHostToDevice, st));
      SQ_HIP_CHECK(hipMemcpyAsync(d_tid.p, eq->tid, L * 4, hipMemcpyHostToDevice, st));
      SQ_HIP_CHECK(hipMemcpyAsync(d_w.p, eq->w, L * 8, hipMemcpyHostToDevice, st));
      SQ_HIP_CHECK(hipMemcpyAsync(d_cntu.p, eq->count, (size_t)E * 8, hipMemcpyHostToDevice, st));
      p_off = d_off.p; p_tid = d_tid.p; p_w = d_w.p; p_cnt = (const unsigned long long*)d_cntu.p;
    }
    p_off_ = p_off; p_tid_ = p_tid; p_w_ = p_w; p_cnt_ = p_cnt;
    DBuf<unsigned long long> key, key2; DBuf<uint32_t> val, val2, ns[4], base[4], d_err, nxt; DBuf<uint8_t> tmp;
    bool ok = !d_eff.alloc(M) && !d_cw.alloc(L) && !d_cnt.alloc(E) && !d_prior.alloc(M) && !d_toff.alloc((size_t)M + 1) &&
        !d_tcls.alloc(L) && !d_tcw.alloc(L) &&
        !key.alloc(L) && !key2.alloc(L) && !val.alloc(L) && !val2.alloc(L) && !d_err.alloc(1) &&
              !d_theta.alloc(M) && !d_inv.alloc(E) && !d_a0.alloc(M) && !d_a1.alloc(M) && !d_part.alloc((size_t)g1 * 3 + 512) &&
                  !d_flags.alloc(4) &&
                  !d_maxrel.alloc(1) && !d_log.alloc(1) && !d_lognorm.alloc(1) && !d_psi0.alloc(M) && !d_psi1.alloc(M);
    for (int l = 0; l < 4 && ok; ++l) ok = !ns[l].alloc((size_t)M + 1) && !base[l].alloc((size_t)M + 1);
    if (!ok) { sq_set_error("device allocation failed in EM (%s)", hipGetErrorString(hipGetLastError())); return SQ_ERR_NOMEM; }
    pt.mark("buffers");
    h_stage = (double*)arena->pinned(0, (size_t)3 * M * 8 + 64);   // [0,M) eff_len up, [M,2M) alphas up, [2M,3M) alphas down, then two look slots (run()); nullptr: plain pageable copies
    h_look = h_stage ? (EmIterState*)(h_stage + (size_t)3 * M) : nullptr;
    if (h_stage) {
      memcpy(h_stage, txp->eff_len, (size_t)M * 8);
      SQ_HIP_CHECK(hipMemcpyAsync(d_eff.p, h_stage, (size_t)M * 8, hipMemcpyHostToDevice, st));
    }
    else SQ_HIP_CHECK(hipMemcpyAsync(d_eff.p, txp->eff_len, (size_t)M * 8, hipMemcpyHostToDevice, st));
    SQ_HIP_CHECK(hipMemsetAsync(d_err.p, 0, 4, st)); SQ_HIP_CHECK(hipMemsetAsync(d_toff.p, 0, ((size_t)M + 1) * 8, st));
    pt.mark("alloc+upload");
    // [r2] the "first submission gap": without this drain the first synchronisation of the set-up (after the sort and the block plans, 0.9 ms
    // of device work in the kernel trace) returned 17-21 ms late when the stream had been idle since the export; with it the whole
    // set-up takes 2 ms (SQ_TIMING, MI355X / ROCm 7.2).  The upload is 1.3 MB from page-locked memory: the drain itself costs 0.03 ms.
    SQ_HIP_CHECK(sq_em_wait(st)); pt.mark("upload drained");
    // combined weights, prior, CSC
    if (E) k_prep_cw<<<nb(E), TB, 0, st>>>(E, M, p_off, p_tid, p_w, (const uint64_t*)p_cnt, d_eff.p, o->no_rich_eq_classes,
        o->eq_class_mode, d_cw.p, d_cnt.p, d_err.p);
    pt.mark("prep:first-launch-returned");
    k_prep_prior<<<nb(M), TB, 0, st>>>(M, d_eff.p, o->vb_prior, o->per_transcript_prior, d_prior.p);
    if (L) {
      k_prep_keys<<<nb(E), TB, 0, st>>>(E, p_off, p_tid, key.p, val.p);
      int tbits = 1; while ((1ull << tbits) < M) ++tbits;
      size_t tb = 0; (void)rocprim::radix_sort_pairs(nullptr, tb, key.p, key2.p, val.p, val2.p, (size_t)L, 0u, (unsigned)(32 + tbits), st);
      if (tmp.alloc(tb + 256)) { sq_set_error("device allocation failed in EM (sort)"); return SQ_ERR_NOMEM; }
      SQ_HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, tb, key.p, key2.p, val.p, val2.p, (size_t)L, 0u, (unsigned)(32 + tbits), st));
      k_prep_csc<<<nb(L), TB, 0, st>>>(L, M, key2.p, val2.p, d_cw.p, d_tcls.p, d_tcw.p, d_toff.p);
      if (keep_cscpos) { if (d_cscpos.alloc(L)) { sq_set_error("device allocation failed in EM (CSC positions)"); return SQ_ERR_NOMEM; }
        SQ_HIP_CHECK(hipMemcpyAsync(d_cscpos.p, val2.p, L * 4, hipMemcpyDeviceToDevice, st)); }
      pt.mark("prep:sort+csc-launched");
    }
    // blocked-64 plan: per-transcript segment counts, exclusive scans, fill
    k_plan_counts<<<nb((uint64_t)M + 1), TB, 0, st>>>(M, d_toff.p, ns[0].p, ns[1].p, ns[2].p, ns[3].p, d_err.p);
    { size_t tb = 0; (void)rocprim::exclusive_scan(nullptr, tb, ns[0].p, base[0].p, 0u, (size_t)M + 1, rocprim::plus<uint32_t>(), st);
      DBuf<uint8_t> stmp; if (stmp.alloc(tb + 256)) { sq_set_error("device allocation failed in EM (scan)"); return SQ_ERR_NOMEM; }
      for (int l = 0; l < 4; ++l) {
        size_t t2 = tb + 256;
        SQ_HIP_CHECK(rocprim::exclusive_scan(stmp.p, t2, ns[l].p, base[l].p, 0u, (size_t)M + 1, rocprim::plus<uint32_t>(), st));
      }
      uint32_t S[4] = {0, 0, 0, 0}, herr = 0;
      for (int l = 0; l < 4; ++l) SQ_HIP_CHECK(hipMemcpyAsync(&S[l], base[l].p + M, 4, hipMemcpyDeviceToHost, st));
      SQ_HIP_CHECK(hipMemcpyAsync(&herr, d_err.p, 4, hipMemcpyDeviceToHost, st));
      SQ_HIP_CHECK(sq_em_wait(st));
      pt.mark("prep:csc+plan-sizes");
      if (herr == 0xFFFFFFFFu) { sq_set_error("EM reduction plan deeper than 4 levels"); return SQ_ERR_OVERFLOW; }
      if (herr) { sq_set_error("eq-class label references transcript %u >= %u", herr - 1, M); return SQ_ERR_ARG; }
      d.nlevels = 0;
      for (int l = 0; l < 4; ++l) { d.nseg[l] = S[l]; if (S[l]) d.nlevels = l + 1; }
    }
    for (int l = 0; l < 4; ++l) {
      const size_t n = d.nseg[l];
      if (d_slo[l].alloc(n + 1) || d_stx[l].alloc(n + 1) || d_scn[l].alloc(n + 1) || d_lpart[l].alloc(n + 1)) {
        sq_set_error("device allocation failed in EM plan");
        return SQ_ERR_NOMEM;
      }
      if (n) k_plan_fill<<<nb(M), TB, 0, st>>>(l, M, d_toff.p, l ? ns[l - 1].p : nullptr, l ? base[l - 1].p : nullptr, ns[l].p, base[l].p,
          d_slo[l].p, d_scn[l].p,
          d_stx[l].p);
    }
    // block plans (greedy packing = following a jump table; the chain has ~L/2048 links)
    std::vector<uint32_t> h_chunk, h_cchunk;
    { const uint32_t S0 = d.nseg[0]; const uint32_t Lu = (uint32_t)L;
      SQ_HIP_CHECK(hipMemcpyAsync(d_slo[0].p + S0, &Lu, 4, hipMemcpyHostToDevice, st));   // sentinel: seg_lo[S0] = L
      if (nxt.alloc(std::max<size_t>(S0, E) + 1)) { sq_set_error("device allocation failed in EM plan"); return SQ_ERR_NOMEM; }
      std::vector<uint32_t> hn_pageable; uint32_t* hn = (uint32_t*)arena->pinned(1, (std::max<size_t>(S0, E) + 1) * 4);
      if (!hn) { hn_pageable.resize(std::max<size_t>(S0, E) + 1); hn = hn_pageable.data(); }
      if (S0) { k_next_block<uint32_t><<<nb(S0), TB, 0, st>>>(S0, d_slo[0].p, (uint32_t)l1_chunk, L1_TB, nxt.p);
        SQ_HIP_CHECK(hipMemcpyAsync(hn, nxt.p, (size_t)S0 * 4, hipMemcpyDeviceToHost, st)); SQ_HIP_CHECK(sq_em_wait(st));
        for (uint32_t g = 0; g < S0; g = hn[g]) h_chunk.push_back(g); h_chunk.push_back(S0); }
      pt.mark("prep:jump1");
      if (E) { k_next_block<uint64_t><<<nb(E), TB, 0, st>>>(E, p_off, (uint32_t)cl_chunk, (uint32_t)cl_chunk, nxt.p);
        SQ_HIP_CHECK(hipMemcpyAsync(hn, nxt.p, (size_t)E * 4, hipMemcpyDeviceToHost, st)); SQ_HIP_CHECK(sq_em_wait(st));
        for (uint32_t c = 0; c < E; c = hn[c]) h_cchunk.push_back(c); h_cchunk.push_back(E); }
      pt.mark("prep:jump2");
      if (d_chunk.upload(h_chunk) || d_cchunk.upload(h_cchunk) || d_seg8.alloc(L)) {
        sq_set_error("device allocation failed in EM plan");
        return SQ_ERR_NOMEM;
      }
      if (S0) k_plan_seg8<<<nb(S0), TB, 0, st>>>(S0, d_chunk.p, (uint32_t)h_chunk.size() - 1, d_slo[0].p, d_scn[0].p, d_seg8.p);
      // [r6] the block plans as one 16-byte record per block (first / end unit, first / end entry): a block's first request brings everything it needs to issue the rest
      const uint32_t ncc = h_cchunk.size() > 1 ? (uint32_t)h_cchunk.size() - 1 : 0, nlc = h_chunk.size() > 1 ? (uint32_t)h_chunk.size() - 1 : 0;
      nrec = 4 * ((g1 + 63) / 64);
      if (d_cplan.alloc(ncc + 1) || d_lplan.alloc(nlc + 1) || d_state.alloc(2) || d_rec.alloc(nrec + 4)) { sq_set_error("device allocation failed in EM plan"); return SQ_ERR_NOMEM; }
      if (ncc) k_pack_cplan<<<nb(ncc), TB, 0, st>>>(ncc, d_cchunk.p, p_off, d_cplan.p);
      if (nlc) k_pack_lplan<<<nb(nlc), TB, 0, st>>>(nlc, d_chunk.p, d_slo[0].p, d_scn[0].p, d_lplan.p);
    }
    const bool fold_l2 = d.nlevels == 2;
    if (fold_l2) {
      if (d_l2lo.alloc(M) || d_l2cnt.alloc(M)) { sq_set_error("device allocation failed in EM plan"); return SQ_ERR_NOMEM; }
      SQ_HIP_CHECK(hipMemsetAsync(d_l2cnt.p, 0, M, st)); SQ_HIP_CHECK(hipMemsetAsync(d_l2lo.p, 0, (size_t)M * 4, st));
      k_plan_l2<<<nb(d.nseg[1]), TB, 0, st>>>(d.nseg[1], d_slo[1].p, d_scn[1].p, d_stx[1].p, d_l2lo.p, d_l2cnt.p);
    }
    SQ_HIP_CHECK(sq_em_wait(st));
    d.M = M;
    d.E = E;
    d.L = L;
    d.off = p_off;
    d.tid = p_tid;
    d.cw = d_cw.p;
    d.cnt = d_cnt.p;
    d.t_off = d_toff.p;
    d.t_cls = d_tcls.p;
    d.t_cw = d_tcw.p;
    d.prior = d_prior.p;
    d.theta = d_theta.p;
    d.inv = d_inv.p;
    d.partial = d_part.p;
    d.flags = d_flags.p;
    d.maxrel = d_maxrel.p;
    d.tol = o->rel_diff_tolerance;
    d.use_vbem = o->use_vbem; d.first_add = 0.0;
    for (int l = 0; l < 4; ++l) {
      d.seg_lo[l] = d_slo[l].p;
      d.seg_cnt[l] = d_scn[l].p;
      d.seg_txp[l] = d_stx[l].p;
      d.part[l] = d_lpart[l].p;
    }
    d.cchunk = d_cchunk.p; d.ncchunks = h_cchunk.size() > 1 ? (uint32_t)h_cchunk.size() - 1 : 0;
    d.t_seg8 = d_seg8.p; d.chunk_seg = d_chunk.p; d.nchunks = h_chunk.size() > 1 ? (uint32_t)h_chunk.size() - 1 : 0;
    d.l2_lo = fold_l2 ? d_l2lo.p : nullptr; d.l2_cnt = fold_l2 ? d_l2cnt.p : nullptr;
    d.psi_mode = 0; d.psi = nullptr; d.log_norm = d_lognorm.p; d.psi_out = nullptr; d.log_norm_out = d_lognorm.p;
    d.state = d_state.p; d.rec = d_rec.p; d.nrec = nrec; d.cplan = d_cplan.p; d.lplan = d_lplan.p;
    SQ_HIP_CHECK(sq_em_wait(st));
    pt.mark("device-prepare");
    return SQ_OK;
  }

  // mode 0: optimise to convergence (min_iter / o->max_iter); mode 1: exactly `fixed_iters` steps.
  // alpha_dev != nullptr: the initial alphas are already in d_a0 (device); else they are uploaded from `alpha`.
  int run(std::vector<double>& alpha, int mode, uint32_t fixed_iters, uint32_t min_iter, sq_em_report* rep, bool alpha_on_device = false,
      bool fetch = true, uint32_t it0 = 0 /* iterations already done (the bias hook splits an optimisation in two runs) */,
      uint32_t max_iter_cap = 0 /* > 0: stop after this many iterations at the latest (the part before the bias hook) */) {
    const int TB = 256;
    if (!alpha_on_device) {
      if (h_stage) {
        memcpy(h_stage + M, alpha.data(), (size_t)M * 8);
        SQ_HIP_CHECK(hipMemcpyAsync(d_a0.p, h_stage + M, (size_t)M * 8, hipMemcpyHostToDevice, st));
      }
      else SQ_HIP_CHECK(hipMemcpyAsync(d_a0.p, alpha.data(), (size_t)M * 8, hipMemcpyHostToDevice, st));
    }
    SQ_HIP_CHECK(hipMemsetAsync(d_flags.p, 0, 4 * sizeof(uint32_t), st));
    if (it0 == 0) num_degenerate = 0;
    if (mark_degenerate && E && it0 == 0) {   // optimize() only: the initial alphas decide (flags[3] counts the dropped classes)
      k_mark_degenerate<<<(E + TB - 1) / TB, TB, 0, st>>>(E, d.off, d.tid, d_cw.p, d_a0.p, d_cnt.p, d_flags.p + 3);
      SQ_HIP_CHECK(hipMemcpyAsync(&num_degenerate, d_flags.p + 3, 4, hipMemcpyDeviceToHost, st));
    }
    SQ_HIP_CHECK(hipMemsetAsync(d_maxrel.p, 0, 8, st));
    SQ_HIP_CHECK(hipMemsetAsync(d_log.p, 0, 8, st));
    d.min_iter = (mode == 0) ? min_iter : 0xFFFFFFFFu;
    double* cur = d_a0.p; double* nxt = d_a1.p;
    // level-1 partials of (alpha + prior) live in d.partial (written by k_fin); k_top finishes the
    // levels above (<= 4096 partials); for M > 262144 k_sum_level kernels shrink the list first.
    double* part_lvl1 = d.partial; double* part_tmp = d.partial + g1 + 64;
    auto launch_top = [&](int close_prev, uint32_t prev_it) {
      const double* pin = part_lvl1; uint32_t n1 = g1;
      double* a = part_tmp; double* b = part_tmp + g1 / 64 + 64;
      while (n1 > 4096) {
        k_sum_level<<<(n1 + TB - 1) / TB, TB, 0, st>>>(pin, nullptr, n1, a);
        pin = a;
        n1 = (n1 + 63) / 64;
        std::swap(a, b);
      }
      k_top<<<1, 1024, 0, st>>>(d, pin, n1, close_prev, prev_it, d_log.p, d_lognorm.p);
    };
    // Measured on MI355X (c2, E = 0.62 M, L = 1.64 M, M = 191 k): five launches 40.9 us per iteration; [r4] four (k_theta folded into the gathers, k_top closing from per-block
    // results) 38.5; [r6] three — k_top's work as the prologue of the class pass, one 16-byte plan record per block, every request of a block issued before the first is waited
    // for, 1024-entry blocks — 29.4 us (profiles/r06_kernel_stats_c2_*.txt).  The five-launch form remains for plans with more than two reduction levels or M > 262 144.
    // [r6] three launches (k_top's work is the prologue of the class pass: the canonical sum reaches it as <= 256 records): the default under the same conditions
    const bool three = g1 <= 4096 && (d.l2_cnt || d.nlevels <= 1);
    if (three) SQ_HIP_CHECK(hipMemsetAsync(d_state.p, 0, 2 * sizeof(EmIterState), st));
    bool first3 = true;
    double* psi_cur = d_psi0.p; double* psi_nxt = d_psi1.p;
    const bool plus_one = mark_degenerate && !o->use_vbem;   // optimize() only (the replicates' alphasPrime start at zero: :413-414)
    auto launch_iter5 = [&](uint32_t it) {
      const double* theta_src = cur;
      if (o->use_vbem) { k_theta<<<(M + TB - 1) / TB, TB, 0, st>>>(d, cur, d_lognorm.p); theta_src = d.theta; }
      if (d.ncchunks) k_class<<<d.ncchunks, CL_TB, 0, st>>>(d, theta_src);
      if (d.nchunks) k_l1<<<d.nchunks, L1_TB, 0, st>>>(d, theta_src, nxt);
      if (!d.l2_cnt) { int l = 1; for (; l < d.nlevels && d.nseg[l] > 1024; ++l) k_level<<<(d.nseg[l] + TB - 1) / TB, TB, 0, st>>>(d, l,
          nxt);
        if (l < d.nlevels) k_upper<<<1, 1024, 0, st>>>(d, l, nxt); }
      { EmDev df = d; df.first_add = (plus_one && it == 0) ? 1.0 : 0.0; k_fin<<<(M + TB - 1) / TB, TB, 0, st>>>(df, cur, nxt, o->use_vbem ? part_lvl1 : nullptr); }
      // closes iteration `it`; VBEM: also logNorm for the next one
      if (o->use_vbem) launch_top(1, it);
      else k_close<<<1, 1, 0, st>>>(d, it, d_log.p);
      std::swap(cur, nxt);
    };
    auto launch_iter3 = [&](uint32_t it) {
      EmDev dd = d; dd.first_add = (plus_one && it == 0) ? 1.0 : 0.0;
      if (o->use_vbem) { dd.psi_mode = 1; dd.psi_out = psi_nxt; }
      const double* src = o->use_vbem ? psi_cur : cur;
      // (a problem without classes still runs the class pass's prologue: one block that only closes the iteration before)
      const int cp = first3 ? 0 : 1;
      if (dd.ncchunks) k_class3<CL_CHUNK_5><<<dd.ncchunks, CL_TB, 0, st>>>(dd, src, it, cp); else { EmDev de = dd; if (first3) de.min_iter = 0xFFFFFFFFu; k_close3<<<1, 64, 0, st>>>(de, it); }
      if (dd.nchunks) k_l13<L1_CHUNK_5><<<dd.nchunks, L1_TB, 0, st>>>(dd, src, nxt, it);
      k_fin4<<<nrec, 1024, 0, st>>>(dd, cur, nxt, it);
      first3 = false;
      std::swap(cur, nxt); std::swap(psi_cur, psi_nxt);
    };
    auto launch_iter = [&](uint32_t it) { if (three) launch_iter3(it); else launch_iter5(it); };
    if (three) k_leaf0<<<nrec, 1024, 0, st>>>(d, cur, o->use_vbem ? psi_cur : nullptr);
    else if (o->use_vbem) { k_sum_level<<<(M + TB - 1) / TB, TB, 0, st>>>(cur, d.prior, M, part_lvl1); launch_top(0, 0); }
    uint32_t it = it0, executed = 0; uint32_t done = 0; uint32_t hflags[4] = {0, 0, 0, 0}; EmIterState hstate = {0, 0, 0};
    SQ_HIP_CHECK(hipEventRecord(e0, st));
    if (mode == 1) {
      for (; it < it0 + fixed_iters; ++it) launch_iter(it);
      executed = it0 + fixed_iters;
      if (three && fixed_iters) { EmDev de = d; de.min_iter = 0xFFFFFFFFu; k_close3<<<1, 64, 0, st>>>(de, it); SQ_HIP_CHECK(hipMemcpyAsync(&hstate, d_state.p + ((it + 1) & 1), sizeof(hstate), hipMemcpyDeviceToHost, st)); }
    } else {
      const uint32_t maxIter = max_iter_cap ? max_iter_cap : o->max_iter, minIter = min_iter;
      // run to min_iter without looking, then in chunks; kernels of iterations after convergence are no-ops
      // [r6] three-launch form with page-locked look slots: the host stays ONE CHUNK AHEAD of the look — chunk k + 1 is queued before the host waits for what chunk k left
      // (an event behind the copy of its closing state), so the stream never drains inside the loop; a chunk queued past convergence is 192 no-op launches
      if (three && h_look) {
        if (!ev_look[0]) { SQ_HIP_CHECK(hipEventCreateWithFlags(&ev_look[0], hipEventDisableTiming)); SQ_HIP_CHECK(hipEventCreateWithFlags(&ev_look[1], hipEventDisableTiming)); }
        int pend[2]; int np = 0, nxt_slot = 0;
        const uint32_t lim = std::max(maxIter, minIter);
        for (;;) {
          if (it < lim && np < 2) {
            uint32_t chunk = (it < minIter) ? (minIter - it) : 64; if (it + chunk > lim) chunk = lim - it;
            for (uint32_t j = 0; j < chunk; ++j, ++it) launch_iter(it);
            k_close3<<<1, 64, 0, st>>>(d, it);
            SQ_HIP_CHECK(hipMemcpyAsync(h_look + nxt_slot, d_state.p + ((it + 1) & 1), sizeof(EmIterState), hipMemcpyDeviceToHost, st));
            SQ_HIP_CHECK(hipEventRecord(ev_look[nxt_slot], st));
            pend[np++] = nxt_slot; nxt_slot ^= 1;
            continue;
          }
          if (!np) break;
          SQ_HIP_CHECK(hipEventSynchronize(ev_look[pend[0]]));
          hstate = h_look[pend[0]]; pend[0] = pend[1]; --np;
          if (hstate.done) { done = hstate.done; break; }
        }
      } else
      while (it < maxIter || it < minIter) {
        uint32_t chunk = (it < minIter) ? (minIter - it) : 64;   // a look costs a stream drain (~25 us); an iteration queued past convergence is five no-op launches
        // [r2] The loop is bound by the DEVICE: 44 us of kernels + five ~2 us hand-overs per iteration (the host fills the hardware queue and
        // then blocks in the launch calls).  Tried on MI355X and dropped, each at an unchanged 55 us per iteration: replaying pairs of
        // iterations as a HIP graph (device-side iteration counter); folding k_top into k_theta (every block finishing the canonical sum).
        uint32_t lim = std::max(maxIter, minIter);
        if (it + chunk > lim) chunk = lim - it;
        for (uint32_t j = 0; j < chunk; ++j, ++it) launch_iter(it);
        if (three) {   // the last iteration of the chunk is closed by the class pass that follows it: do that part now, then look at the slot it wrote
          k_close3<<<1, 64, 0, st>>>(d, it);
          SQ_HIP_CHECK(hipMemcpyAsync(&hstate, d_state.p + ((it + 1) & 1), sizeof(hstate), hipMemcpyDeviceToHost, st));
          SQ_HIP_CHECK(sq_em_wait(st));
          if (hstate.done) { done = hstate.done; break; }
          continue;
        }
        SQ_HIP_CHECK(hipMemcpyAsync(hflags, d_flags.p, sizeof(hflags), hipMemcpyDeviceToHost, st));
        SQ_HIP_CHECK(sq_em_wait(st));
        if (hflags[0]) { done = hflags[0]; break; }
      }
      executed = done ? done : it;
    }
    SQ_HIP_CHECK(hipEventRecord(e1, st)); SQ_HIP_CHECK(sq_em_wait(st));
    float ms = 0; SQ_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    result_dev = ((executed - it0) % 2 == 0) ? d_a0.p : d_a1.p;   // after `executed - it0` swaps starting from d_a0
    if (fetch) {
      if (h_stage) {
        SQ_HIP_CHECK(hipMemcpyAsync(h_stage + 2 * (size_t)M, result_dev, (size_t)M * 8, hipMemcpyDeviceToHost, st));
        SQ_HIP_CHECK(sq_em_wait(st));
        memcpy(alpha.data(), h_stage + 2 * (size_t)M, (size_t)M * 8);
      }
      else SQ_HIP_CHECK(hipMemcpy(alpha.data(), result_dev, (size_t)M * 8, hipMemcpyDeviceToHost));
    }
    unsigned long long mr = 0; if (three) mr = hstate.maxrel; else SQ_HIP_CHECK(hipMemcpy(&mr, d_log.p, 8, hipMemcpyDeviceToHost));
    if (rep) {
      rep->iters = executed; rep->converged = (mode == 0) ? (done != 0) : 0; double mrd; memcpy(&mrd, &mr, 8); rep->max_rel_diff = mrd;
      rep->device_ms = ms; rep->ms_per_iter = executed > it0 ? ms / (double)(executed - it0) : 0.0; rep->alpha_sum = 0; rep->num_degenerate = num_degenerate; rep->_pad = 0;
    }
    return SQ_OK;
  }
  double* result_dev = nullptr; double* h_stage = nullptr; EmIterState* h_look = nullptr; hipEvent_t ev_look[2] = {nullptr, nullptr};
  bool mark_degenerate = false, keep_cscpos = false; uint32_t num_degenerate = 0;
  // updateEqClassWeights + populatePriorAlphas_ after the bias hook (CollapsedEMOptimizer.cpp:160-176, 906-916): new effective lengths ->
  // combined weights (classes dropped as degenerate stay dropped), their CSC copies, priors
  const uint64_t* p_off_ = nullptr; const uint32_t* p_tid_ = nullptr; const double* p_w_ = nullptr; const unsigned long long* p_cnt_ = nullptr;
  DBuf<uint32_t> d_cscpos;
  int refresh_weights(const double* eff_host) {
    const int TB = 256;
    SQ_HIP_CHECK(hipMemcpyAsync(d_eff.p, eff_host, (size_t)M * 8, hipMemcpyHostToDevice, st));
    if (E) k_prep_cw<<<(E + TB - 1) / TB, TB, 0, st>>>(E, M, p_off_, p_tid_, p_w_, (const uint64_t*)p_cnt_, d_eff.p, o->no_rich_eq_classes, o->eq_class_mode, d_cw.p, nullptr, d_flags.p + 3);
    if (E) k_zero_dropped<<<(E + TB - 1) / TB, TB, 0, st>>>(E, p_off_, d_cnt.p, d_cw.p);
    k_prep_prior<<<(M + TB - 1) / TB, TB, 0, st>>>(M, d_eff.p, o->vb_prior, o->per_transcript_prior, d_prior.p);
    if (L) k_refresh_tcw<<<(uint32_t)((L + TB - 1) / TB), TB, 0, st>>>(L, d_cscpos.p, d_cw.p, d_tcw.p);
    SQ_HIP_CHECK(sq_em_wait(st));
    return SQ_OK;
  }
};

int run_em(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, std::vector<double>& alpha, int mode,
    uint32_t fixed_iters, sq_em_report* rep,
    const sq_eq_dev_csr* dv = nullptr, EmArena* arena = nullptr, hipStream_t lent = nullptr, bool mark_degenerate = false) {
  PhaseTimer pt("em");
  int rc;
  { EmSession S; S.arena = arena; S.mark_degenerate = mark_degenerate; if (lent) { S.st = lent; S.own_stream = false; } rc = S.setup(device, eq, txp, o, dv); if (rc) return rc;
    pt.mark("setup");
    rc = S.run(alpha, mode, fixed_iters, o->min_iter, rep);
    pt.mark("run"); }
  pt.mark("teardown");
  return rc;
}

// ---- a16 bootstrap (doBootstrap, CollapsedEMOptimizer.cpp:398-552) ----------------------------------
// multinomial resample of the class counts: draw i of replicate b picks the class whose cumulative
// count interval contains mulhi(r64(seed, b, i), total)
__global__ void k_bs_sample(uint64_t total, uint32_t E, const uint64_t* __restrict__ cum, uint64_t seed, uint64_t rep,
    unsigned long long* __restrict__ samp) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t idx = sq_mulhi64(sq_r64(seed, rep, i), total);
    uint32_t lo = 0, hi = E;  // first c with cum[c] > idx
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (cum[mid] > idx) hi = mid; else lo = mid + 1; }
    atomicAdd(&samp[lo], 1ULL);
  }
}
__global__ void k_u64_to_f64(uint32_t n, const unsigned long long* __restrict__ in, double* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = (double)in[i];
}

// ---- a17 Gibbs (sampleRoundNonCollapsedMultithreaded_, CollapsedGibbsSampler.cpp:92-278) --------------
struct GibbsDev { uint32_t M,
    E;
    const uint64_t* off;
    const uint32_t* tid;
    const double* w;
    const uint64_t* cnt;
    const double* eff;
    const double* prior;
    const uint8_t* active;
                  double* mu; double* count_f; unsigned long long* count_i; const uint64_t* draw_off; };
// [r3] the counts of the round before arrive as integers (count_i, filled by the item kernels' atomics) and are cleared here for this round's
// draws — the memset and the integer-to-double pass in between are gone; from_f = 1: the chain (re)starts from count_f (the initial counts)
__global__ void k_gibbs_mu(GibbsDev g, uint64_t seed, uint64_t round_key, int no_gamma, int from_f) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= g.M) return;
  const double cprev = from_f ? g.count_f[i] : (double)g.count_i[i];
  g.count_i[i] = 0;
  if (!g.active[i]) { g.mu[i] = 0.0; return; }
  double ci = cprev + g.prior[i];
  g.mu[i] = no_gamma ? ci / g.eff[i] : sq_gamma_draw(ci, 1.0 / (0.1 + g.eff[i]), seed, round_key, (uint64_t)i);   // beta = 0.1 (:104)
}
__device__ inline double gibbs_class_p(const GibbsDev& g, uint64_t a, uint32_t n, int mode, uint32_t i) {
  uint32_t t = g.tid[a + i];
  return mode == 0 ? (1000.0 * g.mu[t]) * g.w[a + i] : (mode == 1 ? 1.0 / g.eff[t] : 1.0);
}
// [r3] per round and class: the running sums acc_i = p_0 + .. + p_i in label order (the order the draw's scan adds them in) and
// their total; minEQClassWeight fallbacks as in the reference (:224-250).  A draw then picks the first i with u < acc_i.
__global__ void k_gibbs_prep(GibbsDev g, double* __restrict__ cum, double* __restrict__ denom) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; if (c >= g.E) return;
  const uint64_t a = g.off[c]; const uint32_t n = (uint32_t)(g.off[c + 1] - a);
  if (n < 2) { denom[c] = 0.0; return; }
  int mode = 0; double d = 0.0;
  for (uint32_t i = 0; i < n; ++i) { d += gibbs_class_p(g, a, n, 0, i); cum[a + i] = d; }
  if (d <= 2.2250738585072014e-308) { mode = 1; d = 0.0; for (uint32_t i = 0; i < n; ++i) { d += gibbs_class_p(g, a, n, 1, i); cum[a + i] = d; }
    if (d <= 2.2250738585072014e-308) { mode = 2; double acc = 0.0; for (uint32_t i = 0; i < n; ++i) { acc += 1.0; cum[a + i] = acc; } d = (double)n; } }
  (void)mode;
  denom[c] = d;
}
// An item = up to 256 consecutive draws of one class.  Classes of at most NMAX labels: the NMAX - 1 thresholds live in registers and a
// draw is its random number against all of them (no dependent scan); the item's counts leave as one atomic per label instead of one
// per draw — the sums are integers, so the result is the same whatever the order.  pick = #{i < n-1 : !(u < acc_i)}: the running
// sums never decrease, so that is the first i with u < acc_i (n - 1 when there is none).
template <int NMAX>
__device__ inline void gibbs_items_reg(const GibbsDev& g, uint32_t it, uint32_t nitems, const uint32_t* __restrict__ item_cls, const uint32_t* __restrict__ item_s0,
    uint64_t seed, uint64_t round_key, const double* __restrict__ cum, const double* __restrict__ denom) {
  if (it >= nitems) return;
  const uint32_t c = item_cls[it];
  const uint64_t a = g.off[c];
  const uint32_t n = (uint32_t)(g.off[c + 1] - a);
  const uint64_t cnt = g.cnt[c];
  if (n == 1) { if (item_s0[it] == 0) atomicAdd(&g.count_i[g.tid[a]], (unsigned long long)cnt); return; }
  double cu[NMAX - 1]; uint32_t G[NMAX - 1];
#pragma unroll
  for (int i = 0; i < NMAX - 1; ++i) { cu[i] = ((uint32_t)i < n - 1) ? cum[a + i] : __longlong_as_double(0x7FF0000000000000LL); G[i] = 0; }
  const double dn = denom[c];
  const uint64_t s0 = item_s0[it], s1 = min(s0 + 256, cnt), base = g.draw_off[c];
  for (uint64_t sidx = s0; sidx < s1; ++sidx) {
    const double u = sq_u01(sq_r64(seed ^ 0xC1A55ULL, round_key, base + sidx)) * dn;
#pragma unroll
    for (int i = 0; i < NMAX - 1; ++i) G[i] += (u < cu[i]) ? 0u : 1u;
  }
  const uint32_t D = (uint32_t)(s1 - s0);
  uint32_t prev = D;
#pragma unroll
  for (int i = 0; i < NMAX; ++i) {
    if ((uint32_t)i < n) {
      const uint32_t ge = ((uint32_t)i < n - 1) ? G[i < NMAX - 1 ? i : 0] : 0u;   // draws that passed threshold i
      const uint32_t k = prev - ge; prev = ge;
      if (k) atomicAdd(&g.count_i[g.tid[a + i]], (unsigned long long)k);
    }
  }
}
// larger classes: binary search of the running sums per draw
// [r3] a WAVE per item: lane l takes the draws s0 + l, s0 + l + 64, ... — a thread walking its item's 256 draws through four dependent loads
// each was the long pole of a round whenever this list was short (23 items took as long as the 639 000 of the register path)
__device__ inline void gibbs_items_big(const GibbsDev& g, uint32_t it, uint32_t lane, uint32_t nitems, const uint32_t* __restrict__ item_cls, const uint32_t* __restrict__ item_s0,
    uint64_t seed, uint64_t round_key, const double* __restrict__ cum, const double* __restrict__ denom) {
  if (it >= nitems) return;
  const uint32_t c = item_cls[it];
  const uint64_t a = g.off[c];
  const uint32_t n = (uint32_t)(g.off[c + 1] - a);
  const uint64_t cnt = g.cnt[c];
  const double dn = denom[c];
  const uint64_t s0 = item_s0[it], s1 = min(s0 + 256, cnt), base = g.draw_off[c];
  for (uint64_t sidx = s0 + lane; sidx < s1; sidx += 64) {
    const double u = sq_u01(sq_r64(seed ^ 0xC1A55ULL, round_key, base + sidx)) * dn;
    uint32_t lo = 0, hi = n - 1;                       // first i in [0, n-1) with u < cum[i], else n - 1
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (u < cum[a + mid]) hi = mid; else lo = mid + 1; }
    atomicAdd(&g.count_i[g.tid[a + lo]], 1ULL);
  }
}
// [r3] the item lists of a round in as few launches as their sizes allow: blocks [0, nb0) take the items of at most 8 labels (thresholds in
// registers), the blocks behind them the items [rest0, rest0 + nrest) by binary search — the large classes and, when there are only a few of
// them, the 9..16-label ones too (either way a draw picks the first label whose running sum exceeds it; the binary search needs few
// registers, so the common path keeps its occupancy).  Seven stream operations per round became three or four.
__global__ void __launch_bounds__(256) k_gibbs_items(GibbsDev g, uint32_t n0, uint32_t nb0, uint32_t rest0, uint32_t nrest, const uint32_t* __restrict__ item_cls,
    const uint32_t* __restrict__ item_s0, uint64_t seed, uint64_t round_key, const double* __restrict__ cum, const double* __restrict__ denom) {
  const uint32_t b = blockIdx.x;
  if (b < nb0) gibbs_items_reg<8>(g, b * 256u + threadIdx.x, n0, item_cls, item_s0, seed, round_key, cum, denom);
  else gibbs_items_big(g, (b - nb0) * 4u + (threadIdx.x >> 6), threadIdx.x & 63u, nrest, item_cls + rest0, item_s0 + rest0, seed, round_key, cum, denom);
}
__global__ void __launch_bounds__(256) k_gibbs_items16(GibbsDev g, uint32_t n1, const uint32_t* __restrict__ item_cls, const uint32_t* __restrict__ item_s0, uint64_t seed,
    uint64_t round_key, const double* __restrict__ cum, const double* __restrict__ denom) {
  gibbs_items_reg<16>(g, blockIdx.x * 256u + threadIdx.x, n1, item_cls, item_s0, seed, round_key, cum, denom);
}
__global__ void k_gibbs_alpha(GibbsDev g, double scale, double* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= g.M) return;
  double v = (g.mu[i] * g.eff[i]) * scale; out[i] = v > 1e-8 ? v : 0.0;
}
__global__ void k_mul(uint32_t n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] * b[i];
}

}  // namespace

extern "C" int sq_em_optimize_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, double* alpha_out,
    sq_em_report* rep) {
  if (!eq || !txp || !o || !alpha_out || !eq->off || !eq->tid || !eq->w || !eq->count || !txp->eff_len) {
    sq_set_error("sq_em_optimize_dev: bad arguments");
    return SQ_ERR_ARG;
  }
  return sq_em_optimize_impl(device, eq, nullptr, txp, o, alpha_out, rep, nullptr, nullptr);
}

// eq: host table, or dv: the label-major CSR already resident on `device` (the ctx's staged export)
// a ctx's persistent EM workspace (opaque to the other translation units)
int sq_em_arena_reserve(void** slot, size_t bytes, size_t pinned_bytes, size_t pinned_plan_bytes) {
  if (!*slot) *slot = new EmArena();
  EmArena* a = (EmArena*)*slot;
  (void)a->pinned(0, pinned_bytes); (void)a->pinned(1, pinned_plan_bytes);
  if (a->capacity() >= bytes) return SQ_OK;
  if (a->add_chunk(bytes - a->capacity() + ((size_t)8 << 20))) {
    sq_set_error("device allocation failed (EM workspace, %zu bytes)", bytes);
    return SQ_ERR_NOMEM;
  }
  return SQ_OK;
}
void sq_em_arena_free(void* slot) { delete (EmArena*)slot; }
// the allocations of EmSession::setup_impl, rounded up
size_t sq_em_workspace_bytes(uint64_t E, uint64_t L, uint64_t M) {
  return (size_t)(46 * L + 17 * (L / 64 + M) + 24 * E + 128 * M + ((size_t)16 << 20));
}

int sq_em_optimize_impl(int device, const sq_eq_table* eq, const sq_eq_dev_csr* dv, const sq_txp_in* txp, const sq_em_opts* o,
    double* alpha_out, sq_em_report* rep,
    void** arena_slot, void* lent_stream) {
  if (arena_slot && !*arena_slot) *arena_slot = new EmArena();
  const uint32_t M = txp->num_txp;
  // initial alphas (CollapsedEMOptimizer.cpp:778-823)
  std::vector<double> pc(M, 0.0); if (txp->projected_counts) pc.assign(txp->projected_counts, txp->projected_counts + M);
  double totalWeight = canonical_sum_host(pc);
  double uniformPrior = totalWeight / (double)M;
  double fracObserved = std::min(0.999, totalWeight / o->num_required_fragments);
  std::vector<double> alpha(M);
  const bool alt = o->alt_init_mode && txp->unique_count;   // metaGenomeMode or altInitMode (:817-818)
  for (uint32_t i = 0; i < M; ++i) {
    const double uni = alt ? ((double)txp->unique_count[i] + 0.5) * 1e-3 * txp->eff_len[i] : uniformPrior;   // alphasPrime (:790-792)
    alpha[i] = o->init_uniform ? 100.0 : (pc[i] * fracObserved + uni * (1.0 - fracObserved));
  }
  int rc = run_em(device, eq, txp, o, alpha, 0, 0, rep, dv, arena_slot ? (EmArena*)*arena_slot : nullptr, (hipStream_t)lent_stream, true);
  if (rc) return rc;
  for (uint32_t i = 0; i < M; ++i) if (alpha[i] <= 1e-8) alpha[i] = 0.0;  // truncateCountVector (:64-76), minAlpha 1e-8
  double asum = canonical_sum_host(alpha);
  for (uint32_t i = 0; i < M; ++i) alpha_out[i] = alpha[i];
  if (rep) rep->alpha_sum = asum;
  // :1016-1020
  if (asum < 2.2250738585072014e-308) {
    sq_set_error("Total alpha weight was too small! Make sure you ran salmon correctly.");
    return SQ_ERR_STATE;
  }
  return SQ_OK;
}

// optimize() with the bias hook (CollapsedEMOptimizer.cpp:901-928): 11 updates, updateEffectiveLengths through `cb`, new priors and class
// weights, then on to convergence (iteration counts and minIter = 100 run over both parts)
int sq_em_optimize_bias_impl(int device, const sq_eq_table* eq, const sq_eq_dev_csr* dv, const sq_txp_in* txp, const sq_em_opts* o, sq_efflen_cb cb, void* user,
    double* alpha_out, double* eff_len_out, sq_em_report* rep, void** arena_slot, void* lent_stream) {
  if (!txp || !o || !alpha_out || !txp->eff_len || !cb) { sq_set_error("sq_em_optimize_bias: bad arguments"); return SQ_ERR_ARG; }
  if (arena_slot && !*arena_slot) *arena_slot = new EmArena();
  const uint32_t M = txp->num_txp;
  std::vector<double> pc(M, 0.0); if (txp->projected_counts) pc.assign(txp->projected_counts, txp->projected_counts + M);
  const double totalWeight = canonical_sum_host(pc), uniformPrior = totalWeight / (double)M, fracObserved = std::min(0.999, totalWeight / o->num_required_fragments);
  const bool alt = o->alt_init_mode && txp->unique_count;
  std::vector<double> alpha(M), eff(txp->eff_len, txp->eff_len + M), eff2(M);
  for (uint32_t i = 0; i < M; ++i) { const double uni = alt ? ((double)txp->unique_count[i] + 0.5) * 1e-3 * txp->eff_len[i] : uniformPrior; alpha[i] = o->init_uniform ? 100.0 : (pc[i] * fracObserved + uni * (1.0 - fracObserved)); }
  EmSession S; S.arena = arena_slot ? (EmArena*)*arena_slot : nullptr; S.mark_degenerate = true; S.keep_cscpos = true;
  if (lent_stream) { S.st = (hipStream_t)lent_stream; S.own_stream = false; }
  int rc = S.setup(device, eq, txp, o, dv); if (rc) return rc;
  sq_em_report r1{}, r2{};
  // `needBias and (itNum > targetIt or converged)` (CollapsedEMOptimizer.cpp:901): the hook fires after 11 updates, or earlier at the first
  // update after which the convergence test holds (it is evaluated every iteration, whatever minIter says)
  const uint32_t HOOK_AT = 11;
  rc = S.run(alpha, 0, 0, 0, &r1, false, true, 0, HOOK_AT); if (rc) return rc;
  const uint32_t it1 = r1.iters;
  const uint32_t ndeg = S.num_degenerate;
  if (cb(alpha.data(), eff.data(), eff2.data(), M, user)) { sq_set_error("sq_em_optimize_bias: the effective-length callback failed"); return SQ_ERR_STATE; }
  rc = S.refresh_weights(eff2.data()); if (rc) return rc;
  rc = S.run(alpha, 0, 0, o->min_iter, &r2, false, true, it1); if (rc) return rc;
  for (uint32_t i = 0; i < M; ++i) if (alpha[i] <= 1e-8) alpha[i] = 0.0;
  const double asum = canonical_sum_host(alpha);
  memcpy(alpha_out, alpha.data(), (size_t)M * 8);
  if (eff_len_out) memcpy(eff_len_out, eff2.data(), (size_t)M * 8);
  if (rep) { *rep = r2; rep->alpha_sum = asum; rep->device_ms = r1.device_ms + r2.device_ms; rep->num_degenerate = ndeg; }
  if (asum < 2.2250738585072014e-308) { sq_set_error("Total alpha weight was too small! Make sure you ran salmon correctly."); return SQ_ERR_STATE; }
  return SQ_OK;
}

extern "C" int sq_em_steps_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, const double* alpha_in,
    uint32_t iters, double* alpha_out,
    sq_em_report* rep) {
  if (!eq || !txp || !o || !alpha_in || !alpha_out) { sq_set_error("sq_em_steps_dev: bad arguments"); return SQ_ERR_ARG; }
  std::vector<double> alpha(alpha_in, alpha_in + txp->num_txp);
  int rc = run_em(device, eq, txp, o, alpha, 1, iters, rep);
  if (rc) return rc;
  memcpy(alpha_out, alpha.data(), (size_t)txp->num_txp * 8);
  return SQ_OK;
}


extern "C" int sq_bootstrap_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, uint32_t B, uint64_t seed,
    uint64_t num_mapped,
    sq_replicate_cb cb, void* user) {
  return sq_bootstrap_range_dev(device, eq, txp, o, B, 0, B, seed, num_mapped, cb, user);
}
// replicates [first, first + count) of B: replicate b draws from the counter RNG keyed (seed, b), so a replicate is the same
// whichever GPU computes it — the ranks of a multi-GPU job take disjoint ranges (replicates -> GPUs)
extern "C" int sq_bootstrap_range_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, uint32_t B, uint32_t first,
    uint32_t count, uint64_t seed, uint64_t num_mapped, sq_replicate_cb cb, void* user) {
  if (first > B || count > B - first) { sq_set_error("sq_bootstrap_range_dev: range [%u, %u) outside %u replicates", first, first + count, B); return SQ_ERR_ARG; }
  if (count == 0) return SQ_OK;
  if (!eq || !txp || !o || !cb || !eq->off || !eq->tid || !eq->w || !eq->count || !txp->eff_len) {
    sq_set_error("sq_bootstrap_dev: bad arguments");
    return SQ_ERR_ARG;
  }
  EmSession S; int rc = S.setup(device, eq, txp, o); if (rc) return rc;
  const uint32_t M = S.M, E = S.E;
  std::vector<uint64_t> cum(E); uint64_t total = 0; for (uint32_t c = 0; c < E; ++c) { total += eq->count[c]; cum[c] = total; }
  std::vector<uint8_t> active(M, 0); for (uint64_t i = 0; i < S.L; ++i) active[eq->tid[i]] = 1;
  uint32_t nact = 0; for (auto a : active) nact += a;
  // :598-602
  if (nact == 0 || total == 0) {
    sq_set_error("It seems that no transcripts are expressed; something is likely wrong!");
    return SQ_ERR_STATE;
  }
  const double scale = 1.0 / (double)nact, totalNumFrags = (double)num_mapped;
  std::vector<double> init(M), alpha(M); for (uint32_t i = 0; i < M; ++i) init[i] = active[i] ? scale * totalNumFrags : 0.0;   // :605-606
  DBuf<uint64_t> d_cum;
  DBuf<unsigned long long> d_samp;
  if (d_cum.upload(cum) || d_samp.alloc(E)) {
    sq_set_error("device allocation failed (bootstrap)");
    return SQ_ERR_NOMEM;
  }
  const int TB = 256;
  for (uint32_t b = first; b < first + count; ++b) {
    SQ_HIP_CHECK(hipMemsetAsync(d_samp.p, 0, (size_t)E * 8, S.st));
    uint32_t grid = (uint32_t)std::min<uint64_t>((total + TB - 1) / TB, 65536);
    k_bs_sample<<<grid, TB, 0, S.st>>>(total, E, d_cum.p, seed, b, d_samp.p);
    k_u64_to_f64<<<(E + TB - 1) / TB, TB, 0, S.st>>>(E, d_samp.p, S.d_cnt.p);
    alpha = init; sq_em_report rep;
    rc = S.run(alpha, 0, 0, /*minIter=*/50, &rep); if (rc) return rc;                                  // :412
    for (uint32_t i = 0; i < M; ++i) if (alpha[i] <= 1e-8) alpha[i] = 0.0;                             // truncateCountVector (:509-520)
    double asum = canonical_sum_host(alpha);
    if (asum < 2.2250738585072014e-308) {
      sq_set_error("Total alpha weight was too small! Make sure you ran salmon correctly.");
      return SQ_ERR_STATE;
    }
    if (cb(alpha.data(), M, user)) break;
  }
  return SQ_OK;
}

extern "C" int sq_gibbs_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_gibbs_opts* go, const double* alpha_init,
    uint32_t S_n, uint64_t seed,
    uint64_t num_mapped, sq_replicate_cb cb, void* user) {
  return sq_gibbs_range_dev(device, eq, txp, go, alpha_init, S_n, 0, S_n, seed, num_mapped, cb, user);
}
extern "C" uint32_t sq_gibbs_chain_step(uint32_t S_n) {   // samples per chain: 1 / 2 / 4 / 8 chains from 1 / 50 / 100 / 200 samples (CollapsedGibbsSampler.cpp:425-434)
  uint32_t nchains = 1; if (S_n >= 50) nchains = 2; if (S_n >= 100) nchains = 4; if (S_n >= 200) nchains = 8;
  return nchains > 1 ? S_n / nchains : (S_n ? S_n : 1);
}
// samples [first, first + count) of S_n; `first` must start a chain (a multiple of sq_gibbs_chain_step(S_n), below nchains * step): every
// chain restarts from alpha_init and round r of sample s draws from the counter RNG keyed (seed, s * thin + r), so whole chains can
// run on different GPUs and give the samples a single GPU would have produced
extern "C" int sq_gibbs_range_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_gibbs_opts* go, const double* alpha_init,
    uint32_t S_n, uint32_t first, uint32_t count, uint64_t seed, uint64_t num_mapped, sq_replicate_cb cb, void* user) {
  return sq_gibbs_range_report_dev(device, eq, txp, go, alpha_init, S_n, first, count, seed, num_mapped, cb, user, nullptr);
}
extern "C" int sq_gibbs_range_report_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_gibbs_opts* go, const double* alpha_init,
    uint32_t S_n, uint32_t first, uint32_t count, uint64_t seed, uint64_t num_mapped, sq_replicate_cb cb, void* user, sq_gibbs_report* report) {
  if (report) memset(report, 0, sizeof(*report));
  if (first > S_n || count > S_n - first) { sq_set_error("sq_gibbs_range_dev: range [%u, %u) outside %u samples", first, first + count, S_n); return SQ_ERR_ARG; }
  { const uint32_t stp = sq_gibbs_chain_step(S_n); uint32_t nch = 1; if (S_n >= 50) nch = 2; if (S_n >= 100) nch = 4; if (S_n >= 200) nch = 8;
    if (first && (nch == 1 || first % stp != 0 || first / stp >= nch)) { sq_set_error("sq_gibbs_range_dev: sample %u does not start a chain (step %u)", first, stp); return SQ_ERR_ARG; } }
  if (count == 0) return SQ_OK;
  if (!eq || !txp || !go || !alpha_init || !cb || !eq->off || !eq->tid || !eq->w || !eq->count || !txp->eff_len) {
    sq_set_error("sq_gibbs_dev: bad arguments");
    return SQ_ERR_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    sq_set_error("no HIP device %d (found %d): Gibbs has no CPU fallback", device, ndev);
    return SQ_ERR_DEVICE;
  }
  SQ_HIP_CHECK(hipSetDevice(device));
  const uint32_t M = txp->num_txp; const uint32_t E = (uint32_t)eq->num_classes; const uint64_t L = eq->num_labels;
  // prior (CollapsedGibbsSampler.cpp:357-371): 1e-3 per transcript after EM; under VB max(vbPrior,1) per transcript or max(vbPrior,1e-3) per nucleotide
  const bool perTxp = go->use_vbem ? go->per_transcript_prior != 0 : true;
  double pv = 1e-3; if (go->use_vbem) pv = perTxp ? (go->vb_prior < 1.0 ? 1.0 : go->vb_prior) : (go->vb_prior < 1e-3 ? 1e-3 : go->vb_prior);
  std::vector<double> prior(M, pv); if (!perTxp) for (uint32_t i = 0; i < M; ++i) prior[i] = pv * txp->eff_len[i];
  std::vector<uint8_t> active(M, 0); for (uint64_t i = 0; i < L; ++i) active[eq->tid[i]] = 1;
  std::vector<double> init(alpha_init, alpha_init + M); for (uint32_t i = 0; i < M; ++i) if (!active[i]) init[i] = 0.0;
  std::vector<uint64_t> off(eq->off, eq->off + E + 1), cnt(eq->count, eq->count + E), draw_off(E + 1, 0);
  std::vector<uint32_t> tid(eq->tid, eq->tid + L);
  std::vector<double> w(eq->w, eq->w + L), eff(txp->eff_len, txp->eff_len + M);
  // items = runs of up to 256 draws of one class, in three lists by class size (registers for <= 8 and <= 16 labels, binary search above),
  // each list ordered by the item's draw count so that the lanes of a wave run loops of similar length
  std::vector<uint32_t> item_cls, item_s0; uint32_t list_n[3] = {0, 0, 0}; uint64_t draws_per_round = 0;
  { std::vector<std::pair<uint32_t, uint32_t>> lists[3];
    for (uint32_t c = 0; c < E; ++c) { draw_off[c + 1] = draw_off[c] + cnt[c]; const uint64_t n = off[c + 1] - off[c]; if (n == 0 || cnt[c] == 0) continue;
      if (n > 1) draws_per_round += cnt[c];
      const int li = n <= 8 ? 0 : (n <= 16 ? 1 : 2);
      if (n == 1) lists[0].emplace_back(c, 0u); else for (uint64_t s0 = 0; s0 < cnt[c]; s0 += 256) lists[li].emplace_back(c, (uint32_t)s0); }
    for (int li = 0; li < 3; ++li) {
      auto draws = [&](const std::pair<uint32_t, uint32_t>& x) { const uint64_t n = off[x.first + 1] - off[x.first]; return n == 1 ? (uint64_t)0 : std::min<uint64_t>(256, cnt[x.first] - x.second); };
      std::stable_sort(lists[li].begin(), lists[li].end(), [&](const std::pair<uint32_t, uint32_t>& x, const std::pair<uint32_t, uint32_t>& y) { return draws(x) > draws(y); });
      list_n[li] = (uint32_t)lists[li].size();
      for (auto& x : lists[li]) { item_cls.push_back(x.first); item_s0.push_back(x.second); } } }
  DBuf<uint64_t> d_off, d_cnt, d_doff;
  DBuf<uint32_t> d_tid, d_ic, d_is;
  DBuf<double> d_w, d_eff, d_prior, d_mu, d_cf, d_out, d_me, d_cum, d_den;
  DBuf<uint8_t> d_act;
  DBuf<unsigned long long> d_ci;
  if (d_off.upload(off) || d_cnt.upload(cnt) || d_doff.upload(draw_off) || d_tid.upload(tid) || d_ic.upload(item_cls) ||
      d_is.upload(item_s0) || d_w.upload(w) ||
      d_eff.upload(eff) || d_prior.upload(prior) ||
      d_mu.alloc(M) || d_cf.upload(init) || d_out.alloc(M) || d_me.alloc(M) || d_act.upload(active) ||
          d_ci.alloc(M) || d_cum.alloc(L + 1) || d_den.alloc((size_t)E + 1)) { sq_set_error("device allocation failed (Gibbs)"); return SQ_ERR_NOMEM; }
  GibbsDev g;
  g.M = M;
  g.E = E;
  g.off = d_off.p;
  g.tid = d_tid.p;
  g.w = d_w.p;
  g.cnt = d_cnt.p;
  g.eff = d_eff.p;
  g.prior = d_prior.p;
  g.active = d_act.p;
  g.mu = d_mu.p;
  g.count_f = d_cf.p;
  g.count_i = d_ci.p;
  g.draw_off = d_doff.p;
  uint32_t nchains = 1; if (S_n >= 50) nchains = 2; if (S_n >= 100) nchains = 4; if (S_n >= 200) nchains = 8;    // :425-434
  const uint32_t step = nchains > 1 ? S_n / nchains : S_n + 1;
  const uint32_t thin = go->thinning_factor ? go->thinning_factor : 16;
  const int TB = 256;
  std::vector<double> me(M), alphas(M);
  hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
  SQ_HIP_CHECK(hipStreamCreate(&st)); SQ_HIP_CHECK(hipEventCreate(&e0)); SQ_HIP_CHECK(hipEventCreate(&e1));
  struct Cleanup { hipStream_t& s; hipEvent_t& a; hipEvent_t& b; ~Cleanup() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); if (s) (void)hipStreamDestroy(s); } } cleanup{st, e0, e1};
  double round_ms = 0.0; uint64_t rounds = 0;
  const bool fold16 = list_n[1] < 65536;   // few 9..16-label items: they ride with the large ones
  const uint32_t rest0 = list_n[0] + (fold16 ? 0u : list_n[1]), nrest = list_n[2] + (fold16 ? list_n[1] : 0u);
  const uint32_t nb0 = (list_n[0] + TB - 1) / TB, nb1 = fold16 ? 0u : (list_n[1] + TB - 1) / TB, nbr = (nrest + 3) / 4;   // a wave per item
  for (uint32_t sid = first; sid < first + count; ++sid) {
    // chain restart (:452-455); a range that starts at a later chain starts from the initial counts as well
    bool from_f = sid == first;
    if (sid > 0 && nchains > 1 && sid % step == 0 && sid / step < nchains) { SQ_HIP_CHECK(hipMemcpyAsync(d_cf.p, init.data(), (size_t)M * 8,
        hipMemcpyHostToDevice, st)); from_f = true; }
    SQ_HIP_CHECK(hipEventRecord(e0, st));
    for (uint32_t r = 0; r < thin; ++r) {
      const uint64_t key = (uint64_t)sid * thin + r;
      k_gibbs_mu<<<(M + TB - 1) / TB, TB, 0, st>>>(g, seed, key, go->no_gamma_draw, (from_f && r == 0) ? 1 : 0);
      if (E) k_gibbs_prep<<<(E + TB - 1) / TB, TB, 0, st>>>(g, d_cum.p, d_den.p);
      if (nb0 + nbr) k_gibbs_items<<<nb0 + nbr, TB, 0, st>>>(g, list_n[0], nb0, rest0, nrest, d_ic.p, d_is.p, seed, key, d_cum.p, d_den.p);
      if (nb1) k_gibbs_items16<<<nb1, TB, 0, st>>>(g, list_n[1], d_ic.p + list_n[0], d_is.p + list_n[0], seed, key, d_cum.p, d_den.p);
    }
    SQ_HIP_CHECK(hipEventRecord(e1, st));
    k_mul<<<(M + TB - 1) / TB, TB, 0, st>>>(M, d_mu.p, d_eff.p, d_me.p);
    SQ_HIP_CHECK(hipMemcpyAsync(me.data(), d_me.p, (size_t)M * 8, hipMemcpyDeviceToHost, st)); SQ_HIP_CHECK(sq_em_wait(st));
    { float ms = 0; if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) { round_ms += ms; rounds += thin; } }
    double denom = canonical_sum_host(me);                                                                // :489-492 (order-defined sum)
    double scale = (double)num_mapped / denom;
    k_gibbs_alpha<<<(M + TB - 1) / TB, TB, 0, st>>>(g, scale, d_out.p);
    SQ_HIP_CHECK(hipMemcpyAsync(alphas.data(), d_out.p, (size_t)M * 8, hipMemcpyDeviceToHost, st)); SQ_HIP_CHECK(sq_em_wait(st));
    if (cb(alphas.data(), M, user)) break;
  }
  if (report) { report->rounds = rounds; report->device_ms = round_ms; report->ms_per_round = rounds ? round_ms / (double)rounds : 0.0; report->draws_per_round = draws_per_round;
    report->items[0] = list_n[0]; report->items[1] = list_n[1]; report->items[2] = list_n[2]; report->_pad = 0; }
  return SQ_OK;
}
