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
#include <vector>
#include <algorithm>
#include <cmath>
#include "device_index.h"

namespace {

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
};

__device__ inline void em_close(EmDev& d, uint32_t it_index, unsigned long long* maxrel_log) {
  uint32_t it = it_index + 1;
  d.flags[2] = it;
  bool conv = (d.flags[1] == 0);
  maxrel_log[0] = *d.maxrel;
  if (conv && it >= d.min_iter) d.flags[0] = it;
  d.flags[1] = 0;
  *d.maxrel = 0ULL;
}

// First kernel of an iteration (ONE block): closes the previous iteration's convergence bookkeeping,
// then finishes the canonical sum of (alpha + prior) from the level-1 partials (SPEC §D2: 64-leaf
// strided-halving trees, level by level) and publishes logNorm = digamma(sum).  Being a single
// block, the `done` flag it may set is visible to every later kernel of the iteration.
__global__ void __launch_bounds__(1024) k_top(EmDev d, const double* __restrict__ partials, uint32_t n1, int close_prev, uint32_t prev_it, unsigned long long* maxrel_log, double* __restrict__ log_norm) {
  __shared__ double buf[2][4096];
  if (threadIdx.x == 0 && close_prev && !d.flags[0]) em_close(d, prev_it, maxrel_log);
  __syncthreads();
  if (d.flags[0]) return;
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

__global__ void k_class(EmDev d, const double* __restrict__ theta) {
  if (d.flags[0]) return;
  uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d.E) return;
  uint64_t a = d.off[c], b = d.off[c + 1];
  if (b - a <= 1) { d.inv[c] = (b - a == 1) ? -d.cnt[c] : 0.0; return; }  // single-transcript class gets the full count (:316-318)
  double denom = 0.0;
  uint64_t i = a;
  for (; i + 4 <= b; i += 4) {   // gathers issued together, sums kept in label order
    const uint32_t t0 = d.tid[i], t1 = d.tid[i + 1], t2 = d.tid[i + 2], t3 = d.tid[i + 3];
    const double w0 = d.cw[i], w1 = d.cw[i + 1], w2 = d.cw[i + 2], w3 = d.cw[i + 3];
    const double h0 = theta[t0], h1 = theta[t1], h2 = theta[t2], h3 = theta[t3];
    if (!d.use_vbem || h0 > 0.0) denom += h0 * w0;
    if (!d.use_vbem || h1 > 0.0) denom += h1 * w1;
    if (!d.use_vbem || h2 > 0.0) denom += h2 * w2;
    if (!d.use_vbem || h3 > 0.0) denom += h3 * w3;
  }
  for (; i < b; ++i) { double th = theta[d.tid[i]]; if (!d.use_vbem || th > 0.0) denom += th * d.cw[i]; }
  d.inv[c] = (denom <= 2.2250738585072014e-308) ? 0.0 : d.cnt[c] / denom;  // minEQClassWeight (:40)
}

#define SEG_TOP 0x80000000u
// level 1: one thread per run of <= 64 consecutive CSC entries of one transcript
__global__ void k_l1(EmDev d, const double* __restrict__ theta, double* __restrict__ alpha_out) {
  if (d.flags[0]) return;
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= d.nseg[0]) return;
  const uint32_t tt = d.seg_txp[0][g]; const uint32_t t = tt & ~SEG_TOP;
  const double th = theta[t]; const bool live = !d.use_vbem || th > 0.0;
  double acc = 0.0;
  uint64_t p = d.seg_lo[0][g]; const uint64_t e = p + d.seg_cnt[0][g];
  auto term = [&](double iv, double cw) { if (iv < 0.0) acc += -iv; else if (iv != 0.0 && live) { double v = th * cw; acc += v * iv; } };
  for (; p + 8 <= e; p += 8) {   // 8 gathers in flight, terms added in CSC order
    uint32_t c[8]; double w[8], iv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { c[j] = d.t_cls[p + j]; w[j] = d.t_cw[p + j]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) iv[j] = d.inv[c[j]];
#pragma unroll
    for (int j = 0; j < 8; ++j) term(iv[j], w[j]);
  }
  for (; p < e; ++p) term(d.inv[d.t_cls[p]], d.t_cw[p]);
  if (tt & SEG_TOP) alpha_out[t] = acc; else d.part[0][g] = acc;
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
// convergence scan (CollapsedEMOptimizer.cpp:945-957)
__global__ void k_fin(EmDev d, const double* __restrict__ alpha, double* __restrict__ alpha_out, double* __restrict__ partials) {
  if (d.flags[0]) return;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  double rel = -1.0; int bad = 0; double leaf = 0.0;
  if (t < d.M) {
    double acc = (d.t_off[t + 1] == d.t_off[t]) ? 0.0 : alpha_out[t];
    if (d.t_off[t + 1] == d.t_off[t]) alpha_out[t] = 0.0;
    leaf = acc + d.prior[t];
    if (acc > 1e-2) {  // alphaCheckCutoff (:884)
      rel = fabs(alpha[t] - acc) / acc;
      if (rel > d.tol) bad = 1;
    }
  }
  if (partials) { double ls = wave_halving_sum(leaf); if ((threadIdx.x & 63) == 0 && (t >> 6) < ((d.M + 63) >> 6)) partials[t >> 6] = ls; }  // level 1 of next iteration's canonical sum
  for (int s = 32; s >= 1; s >>= 1) { double o = __shfl_down(rel, s, 64); int ob = __shfl_down(bad, s, 64); rel = o > rel ? o : rel; bad |= ob; }
  // block-level combine, then one atomic per block only when it can raise the running maximum
  __shared__ double srel[16]; __shared__ int sbad[16];
  if ((threadIdx.x & 63) == 0) { srel[threadIdx.x >> 6] = rel; sbad[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) { for (uint32_t w = 1; w < (blockDim.x >> 6); ++w) { if (srel[w] > rel) rel = srel[w]; bad |= sbad[w]; } }
  if (threadIdx.x == 0) {
    if (rel >= 0.0) { unsigned long long b = (unsigned long long)__double_as_longlong(rel); if (b > __hip_atomic_load(d.maxrel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(d.maxrel, b); }
    if (bad && __hip_atomic_load(&d.flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) d.flags[1] = 1;
  }
}

// one thread: close the iteration (convergence bookkeeping lives on the device so the host never
// has to synchronise inside the loop)
__global__ void k_close(EmDev d, uint32_t it_index /* 0-based index of the iteration just run */, unsigned long long* maxrel_log) {
  if (d.flags[0]) return;
  em_close(d, it_index, maxrel_log);
}

template <class T>
struct DBuf {
  T* p = nullptr;
  ~DBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)) == hipSuccess ? 0 : -1; }
  int upload(const std::vector<T>& v) { if (alloc(v.size())) return -1; return v.empty() || hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; }
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

struct EmHost {  // host-side preparation (CollapsedEMOptimizer.cpp:760-873)
  std::vector<double> cw, cnt, prior, t_cw; std::vector<uint64_t> t_off; std::vector<uint32_t> t_cls;
  std::vector<uint32_t> seg_lo[4], seg_txp[4]; std::vector<uint8_t> seg_cnt[4]; int nlevels = 0;
};

int prepare(const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, EmHost& H) {
  const uint64_t E = eq->num_classes, L = eq->num_labels; const uint32_t M = txp->num_txp;
  if (E >= 0xFFFFFFFFull) { sq_set_error("too many equivalence classes"); return SQ_ERR_OVERFLOW; }
  for (uint64_t i = 0; i < L; ++i) if (eq->tid[i] >= M) { sq_set_error("eq-class label references transcript %u >= %u", eq->tid[i], M); return SQ_ERR_ARG; }
  H.cw.resize(L); H.cnt.resize(E);
  for (uint64_t c = 0; c < E; ++c) {
    H.cnt[c] = (double)eq->count[c];
    double wsum = 0.0;
    for (uint64_t i = eq->off[c]; i < eq->off[c + 1]; ++i) {
      double el = txp->eff_len[eq->tid[i]]; if (el <= 1.0) el = 1.0;           // :841-844
      double w = o->no_rich_eq_classes ? 1.0 : eq->w[i];                        // :845-848
      double wt = o->eq_class_mode ? w : (double)eq->count[c] * w * (1.0 / el); // :850-853
      H.cw[i] = wt; wsum += wt;
    }
    double wn = 1.0 / wsum;
    for (uint64_t i = eq->off[c]; i < eq->off[c + 1]; ++i) H.cw[i] = H.cw[i] * wn;
  }
  H.prior.assign(M, o->vb_prior);
  if (!o->per_transcript_prior) for (uint32_t i = 0; i < M; ++i) H.prior[i] = o->vb_prior * txp->eff_len[i];  // populatePriorAlphas_ :82-99
  H.t_off.assign((size_t)M + 1, 0);
  for (uint64_t i = 0; i < L; ++i) H.t_off[eq->tid[i] + 1]++;
  for (uint32_t t = 0; t < M; ++t) H.t_off[t + 1] += H.t_off[t];
  H.t_cls.resize(L); H.t_cw.resize(L);
  std::vector<uint64_t> cur(H.t_off.begin(), H.t_off.end() - 1);
  for (uint64_t c = 0; c < E; ++c) for (uint64_t i = eq->off[c]; i < eq->off[c + 1]; ++i) { uint64_t d = cur[eq->tid[i]]++; H.t_cls[d] = (uint32_t)c; H.t_cw[d] = H.cw[i]; }
  // blocked-64 reduction plan (SPEC §D4)
  if (L >= 0x7FFFFFFFull) { sq_set_error("too many label entries for the EM reduction plan"); return SQ_ERR_OVERFLOW; }
  std::vector<uint32_t> cnt_prev(M), lo_prev(M);   // per transcript: number of items and first item index at the previous level
  for (uint32_t t = 0; t < M; ++t) { cnt_prev[t] = (uint32_t)(H.t_off[t + 1] - H.t_off[t]); lo_prev[t] = (uint32_t)H.t_off[t]; }
  H.nlevels = 0;
  for (int lvl = 0; lvl < 4; ++lvl) {
    bool any = false;
    for (uint32_t t = 0; t < M; ++t) {
      uint32_t n = cnt_prev[t]; if (n == 0 || (lvl > 0 && n == 1)) { if (lvl > 0) cnt_prev[t] = 0; continue; }
      any = true;
      uint32_t ns = (n + 63) / 64; uint32_t first = (uint32_t)H.seg_lo[lvl].size();
      for (uint32_t j = 0; j < ns; ++j) { H.seg_lo[lvl].push_back(lo_prev[t] + 64 * j); H.seg_cnt[lvl].push_back((uint8_t)std::min<uint32_t>(64, n - 64 * j)); H.seg_txp[lvl].push_back(t | (ns == 1 ? SEG_TOP : 0u)); }
      cnt_prev[t] = ns; lo_prev[t] = first;
    }
    if (!any) break;
    H.nlevels = lvl + 1;
  }
  for (uint32_t t = 0; t < M; ++t) if (cnt_prev[t] > 1) { sq_set_error("EM reduction plan deeper than 4 levels"); return SQ_ERR_OVERFLOW; }
  return SQ_OK;
}

// Runs the iteration loop. mode 0: optimise to convergence; mode 1: exactly `fixed_iters` steps.
int run_em(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, std::vector<double>& alpha,
           int mode, uint32_t fixed_iters, sq_em_report* rep) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { sq_set_error("no HIP device %d (found %d): EM has no CPU fallback", device, ndev); return SQ_ERR_DEVICE; }
  SQ_HIP_CHECK(hipSetDevice(device));
  EmHost H; int rc = prepare(eq, txp, o, H); if (rc) return rc;
  const uint32_t M = txp->num_txp; const uint32_t E = (uint32_t)eq->num_classes; const uint64_t L = eq->num_labels;
  DBuf<uint64_t> d_off, d_toff; DBuf<uint32_t> d_tid, d_tcls, d_flags; DBuf<double> d_cw, d_cnt, d_tcw, d_prior, d_theta, d_inv, d_a0, d_a1, d_part; DBuf<unsigned long long> d_maxrel, d_log; DBuf<double> d_lognorm;
  std::vector<uint64_t> off(eq->off, eq->off + E + 1); std::vector<uint32_t> tid(eq->tid, eq->tid + L);
  uint32_t g1 = (M + 63) / 64;
  DBuf<uint32_t> d_slo[4], d_stx[4]; DBuf<uint8_t> d_scn[4]; DBuf<double> d_lpart[4];
  for (int l = 0; l < H.nlevels; ++l) if (d_slo[l].upload(H.seg_lo[l]) || d_stx[l].upload(H.seg_txp[l]) || d_scn[l].upload(H.seg_cnt[l]) || d_lpart[l].alloc(H.seg_lo[l].size() + 1)) { sq_set_error("device allocation failed in EM plan"); return SQ_ERR_NOMEM; }
  bool ok = !d_off.upload(off) && !d_tid.upload(tid) && !d_cw.upload(H.cw) && !d_cnt.upload(H.cnt) && !d_toff.upload(H.t_off) && !d_tcls.upload(H.t_cls) &&
            !d_tcw.upload(H.t_cw) && !d_prior.upload(H.prior) && !d_theta.alloc(M) && !d_inv.alloc(E) && !d_a0.upload(alpha) && !d_a1.alloc(M) &&
            !d_part.alloc((size_t)g1 * 3 + 512) && !d_flags.alloc(4) && !d_maxrel.alloc(1) && !d_log.alloc(1) && !d_lognorm.alloc(1);
  if (!ok) { sq_set_error("device allocation failed in EM (%s)", hipGetErrorString(hipGetLastError())); return SQ_ERR_NOMEM; }
  SQ_HIP_CHECK(hipMemset(d_flags.p, 0, 4 * sizeof(uint32_t))); SQ_HIP_CHECK(hipMemset(d_maxrel.p, 0, 8)); SQ_HIP_CHECK(hipMemset(d_log.p, 0, 8));
  EmDev d; d.M = M; d.E = E; d.L = L; d.off = d_off.p; d.tid = d_tid.p; d.cw = d_cw.p; d.cnt = d_cnt.p; d.t_off = d_toff.p; d.t_cls = d_tcls.p; d.t_cw = d_tcw.p;
  d.prior = d_prior.p; d.theta = d_theta.p; d.inv = d_inv.p; d.partial = d_part.p; d.flags = d_flags.p; d.maxrel = d_maxrel.p; d.tol = o->rel_diff_tolerance; d.use_vbem = o->use_vbem;
  d.min_iter = (mode == 0) ? o->min_iter : 0xFFFFFFFFu;
  d.nlevels = H.nlevels; for (int l = 0; l < 4; ++l) { d.seg_lo[l] = d_slo[l].p; d.seg_cnt[l] = d_scn[l].p; d.seg_txp[l] = d_stx[l].p; d.nseg[l] = l < H.nlevels ? (uint32_t)H.seg_lo[l].size() : 0; d.part[l] = d_lpart[l].p; }
  hipStream_t st; SQ_HIP_CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; SQ_HIP_CHECK(hipEventCreate(&e0)); SQ_HIP_CHECK(hipEventCreate(&e1));
  const int TB = 256;
  double* cur = d_a0.p; double* nxt = d_a1.p;
  // level-1 partials of (alpha + prior) live in d.partial; levels above are finished inside k_theta.
  // If there are more than 4096 of them (M > 262144) extra level kernels shrink the list first.
  double* part_lvl1 = d.partial; double* part_tmp = d.partial + g1 + 64;
  bool pending_close = false; uint32_t pending_it = 0;
  auto launch_iter = [&](uint32_t it) {
    const double* theta_src = cur;
    if (o->use_vbem) {
      const double* pin = part_lvl1; uint32_t n1 = g1;
      double* a = part_tmp; double* b = part_tmp + g1 / 64 + 64;
      while (n1 > 4096) { k_sum_level<<<(n1 + TB - 1) / TB, TB, 0, st>>>(pin, nullptr, n1, a); pin = a; n1 = (n1 + 63) / 64; std::swap(a, b); }
      k_top<<<1, 1024, 0, st>>>(d, pin, n1, pending_close ? 1 : 0, pending_it, d_log.p, d_lognorm.p);
      k_theta<<<(M + TB - 1) / TB, TB, 0, st>>>(d, cur, d_lognorm.p);
      pending_close = false;
      theta_src = d.theta;
    } else if (pending_close) { k_close<<<1, 1, 0, st>>>(d, pending_it, d_log.p); pending_close = false; }
    k_class<<<(E + TB - 1) / TB, TB, 0, st>>>(d, theta_src);
    if (d.nseg[0]) k_l1<<<(d.nseg[0] + TB - 1) / TB, TB, 0, st>>>(d, theta_src, nxt);
    { int l = 1; for (; l < d.nlevels && d.nseg[l] > 1024; ++l) k_level<<<(d.nseg[l] + TB - 1) / TB, TB, 0, st>>>(d, l, nxt);
      if (l < d.nlevels) k_upper<<<1, 1024, 0, st>>>(d, l, nxt); }
    k_fin<<<(M + TB - 1) / TB, TB, 0, st>>>(d, cur, nxt, o->use_vbem ? part_lvl1 : nullptr);
    pending_close = true; pending_it = it;
    std::swap(cur, nxt);
  };
  auto flush_close = [&]() { if (pending_close) { k_close<<<1, 1, 0, st>>>(d, pending_it, d_log.p); pending_close = false; } };
  if (o->use_vbem) k_sum_level<<<(M + TB - 1) / TB, TB, 0, st>>>(cur, d.prior, M, part_lvl1);
  uint32_t it = 0, executed = 0; uint32_t done = 0; uint32_t hflags[4] = {0, 0, 0, 0};
  SQ_HIP_CHECK(hipEventRecord(e0, st));
  if (mode == 1) {
    for (; it < fixed_iters; ++it) launch_iter(it);
    flush_close();
    executed = fixed_iters;
  } else {
    const uint32_t maxIter = o->max_iter, minIter = o->min_iter;
    // run to min_iter without looking, then in chunks; kernels of iterations after convergence are no-ops
    while (it < maxIter || it < minIter) {
      uint32_t chunk = (it < minIter) ? (minIter - it) : 16;
      uint32_t lim = std::max(maxIter, minIter);
      if (it + chunk > lim) chunk = lim - it;
      for (uint32_t j = 0; j < chunk; ++j, ++it) launch_iter(it);
      flush_close();
      SQ_HIP_CHECK(hipMemcpyAsync(hflags, d_flags.p, sizeof(hflags), hipMemcpyDeviceToHost, st));
      SQ_HIP_CHECK(hipStreamSynchronize(st));
      if (hflags[0]) { done = hflags[0]; break; }
    }
    executed = done ? done : it;
  }
  SQ_HIP_CHECK(hipEventRecord(e1, st)); SQ_HIP_CHECK(hipStreamSynchronize(st));
  float ms = 0; SQ_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  // result buffer: after `executed` swaps starting from d_a0
  double* res = (executed % 2 == 0) ? d_a0.p : d_a1.p;
  SQ_HIP_CHECK(hipMemcpy(alpha.data(), res, (size_t)M * 8, hipMemcpyDeviceToHost));
  unsigned long long mr = 0; SQ_HIP_CHECK(hipMemcpy(&mr, d_log.p, 8, hipMemcpyDeviceToHost));
  if (rep) {
    rep->iters = executed; rep->converged = (mode == 0) ? (done != 0) : 0; double mrd; memcpy(&mrd, &mr, 8); rep->max_rel_diff = mrd;
    rep->device_ms = ms; rep->ms_per_iter = (mode == 1 ? fixed_iters : it) ? ms / (double)(mode == 1 ? fixed_iters : it) : 0.0; rep->alpha_sum = 0;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
  return SQ_OK;
}

}  // namespace

extern "C" int sq_em_optimize_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, double* alpha_out, sq_em_report* rep) {
  if (!eq || !txp || !o || !alpha_out || !eq->off || !eq->tid || !eq->w || !eq->count || !txp->eff_len) { sq_set_error("sq_em_optimize_dev: bad arguments"); return SQ_ERR_ARG; }
  const uint32_t M = txp->num_txp;
  // initial alphas (CollapsedEMOptimizer.cpp:778-823)
  std::vector<double> pc(M, 0.0); if (txp->projected_counts) pc.assign(txp->projected_counts, txp->projected_counts + M);
  double totalWeight = canonical_sum_host(pc);
  double uniformPrior = totalWeight / (double)M;
  double fracObserved = std::min(0.999, totalWeight / o->num_required_fragments);
  std::vector<double> alpha(M);
  for (uint32_t i = 0; i < M; ++i) alpha[i] = o->init_uniform ? 100.0 : (pc[i] * fracObserved + uniformPrior * (1.0 - fracObserved));
  int rc = run_em(device, eq, txp, o, alpha, 0, 0, rep);
  if (rc) return rc;
  for (uint32_t i = 0; i < M; ++i) if (alpha[i] <= 1e-8) alpha[i] = 0.0;  // truncateCountVector (:64-76), minAlpha 1e-8
  double asum = canonical_sum_host(alpha);
  for (uint32_t i = 0; i < M; ++i) alpha_out[i] = alpha[i];
  if (rep) rep->alpha_sum = asum;
  if (asum < 2.2250738585072014e-308) { sq_set_error("Total alpha weight was too small! Make sure you ran salmon correctly."); return SQ_ERR_STATE; }  // :1016-1020
  return SQ_OK;
}

extern "C" int sq_em_steps_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* o, const double* alpha_in, uint32_t iters, double* alpha_out, sq_em_report* rep) {
  if (!eq || !txp || !o || !alpha_in || !alpha_out) { sq_set_error("sq_em_steps_dev: bad arguments"); return SQ_ERR_ARG; }
  std::vector<double> alpha(alpha_in, alpha_in + txp->num_txp);
  int rc = run_em(device, eq, txp, o, alpha, 1, iters, rep);
  if (rc) return rc;
  memcpy(alpha_out, alpha.data(), (size_t)txp->num_txp * 8);
  return SQ_OK;
}

