// hip/bias.hip — fragment-GC bias (row f-3, `--gcBias`): the expected GC model and the bias-corrected effective lengths of
// salmon::utils::updateEffectiveLengths (reference src/util/SalmonUtils.cpp:1208-1985, the gcBiasCorrect branches; GC model:
// include/salmon/internal/model/GCFragModel.hpp), called from inside the EM at iteration ~10 (CollapsedEMOptimizer.cpp:901-928).
//
// The reference sweeps, for every expressed transcript, every fragment start x every sampled fragment length (O(sum_t len_t * 25))
// twice, adding doubles into 25-bin models.  Here the GPU does the sweep ONCE as integers — a histogram of fragment-GC bins per
// (transcript, sampled length), read off a sampled G/C prefix of the 2-bit reference pool — and the floating-point part (25-term dot
// products in a fixed order) is a few hundred flops per transcript on the host: exact, order-defined, the same on every run (SPEC §B).
#include "ctx.h"
#include "../host/posbias.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace {
#define GC_MAX_SLOTS 256
struct GcHistArgs { uint32_t n_grid; int32_t fld_low, fld_high, samp; };

// block per processed transcript; slot j < n_grid: fl = fld_low + samp * j; slot n_grid: fl = min(refLen, fld_high + 1) - 1 (the last
// length of the effective-length loop).  hist[p][slot][bin] = starts s in [0, refLen - fl) whose fragment [s, s + fl - 1] falls in bin;
// last[p][slot] = bin of the window that ends on the last base (s = refLen - fl), 255 if there is none.
__global__ void __launch_bounds__(256) k_gc_hist(const uint64_t* __restrict__ refseq, const uint32_t* __restrict__ gcpre, const uint64_t* __restrict__ ref_accum,
                                                 const uint32_t* __restrict__ ref_len, const uint32_t* __restrict__ list, GcHistArgs A,
                                                 uint32_t* __restrict__ hist, uint8_t* __restrict__ last) {
  __shared__ uint32_t h[SQ_GC_FRAG_BINS];
  const uint32_t p = blockIdx.x, t = list[p];
  const int32_t refLen = (int32_t)ref_len[t]; const uint64_t g = ref_accum[t];
  const uint32_t nslots = A.n_grid + 1;
  for (uint32_t slot = 0; slot < nslots; ++slot) {
    const int32_t fl = slot < A.n_grid ? A.fld_low + A.samp * (int32_t)slot : std::min(refLen, A.fld_high + 1) - 1;
    if (threadIdx.x < SQ_GC_FRAG_BINS) h[threadIdx.x] = 0;
    __syncthreads();
    uint8_t lastbin = 255;
    if (fl >= 1 && fl <= refLen) {
      const int32_t nfull = refLen - fl;           // starts [0, nfull) are counted, start nfull is the last window
      for (int32_t s = (int32_t)threadIdx.x; s <= nfull; s += 256) {
        const uint64_t c = sq_gc_before(refseq, gcpre, g + (uint64_t)s + (uint64_t)fl) - sq_gc_before(refseq, gcpre, g + (uint64_t)s);
        const int32_t frac = (int32_t)rint((100.0 * (double)c) / (double)fl);     // Transcript::gcFrac (Transcript.hpp:423-428)
        const int32_t bin = sq_gc_frag_bin(frac);
        if (s < nfull) atomicAdd(&h[bin], 1u); else lastbin = (uint8_t)bin;
      }
    }
    __syncthreads();
    if (threadIdx.x < SQ_GC_FRAG_BINS) hist[((size_t)p * nslots + slot) * SQ_GC_FRAG_BINS + threadIdx.x] = h[threadIdx.x];
    if (lastbin != 255) last[(size_t)p * nslots + slot] = lastbin;
    __syncthreads();
  }
}

double canonical_sum_vec(std::vector<double> x) {   // SPEC §D2: 64-leaf strided-halving trees, level by level
  for (;;) {
    const size_t n = x.size(); if (n == 0) return 0.0;
    const size_t gcount = (n + 63) / 64; std::vector<double> p(gcount);
    for (size_t b = 0; b < gcount; ++b) {
      double v[64]; for (int i = 0; i < 64; ++i) v[i] = (b * 64 + i < n) ? x[b * 64 + i] : 0.0;
      for (int st = 32; st >= 1; st >>= 1) for (int i = 0; i < st; ++i) v[i] = v[i] + v[i + st];
      p[b] = v[0];
    }
    if (gcount == 1) return p[0];
    x.swap(p);
  }
}
}  // namespace

// the expected fragment-GC masses of the calling thread's last sweep (aux_info/exp_gc.gz, GZipWriter.cpp:405-413): [context class][GC bin], linear
static thread_local double tl_gc_expected[SQ_GC_COND_BINS * SQ_GC_FRAG_BINS]; static thread_local int tl_gc_expected_rows = 0;
extern "C" int sq_bias_last_gc_expected(double* out75) {
  if (!out75) return SQ_ERR_ARG;
  if (!tl_gc_expected_rows) { sq_set_error("sq_bias_last_gc_expected: no --gcBias sweep has run on this thread"); return SQ_ERR_STATE; }
  memcpy(out75, tl_gc_expected, sizeof(tl_gc_expected)); return SQ_OK;
}
extern "C" int sq_bias_gc_eff_lengths(sq_index* idx, const double* gc_obs, const double* log_pmf, uint32_t M, const double* alphas,
                                      const double* eff_in, double* eff_out, sq_bias_report* rep) {
  if (!idx || !gc_obs || !log_pmf || !alphas || !eff_in || !eff_out) { sq_set_error("sq_bias_gc_eff_lengths: bad arguments"); return SQ_ERR_ARG; }
  if (!idx->dev) { sq_set_error("sq_bias_gc_eff_lengths: the index is not on a device (there is no CPU path)"); return SQ_ERR_DEVICE; }
  if (M > idx->names.size()) { sq_set_error("sq_bias_gc_eff_lengths: %u transcripts but the index has %zu", M, idx->names.size()); return SQ_ERR_ARG; }
  const sq_device_index* di = idx->dev;
  SQ_HIP_CHECK(hipSetDevice(di->device));
  // ---- fragment-length distribution: pdf, cdf, the 0.5 % / 99.5 % quantiles (SalmonUtils.cpp:1265-1290) ----
  const int MAXV = 1000; const int32_t samp = 5;   // biasSpeedSamp (SalmonDefaults.hpp:57)
  std::vector<double> pdf(MAXV + 1), cdf(MAXV + 1);
  int32_t fldLow = 0, fldHigh = 1; bool lb = false, ub = false;
  for (int i = 0; i <= MAXV; ++i) {
    pdf[i] = sq_exp(log_pmf[i]); cdf[i] = i > 0 ? cdf[i - 1] + pdf[i] : pdf[i];
    if (!lb && cdf[i] >= 0.005) { lb = true; fldLow = i; }
    if (!ub && cdf[i] >= 1.0 - 0.005) { ub = true; fldHigh = i; }
  }
  const uint32_t n_grid = fldHigh >= fldLow ? (uint32_t)((fldHigh - fldLow) / samp + 1) : 0u;
  if (n_grid + 1 > GC_MAX_SLOTS) { sq_set_error("sq_bias_gc_eff_lengths: fragment-length range %d..%d needs %u sampled lengths (limit %d)", fldLow, fldHigh, n_grid + 1, GC_MAX_SLOTS); return SQ_ERR_ARG; }
  // ---- transcripts that take part (alpha >= 1e-8, effective length shorter than the transcript, some CDF mass) ----
  std::vector<uint32_t> list; std::vector<int32_t> elen(M), unproc(M);
  for (uint32_t t = 0; t < M; ++t) {
    const int32_t refLen = (int32_t)idx->ref_len[t]; elen[t] = (int32_t)eff_in[t]; unproc[t] = std::max(0, refLen - elen[t]);
    const int32_t cdfMaxArg = std::min(MAXV, refLen);
    if (cdf[cdfMaxArg] < 1e-10 || alphas[t] < 1e-8 || unproc[t] <= 0) continue;
    list.push_back(t);
  }
  const size_t P = list.size(); const uint32_t nslots = n_grid + 1;
  std::vector<uint32_t> hist(P * nslots * SQ_GC_FRAG_BINS, 0); std::vector<uint8_t> last(P * nslots, 255);
  if (P) {
    sq_dbuf<uint32_t> d_list, d_hist; sq_dbuf<uint8_t> d_last;
    if (d_list.ensure(P) || d_hist.ensure(hist.size()) || d_last.ensure(last.size())) { sq_set_error("device allocation failed (GC histograms for %zu transcripts)", P); return SQ_ERR_NOMEM; }
    SQ_HIP_CHECK(hipMemcpy(d_list.p, list.data(), P * 4, hipMemcpyHostToDevice));
    SQ_HIP_CHECK(hipMemset(d_last.p, 0xFF, last.size()));
    GcHistArgs A{n_grid, fldLow, fldHigh, samp};
    k_gc_hist<<<(uint32_t)P, 256>>>(di->refseq, di->gcpre, di->ref_accum, di->ref_len, d_list.p, A, d_hist.p, d_last.p);
    SQ_HIP_CHECK(hipMemcpy(hist.data(), d_hist.p, hist.size() * 4, hipMemcpyDeviceToHost));
    SQ_HIP_CHECK(hipMemcpy(last.data(), d_last.p, last.size(), hipMemcpyDeviceToHost));
    d_list.free_(); d_hist.free_(); d_last.free_();
  }
  // ---- expected model (SalmonUtils.cpp:1574-1607, gc-only: the context bin is always 0) ----
  auto cond_cdf = [&](int32_t refLen, int32_t x) { const int32_t a = std::min(MAXV, refLen); return x > a ? 1.0 : cdf[x] / cdf[a]; };
  std::vector<std::vector<double>> contrib(SQ_GC_FRAG_BINS, std::vector<double>(P, 0.0));
  for (size_t p = 0; p < P; ++p) {
    const uint32_t t = list[p]; const int32_t refLen = (int32_t)idx->ref_len[t];
    const double weight = alphas[t] / eff_in[t];
    double E[SQ_GC_FRAG_BINS] = {0};
    double prev = cond_cdf(refLen, fldLow > 0 ? fldLow - 1 : 0);
    for (uint32_t j = 0; j < n_grid; ++j) {
      const int32_t fl = fldLow + samp * (int32_t)j;
      if (fl > refLen || fl < 1) break;                                    // fragEnd = start + fl - 1 < refLen has no solution any more
      const double d = cond_cdf(refLen, fl) - prev; prev = cond_cdf(refLen, fl);
      const uint32_t* h = &hist[(p * nslots + j) * SQ_GC_FRAG_BINS]; const uint8_t lb2 = last[p * nslots + j];
      const bool last_counts = refLen - fl < refLen - 1;                    // fragStartPos < refLen - K with K = 1
      for (int b = 0; b < SQ_GC_FRAG_BINS; ++b) {
        const double nwin = (double)h[b] + ((last_counts && lb2 == b) ? 1.0 : 0.0);
        E[b] += d * nwin;
      }
    }
    for (int b = 0; b < SQ_GC_FRAG_BINS; ++b) contrib[b][p] = weight * E[b];
  }
  double expect[SQ_GC_COND_BINS][SQ_GC_FRAG_BINS] = {{0}};
  for (int b = 0; b < SQ_GC_FRAG_BINS; ++b) expect[0][b] = canonical_sum_vec(contrib[b]);
  memcpy(tl_gc_expected, expect, sizeof(tl_gc_expected)); tl_gc_expected_rows = SQ_GC_COND_BINS;
  // ---- GCFragModel::normalize (prior 0.1) and ratio (maxRatio 1000), linear space (GCFragModel.hpp:118-140, 191-231) ----
  double obsN[SQ_GC_COND_BINS][SQ_GC_FRAG_BINS], expN[SQ_GC_COND_BINS][SQ_GC_FRAG_BINS], bias[SQ_GC_COND_BINS][SQ_GC_FRAG_BINS];
  auto normalize = [](const double* in, double* out) {
    double row = 0.0; for (int b = 0; b < SQ_GC_FRAG_BINS; ++b) row += (0.1 + in[b]);
    if (row > 0.0) { const double nrm = 1.0 / row; for (int b = 0; b < SQ_GC_FRAG_BINS; ++b) out[b] = (0.1 + in[b]) * nrm; }
    else for (int b = 0; b < SQ_GC_FRAG_BINS; ++b) out[b] = in[b];
  };
  for (int r = 0; r < SQ_GC_COND_BINS; ++r) {
    normalize(gc_obs + r * SQ_GC_FRAG_BINS, obsN[r]); normalize(expect[r], expN[r]);
    for (int b = 0; b < SQ_GC_FRAG_BINS; ++b) { double rat = obsN[r][b] / expN[r][b]; if (rat > 1000.0) rat = 1000.0; if (rat < 1.0 / 1000.0) rat = 1.0 / 1000.0; bias[r][b] = rat; }
  }
  // ---- effective lengths (SalmonUtils.cpp:1737-1975) ----
  for (uint32_t t = 0; t < M; ++t) eff_out[t] = (double)elen[t];            // not processed: the truncated input length (:1968-1970)
  for (size_t p = 0; p < P; ++p) {
    const uint32_t t = list[p]; const int32_t refLen = (int32_t)idx->ref_len[t];
    if (!(cdf[std::min(MAXV, refLen)] > 1e-10)) continue;
    int32_t fl = fldLow; const int32_t maxLen = std::min(refLen, fldHigh + 1);
    bool done = fl >= maxLen;
    double prev = cond_cdf(refLen, fl > 0 ? fl - 1 : 0), effLength = 0.0;
    while (!done) {
      uint32_t slot;
      if (fl >= maxLen) { done = true; fl = maxLen - 1; slot = n_grid; } else slot = (uint32_t)((fl - fldLow) / samp);
      const double w = cond_cdf(refLen, fl) - prev; prev = cond_cdf(refLen, fl);
      double tot = 0.0;
      if (fl >= 1 && fl <= refLen) { const uint32_t* h = &hist[(p * nslots + slot) * SQ_GC_FRAG_BINS]; for (int b = 0; b < SQ_GC_FRAG_BINS; ++b) tot += (double)h[b] * bias[0][b]; }
      effLength += w * tot;
      fl += samp;
    }
    const double offset = std::max(1.0, (double)unproc[t]), noBias = (double)elen[t];
    eff_out[t] = std::max(effLength, std::min(noBias, offset));                                          // barrierLength (:1957-1965)
  }
  if (rep) { rep->num_processed = (uint32_t)P; rep->fld_low = fldLow; rep->fld_high = fldHigh; for (int b = 0; b < SQ_GC_FRAG_BINS; ++b) rep->gc_bias_row0[b] = bias[0][b]; }
  return SQ_OK;
}

// ================================================================================================================================
// --seqBias (alone or with --gcBias): read-start context models (reference src/model/SBModel.cpp) and the bias-corrected effective
// lengths of salmon::utils::updateEffectiveLengths (SalmonUtils.cpp:1208-1985, the seqBiasCorrect branches).  SPEC §B2 fixes the
// order of every floating-point sum so that the result does not depend on the launch geometry:
//   expected context models   per transcript: integer counts of the starts whose conditional-CDF factor is exactly 1 + the other
//                             (<= 1000) factors added in start order by one lane per model column; across transcripts: canonical sums
//   expected GC model         integer windows per (sampled length, context bin, GC bin)
//   effective length          per sampled length the lane-strided sum over fragment starts (256 lanes, strided-halving tree)
namespace {
#define SB_K 9
#define SB_LEFT 3
__device__ __host__ inline uint32_t sb_cell(uint32_t v, int i) {   // SBModel _shifts / _widths (:44-52): orders {0,1,2,2,2,2,2,2,2}
  const int order = i == 0 ? 0 : (i == 1 ? 1 : 2); const int shift = 2 * SB_K - 2 * (i + 1), width = 2 * (order + 1);
  return (uint32_t)i * 64u + ((v >> shift) & ((1u << width) - 1u));
}
__device__ inline uint32_t sb_ctx(const uint64_t* refseq, uint64_t g, int32_t p) {   // 9 bases from p, first base in the high bits
  const uint64_t w = sq_fetch_bases(refseq, g + (uint64_t)p, SB_K); uint32_t v = 0;
#pragma unroll
  for (int i = 0; i < SB_K; ++i) v = (v << 2) | (uint32_t)((w >> (2 * i)) & 3u);
  return v;
}
__device__ inline uint32_t sb_rc(uint32_t v) { uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < SB_K; ++i) { r = (r << 2) | (3u - (v & 3u)); v >>= 2; } return r; }

// block per processed transcript: out[p][0..575] = weight * (count + fraction) of the forward contexts, [576..1151] of the reverse ones
__global__ void __launch_bounds__(256) k_seq_expect(const uint64_t* __restrict__ refseq, const uint64_t* __restrict__ ref_accum, const uint32_t* __restrict__ ref_len,
                                                    const uint32_t* __restrict__ list, const double* __restrict__ weight, const double* __restrict__ cdf /*[1001]*/,
                                                    double* __restrict__ out) {
  __shared__ uint32_t cnt[2][576]; __shared__ double frac[2][576];
  const uint32_t p = blockIdx.x, t = list[p]; const int32_t refLen = (int32_t)ref_len[t]; const uint64_t g = ref_accum[t];
  for (int i = threadIdx.x; i < 1152; i += 256) { (&cnt[0][0])[i] = 0; (&frac[0][0])[i] = 0.0; }
  __syncthreads();
  const int32_t cdfMaxArg = refLen < 1000 ? refLen : 1000; const double cdfMaxVal = cdf[cdfMaxArg];
  const int32_t nstart = refLen - SB_K;                       // fragStartPos in [0, nstart)
  // maxFragLen = refLen - (fsp + 3) > cdfMaxArg  <=>  fsp < refLen - 3 - cdfMaxArg
  int32_t tail0 = refLen - SB_LEFT - cdfMaxArg; if (tail0 < 0) tail0 = 0; if (tail0 > nstart) tail0 = nstart < 0 ? 0 : nstart;
  for (int32_t fsp = (int32_t)threadIdx.x; fsp < tail0; fsp += 256) {
    const uint32_t fw = sb_ctx(refseq, g, fsp), rc = sb_rc(sb_ctx(refseq, g, refLen - SB_K - fsp));
#pragma unroll
    for (int i = 0; i < SB_K; ++i) { atomicAdd(&cnt[0][sb_cell(fw, i)], 1u); atomicAdd(&cnt[1][sb_cell(rc, i)], 1u); }
  }
  if (threadIdx.x < 2 * SB_K) {   // one lane per (strand, model column): the fractional factors in start order
    const int strand = threadIdx.x / SB_K, col = threadIdx.x % SB_K;
    for (int32_t fsp = tail0; fsp < nstart; ++fsp) {
      const int32_t x = refLen - (fsp + SB_LEFT);
      const uint32_t v = strand == 0 ? sb_ctx(refseq, g, fsp) : sb_rc(sb_ctx(refseq, g, refLen - SB_K - fsp));
      frac[strand][sb_cell(v, col)] += cdf[x] / cdfMaxVal;
    }
  }
  __syncthreads();
  const double w = weight[p];
  for (int i = threadIdx.x; i < 1152; i += 256) out[(size_t)p * 1152 + i] = w * ((double)(&cnt[0][0])[i] + (&frac[0][0])[i]);
}
// one level of the canonical sum (SPEC §D2) over the rows of X[n][ncol]: out[g][c] = strided-halving tree of rows 64 g .. 64 g + 63
__global__ void k_canon_level(const double* __restrict__ X, uint64_t n, uint32_t ncol, double* __restrict__ out) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; const uint64_t ng = (n + 63) / 64;
  if (gid >= ng * ncol) return;
  const uint64_t gq = gid / ncol; const uint32_t c = (uint32_t)(gid % ncol);
  double v[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) { const uint64_t r = gq * 64 + (uint64_t)i; v[i] = r < n ? X[r * ncol + c] : 0.0; }
#pragma unroll
  for (int st = 32; st >= 1; st >>= 1)
#pragma unroll
    for (int i = 0; i < st; ++i) v[i] = v[i] + v[i + st];
  out[gq * ncol + c] = v[0];
}

// GC of [s, e] and the context fraction of (s, e): Transcript::gcFrac and populateContextCounts (SalmonUtils.cpp:1372-1424) in closed form —
// count(it) = G(it + 1) - G(it - 4) while the window has not reached the last base, G(n-1) + (it + 2 - n) gc(n-1) - G(it - 4) after
// (the loop as written keeps adding the last base); window length it + 2 for it <= 3, else 5, else n + 3 - it.  cFP[f] = count(f),
// cTP[t] = count(t + 2), likewise the window lengths.
struct GcCtx { const uint64_t* refseq; const uint32_t* gcpre; uint64_t g; int32_t n; uint64_t g0; int32_t last_gc; };
__device__ inline int64_t gc_G(const GcCtx& C, int32_t i) { if (i < 0) return 0; if (i > C.n - 1) i = C.n - 1; return (int64_t)(sq_gc_before(C.refseq, C.gcpre, C.g + (uint64_t)i + 1) - C.g0); }
__device__ inline int32_t ctx_count(const GcCtx& C, int32_t it) { return (it + 1 <= C.n - 1) ? (int32_t)(gc_G(C, it + 1) - gc_G(C, it - 4)) : (int32_t)(gc_G(C, C.n - 1) + (int64_t)(it + 2 - C.n) * C.last_gc - gc_G(C, it - 4)); }
__device__ inline int32_t ctx_wl(const GcCtx& C, int32_t it) { return it <= 3 ? it + 2 : ((it + 1 <= C.n - 1) ? 5 : C.n + 3 - it); }
__device__ inline int32_t ctx_frac(const GcCtx& C, int32_t s, int32_t e) {
  if (C.n <= 5) return 0;
  const double cl = (double)(ctx_wl(C, s) + ctx_wl(C, e + 2));
  return cl > 0 ? (int32_t)rint(100.0 * (double)(ctx_count(C, s) + ctx_count(C, e + 2)) / cl) : 0;
}
__device__ inline GcCtx make_gcctx(const uint64_t* refseq, const uint32_t* gcpre, uint64_t g, int32_t n) {
  GcCtx C; C.refseq = refseq; C.gcpre = gcpre; C.g = g; C.n = n; C.g0 = sq_gc_before(refseq, gcpre, g);
  const uint32_t b = n > 0 ? sq_fetch_base(refseq, g + (uint64_t)n - 1) : 0u; C.last_gc = (b == 1 || b == 2) ? 1 : 0; return C;
}
// expected GC model with context bins: hist[p][slot][ctx * 25 + bin] = starts s in [0, refLen - K) with s + fl - 1 < refLen (the loop of :1577-1623)
__global__ void __launch_bounds__(256) k_gc_hist_ctx(const uint64_t* __restrict__ refseq, const uint32_t* __restrict__ gcpre, const uint64_t* __restrict__ ref_accum,
                                                     const uint32_t* __restrict__ ref_len, const uint32_t* __restrict__ list, int32_t fld_low_g, int32_t fld_high_g, int32_t samp,
                                                     uint32_t nslots, uint32_t* __restrict__ hist, int32_t K /* 9 with --seqBias, else 1 */, int use_ctx /* contexts only with --seqBias (:1566) */) {
  __shared__ uint32_t h[75];
  const uint32_t p = blockIdx.x, t = list[p]; const int32_t refLen = (int32_t)ref_len[t]; const uint64_t g = ref_accum[t];
  const GcCtx C = make_gcctx(refseq, gcpre, g, refLen);
  const int32_t cdfMaxArg = refLen < 1000 ? refLen : 1000;
  const int32_t lo = (refLen < cdfMaxArg) ? 1 : fld_low_g, hi = (refLen < cdfMaxArg) ? cdfMaxArg : fld_high_g;   // refLen < cdfMaxArg never holds; kept as the reference writes it
  for (uint32_t slot = 0; slot < nslots; ++slot) {
    const int32_t fl = lo + samp * (int32_t)slot;
    if (threadIdx.x < 75) h[threadIdx.x] = 0;
    __syncthreads();
    if (fl <= hi && fl >= 1 && fl <= refLen) {
      int32_t nst = refLen - K; const int32_t lim = refLen - fl + 1; if (lim < nst) nst = lim;      // s < refLen - K and s + fl - 1 < refLen
      for (int32_t s = (int32_t)threadIdx.x; s < nst; s += 256) {
        const int32_t e = s + fl - 1;
        const uint64_t c = sq_gc_before(refseq, gcpre, g + (uint64_t)e + 1) - sq_gc_before(refseq, gcpre, g + (uint64_t)s);
        const int32_t frac = (int32_t)rint((100.0 * (double)c) / (double)fl);
        atomicAdd(&h[sq_gc_ctx_bin(use_ctx ? ctx_frac(C, s, e) : 0) * SQ_GC_FRAG_BINS + sq_gc_frag_bin(frac)], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < 75) hist[((size_t)p * nslots + slot) * 75 + threadIdx.x] = h[threadIdx.x];
    __syncthreads();
  }
}
// per position of a transcript: seqFW[readStart] = exp(obs5 - exp5) of the forward context, seqRC (already reversed into 5'->3' order)
__device__ inline double sb_eval(const double* __restrict__ logp, uint32_t v) { double p = 0.0;
#pragma unroll
  for (int i = 0; i < SB_K; ++i) p += logp[sb_cell(v, i)]; return p; }
__global__ void k_seq_factors(const uint64_t* __restrict__ refseq, const uint64_t* __restrict__ ref_accum, const uint32_t* __restrict__ ref_len, const uint32_t* __restrict__ list,
                              const uint64_t* __restrict__ foff /* start of transcript p's factors */, const double* __restrict__ models /* exp_fw, exp_rc, obs_fw, obs_rc */,
                              double* __restrict__ sfw, double* __restrict__ src) {
  const uint32_t p = blockIdx.y, t = list[p]; const int32_t refLen = (int32_t)ref_len[t]; const uint64_t g = ref_accum[t], o = foff[p];
  for (int32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < refLen; j += gridDim.x * blockDim.x) {
    // forward factor at position j: set by fs = j - 3 when 0 <= fs < refLen - K
    double f = 1.0; { const int32_t fs = j - SB_LEFT; if (fs >= 0 && fs < refLen - SB_K) { const uint32_t v = sb_ctx(refseq, g, fs); f = sq_exp(sb_eval(models + 1152, v) - sb_eval(models, v)); } }
    sfw[o + (uint64_t)j] = f;
    // reverse factor at 5'->3' position j = the value the loop stores at rc position refLen - 1 - j, i.e. fs = refLen - 1 - j - 3
    double r = 1.0; { const int32_t fs = refLen - 1 - j - SB_LEFT; if (fs >= 0 && fs < refLen - SB_K) { const uint32_t v = sb_rc(sb_ctx(refseq, g, refLen - SB_K - fs)); r = sq_exp(sb_eval(models + 1728, v) - sb_eval(models + 576, v)); } }
    src[o + (uint64_t)j] = r;
  }
}
struct EffArgs { int32_t fld_low, fld_high, samp; int use_gc, use_ctx; double bias[75]; };
// ---- --posBias (SalmonUtils.cpp:1639-1652, 1815-1835, 1941-1944; SimplePosBias.cpp) --------------------------------------------------
// the bin of read start p on a transcript of length len: SimplePosBias::addMass(pos, length, .)
__device__ inline int pos_bin_dev(int32_t p, int32_t len) { const double step = (double)len / 20.0; const int b = (int)floor((double)p / step); return b > 19 ? 19 : b; }
// expected read-start models: block per processed transcript; for every bin the lane-strided sum over its starts s < refLen - K of
// weight * conditionalCDF(refLen - s + 1) (5' model) and weight * conditionalCDF(s) (3' model), terms <= EPSILON dropped;
// x[p][dir * 100 + class * 20 + bin], zero elsewhere (the canonical sum over the transcripts follows: k_canon_level)
__global__ void __launch_bounds__(256) k_pos_expect(const uint32_t* __restrict__ ref_len, const uint32_t* __restrict__ list, const double* __restrict__ weight,
                                                    const double* __restrict__ cdf, const uint8_t* __restrict__ lenclass, int32_t K, double* __restrict__ x /*[P][200]*/) {
  __shared__ double v5[256], v3[256]; __shared__ int32_t lo[21];
  const uint32_t p = blockIdx.x, t = list[p]; const int32_t refLen = (int32_t)ref_len[t], ns = refLen - K; const double w = weight[p];
  const int32_t cdfMaxArg = refLen < 1000 ? refLen : 1000; const double cdfMaxVal = cdf[cdfMaxArg];
  auto cCDF = [&](int32_t a) { return a > cdfMaxArg ? 1.0 : cdf[a] / cdfMaxVal; };
  for (int i = (int)threadIdx.x; i < 200; i += 256) x[(size_t)p * 200 + i] = 0.0;
  if (threadIdx.x <= 20) {   // lo[b] = the first start whose bin is >= b (bins grow with the start): an estimate, then a short walk
    const int b = (int)threadIdx.x; int32_t s = (int32_t)((double)b * ((double)refLen / 20.0)) - 2; if (s < 0) s = 0;
    while (s > 0 && pos_bin_dev(s - 1, refLen) >= b) --s;
    while (s < refLen && pos_bin_dev(s, refLen) < b) ++s;
    lo[b] = b == 20 ? refLen : s;
  }
  __syncthreads();
  const uint32_t li = lenclass[t];
  for (int b = 0; b < 20; ++b) {
    const int32_t s0 = lo[b] < ns ? lo[b] : ns, s1 = lo[b + 1] < ns ? lo[b + 1] : ns;
    double a5 = 0.0, a3 = 0.0;
    for (int32_t s = s0 + (int32_t)threadIdx.x; s < s1; s += 256) {
      const double f5 = w * cCDF(refLen - s + 1), f3 = w * cCDF(s);
      a5 += f5 > 0.375e-10 ? f5 : 0.0; a3 += f3 > 0.375e-10 ? f3 : 0.0;
    }
    v5[threadIdx.x] = a5; v3[threadIdx.x] = a3;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) { if ((int)threadIdx.x < st) { v5[threadIdx.x] = v5[threadIdx.x] + v5[threadIdx.x + st]; v3[threadIdx.x] = v3[threadIdx.x] + v3[threadIdx.x + st]; } __syncthreads(); }
    if (threadIdx.x == 0 && s1 > s0) { x[(size_t)p * 200 + li * 20 + b] = v5[0]; x[(size_t)p * 200 + 100 + li * 20 + b] = v3[0]; }
    __syncthreads();
  }
}
// tk::spline::operator() inside the knots + projectWeights' floor (SimplePosBias.cpp:32-39)
__device__ inline double pos_weight_dev(const sq_pos_spline& S, int32_t p, int32_t len) {
  const double f = (double)p / (double)len;
  int idx = 0; while (idx < SQ_POS_KNOTS && S.x[idx] < f) ++idx;   // lower_bound: the first knot >= f
  idx = idx == 0 ? 0 : idx - 1;
  const double h = f - S.x[idx], r = ((S.a[idx] * h + S.b[idx]) * h + S.c[idx]) * h + S.y[idx];
  return r > 0.001 ? r : 0.001;
}
// posFactorsFW / posFactorsRC (:1829-1834): observed over expected weight at every start < refLen - K, 1 elsewhere
__global__ void k_pos_factors(const uint32_t* __restrict__ ref_len, const uint32_t* __restrict__ list, const uint64_t* __restrict__ foff, const uint8_t* __restrict__ lenclass,
                              const sq_pos_spline* __restrict__ sp /* [obs5, obs3, exp5, exp3][class] */, int32_t K, double* __restrict__ pfw, double* __restrict__ prc) {
  __shared__ sq_pos_spline S[4];
  const uint32_t p = blockIdx.y, t = list[p]; const int32_t refLen = (int32_t)ref_len[t]; const uint64_t o = foff[p]; const uint32_t li = lenclass[t];
  for (int i = (int)threadIdx.x; i < (int)(4 * sizeof(sq_pos_spline) / 8); i += (int)blockDim.x) {
    const int m = i / (int)(sizeof(sq_pos_spline) / 8), j = i % (int)(sizeof(sq_pos_spline) / 8);
    ((double*)&S[m])[j] = ((const double*)&sp[m * SQ_POS_CLASSES + li])[j];
  }
  __syncthreads();
  for (int32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < refLen; j += gridDim.x * blockDim.x) {
    double f = 1.0, r = 1.0;
    if (j < refLen - K) { f = pos_weight_dev(S[0], j, refLen) / pos_weight_dev(S[2], j, refLen); r = pos_weight_dev(S[1], j, refLen) / pos_weight_dev(S[3], j, refLen); }
    pfw[o + (uint64_t)j] = f; prc[o + (uint64_t)j] = r;
  }
}
// block per transcript: the effective length loop of :1888-1945 with the lane-strided sum over fragment starts
__global__ void __launch_bounds__(256) k_seq_efflen(const uint64_t* __restrict__ refseq, const uint32_t* __restrict__ gcpre, const uint64_t* __restrict__ ref_accum,
                                                    const uint32_t* __restrict__ ref_len, const uint32_t* __restrict__ list, const uint64_t* __restrict__ foff,
                                                    const double* __restrict__ sfw, const double* __restrict__ src /* null without --seqBias */, const double* __restrict__ pfw,
                                                    const double* __restrict__ prc /* null without --posBias */, const double* __restrict__ cdf, EffArgs A, double* __restrict__ eff /*[P]*/) {
  __shared__ double v[256];
  const uint32_t p = blockIdx.x, t = list[p]; const int32_t refLen = (int32_t)ref_len[t]; const uint64_t g = ref_accum[t], o = foff[p];
  const GcCtx C = make_gcctx(refseq, gcpre, g, refLen);
  const int32_t cdfMaxArg = refLen < 1000 ? refLen : 1000; const double cdfMaxVal = cdf[cdfMaxArg];
  const int32_t lo = (refLen < cdfMaxArg) ? 1 : A.fld_low, hi = (refLen < cdfMaxArg) ? cdfMaxArg : A.fld_high;
  int32_t fl = lo; const int32_t maxLen = refLen < hi + 1 ? refLen : hi + 1; bool done = fl >= maxLen;
  auto cCDF = [&](int32_t x) { return x > cdfMaxArg ? 1.0 : cdf[x] / cdfMaxVal; };
  double prev = cCDF(fl > 0 ? fl - 1 : 0), effLength = 0.0;
  while (!done) {
    if (fl >= maxLen) { done = true; fl = maxLen - 1; }
    const double flWeight = cCDF(fl) - prev; prev = cCDF(fl);
    const int32_t ns = refLen - fl > 0 ? refLen - fl : 0;
    double a = 0.0;
    for (int32_t s = (int32_t)threadIdx.x; s < ns; s += 256) {
      const int32_t e = s + fl - 1;
      double f = sfw ? sfw[o + (uint64_t)s] * src[o + (uint64_t)e] : 1.0;
      if (A.use_gc) {
        const uint64_t c = sq_gc_before(refseq, gcpre, g + (uint64_t)e + 1) - sq_gc_before(refseq, gcpre, g + (uint64_t)s);
        const int32_t frac = (int32_t)rint((100.0 * (double)c) / (double)fl);
        f *= A.bias[sq_gc_ctx_bin(A.use_ctx ? ctx_frac(C, s, e) : 0) * SQ_GC_FRAG_BINS + sq_gc_frag_bin(frac)];
      }
      if (pfw) f *= pfw[o + (uint64_t)s] * prc[o + (uint64_t)e];
      a += f;
    }
    v[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) { if ((int)threadIdx.x < st) v[threadIdx.x] = v[threadIdx.x] + v[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) effLength += flWeight * v[0];
    __syncthreads();
    fl += A.samp;
  }
  if (threadIdx.x == 0) eff[p] = effLength;
}
void sb_normalize_host(const double* counts, double* logp) {   // SBModel::normalize (:216-252)
  for (int i = 0; i < SB_K; ++i) {
    const int order = i == 0 ? 0 : (i == 1 ? 1 : 2), nstates = 1 << (2 * order);
    for (int c = 0; c < 64; ++c) logp[i * 64 + c] = 0.0;
    for (int gq = 0; gq < nstates; ++gq) {
      const double* q = counts + i * 64 + 4 * gq; const double tot = ((q[0] + q[1]) + q[2]) + q[3];
      for (int b = 0; b < 4; ++b) { const double pr = q[b] / tot; logp[i * 64 + 4 * gq + b] = pr > 0.0 ? sq_log(pr) : sq_log(1e-5); }
    }
  }
}
}  // namespace

// the per-position sweep every combination with --seqBias or --posBias takes (--gcBias alone: sq_bias_gc_eff_lengths' histograms)
static int bias_sweep(sq_index* idx, int use_gc, const double* gc_obs, const uint64_t* seq_fw, const uint64_t* seq_rc, const double* pos_obs, uint32_t threads,
                      const double* log_pmf, uint32_t M, const double* alphas, const double* eff_in, double* eff_out, double* models_out /* [4][576] or NULL */,
                      double* pos_models_out /* [4][100] or NULL */, sq_bias_report* rep) {
  const bool use_seq = seq_fw != nullptr, use_pos = pos_obs != nullptr; const int32_t K = use_seq ? SB_K : 1;
  if (!idx || (use_seq && !seq_rc) || !log_pmf || !alphas || !eff_in || !eff_out || (use_gc && !gc_obs) || (!use_seq && !use_pos)) { sq_set_error("sq_bias_eff_lengths: bad arguments"); return SQ_ERR_ARG; }
  if (!idx->dev) { sq_set_error("sq_bias_eff_lengths: the index is not on a device (there is no CPU path)"); return SQ_ERR_DEVICE; }
  if (M > idx->names.size()) { sq_set_error("sq_bias_eff_lengths: %u transcripts but the index has %zu", M, idx->names.size()); return SQ_ERR_ARG; }
  const sq_device_index* di = idx->dev;
  SQ_HIP_CHECK(hipSetDevice(di->device));
  const int MAXV = 1000; const int32_t samp = 5;
  std::vector<double> pdf(MAXV + 1), cdf(MAXV + 1); int32_t fldLow = 0, fldHigh = 1; bool lb = false, ub = false;
  for (int i = 0; i <= MAXV; ++i) {
    pdf[i] = sq_exp(log_pmf[i]); cdf[i] = i > 0 ? cdf[i - 1] + pdf[i] : pdf[i];
    if (!lb && cdf[i] >= 0.005) { lb = true; fldLow = i; }
    if (!ub && cdf[i] >= 1.0 - 0.005) { ub = true; fldHigh = i; }
  }
  std::vector<uint32_t> list; std::vector<int32_t> elen(M), unproc(M); std::vector<double> weight;
  for (uint32_t t = 0; t < M; ++t) {
    const int32_t refLen = (int32_t)idx->ref_len[t]; elen[t] = (int32_t)eff_in[t]; unproc[t] = std::max(0, refLen - elen[t]);
    if (cdf[std::min(MAXV, refLen)] < 1e-10 || alphas[t] < 1e-8 || unproc[t] <= 0) continue;
    list.push_back(t); weight.push_back(alphas[t] / eff_in[t]);
  }
  const size_t P = list.size();
  for (uint32_t t = 0; t < M; ++t) eff_out[t] = (double)elen[t];
  double models[4 * 576]; double bias[75]; for (double& b : bias) b = 1.0;
  double cnt[4][576];
  for (int c = 0; c < 576; ++c) { cnt[0][c] = 1e-10; cnt[1][c] = 1e-10; cnt[2][c] = 1e-10 + (use_seq ? (double)seq_fw[c] : 0.0); cnt[3][c] = 1e-10 + (use_seq ? (double)seq_rc[c] : 0.0); }
  sq_dbuf<uint32_t> d_list; sq_dbuf<double> d_w, d_cdf, d_x, d_y, d_models, d_sfw, d_src, d_pfw, d_prc, d_eff; sq_dbuf<uint64_t> d_foff; sq_dbuf<uint32_t> d_hist; sq_dbuf<uint8_t> d_cls; sq_dbuf<sq_pos_spline> d_sp;
  auto release = [&]() { d_list.free_(); d_w.free_(); d_cdf.free_(); d_x.free_(); d_y.free_(); d_models.free_(); d_sfw.free_(); d_src.free_(); d_pfw.free_(); d_prc.free_(); d_eff.free_(); d_foff.free_(); d_hist.free_(); d_cls.free_(); d_sp.free_(); };
  struct Guard { decltype(release)& r; ~Guard() { r(); } } guard{release};
  if (P) {
    const size_t xw = use_seq ? 1152 : 200;
    if (d_list.ensure(P) || d_w.ensure(P) || d_cdf.ensure(MAXV + 1) || d_x.ensure(P * xw) || d_y.ensure(((P + 63) / 64) * xw + xw) || d_models.ensure(4 * 576)) {
      sq_set_error("device allocation failed (sequence-bias models for %zu transcripts)", P); return SQ_ERR_NOMEM; }
    SQ_HIP_CHECK(hipMemcpy(d_list.p, list.data(), P * 4, hipMemcpyHostToDevice));
    SQ_HIP_CHECK(hipMemcpy(d_w.p, weight.data(), P * 8, hipMemcpyHostToDevice));
    SQ_HIP_CHECK(hipMemcpy(d_cdf.p, cdf.data(), (MAXV + 1) * 8, hipMemcpyHostToDevice));
    // ---- expected context models: per-transcript terms, then the canonical sum over the processed transcripts ----
    if (use_seq) { k_seq_expect<<<(uint32_t)P, 256>>>(di->refseq, di->ref_accum, di->ref_len, d_list.p, d_w.p, d_cdf.p, d_x.p);
      double* in = d_x.p; double* out = d_y.p; uint64_t n = P;
      while (n > 1) { const uint64_t ng = (n + 63) / 64; k_canon_level<<<(uint32_t)((ng * 1152 + 255) / 256), 256>>>(in, n, 1152, out); std::swap(in, out); n = ng; }
      double e[1152]; SQ_HIP_CHECK(hipMemcpy(e, in, sizeof(e), hipMemcpyDeviceToHost));
      for (int c = 0; c < 576; ++c) { cnt[0][c] = 1e-10 + e[c]; cnt[1][c] = 1e-10 + e[576 + c]; } }
  }
  // ---- --posBias: expected read-start models, then the four splines per length class ----
  if (use_pos) {
    double expect[200] = {0};
    std::vector<uint8_t> cls; uint32_t quant[SQ_POS_CLASSES]; sq_pos_length_classes(idx, quant, cls);
    if (d_cls.ensure(cls.size() + 8) || d_sp.ensure(4 * SQ_POS_CLASSES)) { sq_set_error("device allocation failed (positional models)"); return SQ_ERR_NOMEM; }
    SQ_HIP_CHECK(hipMemcpy(d_cls.p, cls.data(), cls.size(), hipMemcpyHostToDevice));
    if (P) {
      k_pos_expect<<<(uint32_t)P, 256>>>(di->ref_len, d_list.p, d_w.p, d_cdf.p, d_cls.p, K, d_x.p);
      double* in = d_x.p; double* out = d_y.p; uint64_t n = P;
      while (n > 1) { const uint64_t ng = (n + 63) / 64; k_canon_level<<<(uint32_t)((ng * 200 + 255) / 256), 256>>>(in, n, 200, out); std::swap(in, out); n = ng; }
      SQ_HIP_CHECK(hipMemcpy(expect, in, sizeof(expect), hipMemcpyDeviceToHost));
    }
    // every bin of the shared model and of each of the T workers' local models starts with mass 1 (LOG_1): BiasParams.hpp:41, WorkerRuntimeContext.hpp:40-45; the
    // expected side likewise (CombineableBiasParams :1319-1320 per thread + the library's model)
    const double prior = 1.0 + (double)threads;
    sq_pos_spline sp[4 * SQ_POS_CLASSES]; double norm[4][100];
    for (int li = 0; li < SQ_POS_CLASSES; ++li) {
      double m[4][SQ_POS_BINS];
      for (int b = 0; b < SQ_POS_BINS; ++b) { m[0][b] = prior + pos_obs[li * 20 + b]; m[1][b] = prior + pos_obs[100 + li * 20 + b]; m[2][b] = prior + expect[li * 20 + b]; m[3][b] = prior + expect[100 + li * 20 + b]; }
      for (int k = 0; k < 4; ++k) sq_pos_finalize(m[k], &sp[k * SQ_POS_CLASSES + li], &norm[k][li * 20]);
    }
    if (pos_models_out) memcpy(pos_models_out, norm, sizeof(norm));
    SQ_HIP_CHECK(hipMemcpy(d_sp.p, sp, sizeof(sp), hipMemcpyHostToDevice));
  }
  if (use_seq) for (int m = 0; m < 4; ++m) sb_normalize_host(cnt[m], models + m * 576); else memset(models, 0, sizeof(models));
  if (models_out) memcpy(models_out, models, sizeof(models));
  if (P) {
    SQ_HIP_CHECK(hipMemcpy(d_models.p, models, sizeof(models), hipMemcpyHostToDevice));
    // ---- expected GC model with context bins ----
    if (use_gc) {
      const uint32_t nslots = fldHigh >= fldLow ? (uint32_t)((fldHigh - fldLow) / samp + 1) : 0u;
      std::vector<uint32_t> hist(P * (size_t)nslots * 75, 0);
      if (nslots) {
        if (d_hist.ensure(hist.size())) { sq_set_error("device allocation failed (GC histograms)"); return SQ_ERR_NOMEM; }
        k_gc_hist_ctx<<<(uint32_t)P, 256>>>(di->refseq, di->gcpre, di->ref_accum, di->ref_len, d_list.p, fldLow, fldHigh, samp, nslots, d_hist.p, K, use_seq ? 1 : 0);
        SQ_HIP_CHECK(hipMemcpy(hist.data(), d_hist.p, hist.size() * 4, hipMemcpyDeviceToHost));
      }
      auto cond_cdf = [&](int32_t refLen, int32_t x) { const int32_t a = std::min(MAXV, refLen); return x > a ? 1.0 : cdf[x] / cdf[a]; };
      std::vector<std::vector<double>> contrib(75, std::vector<double>(P, 0.0));
      for (size_t p = 0; p < P; ++p) {
        const uint32_t t = list[p]; const int32_t refLen = (int32_t)idx->ref_len[t]; double E[75] = {0};
        double prev = cond_cdf(refLen, fldLow > 0 ? fldLow - 1 : 0);
        for (uint32_t j = 0; j < nslots; ++j) {
          const int32_t fl = fldLow + samp * (int32_t)j;
          if (fl > refLen || fl < 1) break;
          const double d = cond_cdf(refLen, fl) - prev; prev = cond_cdf(refLen, fl);
          const uint32_t* h = &hist[(p * nslots + j) * 75];
          for (int b = 0; b < 75; ++b) E[b] += d * (double)h[b];
        }
        for (int b = 0; b < 75; ++b) contrib[b][p] = weight[p] * E[b];
      }
      double expect[75]; for (int b = 0; b < 75; ++b) expect[b] = canonical_sum_vec(contrib[b]);
      memcpy(tl_gc_expected, expect, sizeof(tl_gc_expected)); tl_gc_expected_rows = SQ_GC_COND_BINS;
      auto normalize = [](const double* in, double* out) { double row = 0.0; for (int b = 0; b < SQ_GC_FRAG_BINS; ++b) row += (0.1 + in[b]);
        if (row > 0.0) { const double nrm = 1.0 / row; for (int b = 0; b < SQ_GC_FRAG_BINS; ++b) out[b] = (0.1 + in[b]) * nrm; } else for (int b = 0; b < SQ_GC_FRAG_BINS; ++b) out[b] = in[b]; };
      for (int r = 0; r < SQ_GC_COND_BINS; ++r) { double on[25], en[25]; normalize(gc_obs + r * 25, on); normalize(expect + r * 25, en);
        for (int b = 0; b < 25; ++b) { double rat = on[b] / en[b]; if (rat > 1000.0) rat = 1000.0; if (rat < 1.0 / 1000.0) rat = 1.0 / 1000.0; bias[r * 25 + b] = rat; } }
    }
    // ---- effective lengths, in groups of transcripts whose factors fit the scratch ----
    EffArgs A; A.fld_low = fldLow; A.fld_high = fldHigh; A.samp = samp; A.use_gc = use_gc ? 1 : 0; A.use_ctx = use_seq ? 1 : 0; memcpy(A.bias, bias, sizeof(bias));
    const uint64_t CH = 64ull << 20;   // positions per group
    std::vector<double> effh;
    for (size_t p0 = 0; p0 < P;) {
      size_t p1 = p0; uint64_t tot = 0; std::vector<uint64_t> foff;
      while (p1 < P && (p1 == p0 || tot + idx->ref_len[list[p1]] <= CH)) { foff.push_back(tot); tot += idx->ref_len[list[p1]]; ++p1; }
      const size_t np = p1 - p0;
      if ((use_seq && (d_sfw.ensure(tot + 8) || d_src.ensure(tot + 8))) || (use_pos && (d_pfw.ensure(tot + 8) || d_prc.ensure(tot + 8))) || d_foff.ensure(np) || d_eff.ensure(np)) {
        sq_set_error("device allocation failed (per-position bias factors)"); return SQ_ERR_NOMEM; }
      SQ_HIP_CHECK(hipMemcpy(d_foff.p, foff.data(), np * 8, hipMemcpyHostToDevice));
      if (use_seq) k_seq_factors<<<dim3(8, (uint32_t)np), 256>>>(di->refseq, di->ref_accum, di->ref_len, d_list.p + p0, d_foff.p, d_models.p, d_sfw.p, d_src.p);
      if (use_pos) k_pos_factors<<<dim3(8, (uint32_t)np), 256>>>(di->ref_len, d_list.p + p0, d_foff.p, d_cls.p, d_sp.p, K, d_pfw.p, d_prc.p);
      k_seq_efflen<<<(uint32_t)np, 256>>>(di->refseq, di->gcpre, di->ref_accum, di->ref_len, d_list.p + p0, d_foff.p, use_seq ? d_sfw.p : nullptr, use_seq ? d_src.p : nullptr,
                                          use_pos ? d_pfw.p : nullptr, use_pos ? d_prc.p : nullptr, d_cdf.p, A, d_eff.p);
      effh.resize(np); SQ_HIP_CHECK(hipMemcpy(effh.data(), d_eff.p, np * 8, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < np; ++i) { const uint32_t t = list[p0 + i];
        const double offset = std::max(1.0, (double)unproc[t]), noBias = (double)elen[t];
        eff_out[t] = std::max(effh[i], std::min(noBias, offset)); }
      p0 = p1;
    }
  }
  if (rep) { rep->num_processed = (uint32_t)P; rep->fld_low = fldLow; rep->fld_high = fldHigh; for (int b = 0; b < SQ_GC_FRAG_BINS; ++b) rep->gc_bias_row0[b] = bias[b]; }
  return SQ_OK;
}

extern "C" int sq_bias_seq_eff_lengths(sq_index* idx, int use_gc, const double* gc_obs, const uint64_t* seq_fw, const uint64_t* seq_rc, const double* log_pmf, uint32_t M,
                                       const double* alphas, const double* eff_in, double* eff_out, double* models_out /* [4][576] or NULL */, sq_bias_report* rep) {
  if (!seq_fw || !seq_rc) { sq_set_error("sq_bias_seq_eff_lengths: bad arguments"); return SQ_ERR_ARG; }
  return bias_sweep(idx, use_gc, gc_obs, seq_fw, seq_rc, nullptr, 0, log_pmf, M, alphas, eff_in, eff_out, models_out, nullptr, rep);
}
extern "C" int sq_bias_eff_lengths(sq_index* idx, const sq_bias_models* m, const double* log_pmf, uint32_t M, const double* alphas, const double* eff_in, double* eff_out,
                                   double* seq_models_out, double* pos_models_out, sq_bias_report* rep) {
  if (!m) { sq_set_error("sq_bias_eff_lengths: bad arguments"); return SQ_ERR_ARG; }
  if (!m->seq_fw && !m->pos_observed) {   // --gcBias alone: the histogram form
    if (!m->gc_observed) { sq_set_error("sq_bias_eff_lengths: no model given"); return SQ_ERR_ARG; }
    return sq_bias_gc_eff_lengths(idx, m->gc_observed, log_pmf, M, alphas, eff_in, eff_out, rep);
  }
  return bias_sweep(idx, m->gc_observed != nullptr, m->gc_observed, m->seq_fw, m->seq_rc, m->pos_observed, m->threads, log_pmf, M, alphas, eff_in, eff_out, seq_models_out, pos_models_out, rep);
}
extern "C" int sq_index_length_classes(const sq_index* idx, uint32_t* quantiles5, uint8_t* cls /* [num_refs] or NULL */) {
  if (!idx || !quantiles5) { sq_set_error("sq_index_length_classes: bad arguments"); return -1; }
  std::vector<uint8_t> c; const int n = sq_pos_length_classes(idx, quantiles5, c);
  if (cls) memcpy(cls, c.data(), c.size());
  return n;
}

const std::vector<uint32_t>& sq_index_gc_prefix(sq_index* idx) {
  if (idx->gcpre.size() != idx->refseq.size() + 1) {
    idx->gcpre.assign(idx->refseq.size() + 1, 0);
    uint64_t acc = 0;
    for (size_t w = 0; w < idx->refseq.size(); ++w) { idx->gcpre[w] = (uint32_t)acc; acc += sq_gc_word(idx->refseq[w]); }
    idx->gcpre[idx->refseq.size()] = (uint32_t)acc;
  }
  return idx->gcpre;
}
