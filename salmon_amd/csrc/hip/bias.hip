// hip/bias.hip — fragment-GC bias (row f-3, `--gcBias`): the expected GC model and the bias-corrected effective lengths of
// salmon::utils::updateEffectiveLengths (reference src/util/SalmonUtils.cpp:1208-1985, the gcBiasCorrect branches; GC model:
// include/salmon/internal/model/GCFragModel.hpp), called from inside the EM at iteration ~10 (CollapsedEMOptimizer.cpp:901-928).
//
// The reference sweeps, for every expressed transcript, every fragment start x every sampled fragment length (O(sum_t len_t * 25))
// twice, adding doubles into 25-bin models.  Here the GPU does the sweep ONCE as integers — a histogram of fragment-GC bins per
// (transcript, sampled length), read off a sampled G/C prefix of the 2-bit reference pool — and the floating-point part (25-term dot
// products in a fixed order) is a few hundred flops per transcript on the host: exact, order-defined, the same on every run (SPEC §B).
#include "ctx.h"
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

const std::vector<uint32_t>& sq_index_gc_prefix(sq_index* idx) {
  if (idx->gcpre.size() != idx->refseq.size() + 1) {
    idx->gcpre.assign(idx->refseq.size() + 1, 0);
    uint64_t acc = 0;
    for (size_t w = 0; w < idx->refseq.size(); ++w) { idx->gcpre[w] = (uint32_t)acc; acc += sq_gc_word(idx->refseq[w]); }
    idx->gcpre[idx->refseq.size()] = (uint32_t)acc;
  }
  return idx->gcpre;
}
