// oracle/ref_fld_shim.cpp — TEST INFRASTRUCTURE.  C entry points around the reference's fragment-length distribution, compiled from where the source
// lies under /root/reference (never copied) into oracle/_ref/libfld_ref.so by oracle/Makefile:
//   src/model/FragmentLengthDistribution.cpp + include/salmon/internal/model/FragmentLengthDistribution.hpp
//   src/util/DistributionUtils.cpp: correctionFactorsFromMass + computeSmoothedEffectiveLengths (the effective lengths the online phase uses once it is
//     burned in), samplesFromLogPMF (frag_length_mean / frag_length_sd of meta_info.json), evaluateLogCMF + LogCMFCache (the fragment-length
//     probability of an orphan / single-end alignment)
// Boost (normal cdf, binomial pdf), RapMap's SpinLock and the heavy SalmonUtils.hpp are stood in for by oracle/_stub (each says what it replaces).
// Pins the checker's FLD (oracle.cpp: struct FLD — prior, kernel placement in addVal, pmf, cacheCMF / getLockedPMF, the minimum) — tests/test_fld_pin.py.
#include "salmon/internal/model/FragmentLengthDistribution.hpp"
#include "salmon/internal/model/Transcript.hpp"
#include "salmon/internal/util/DistributionUtils.hpp"
#include "salmon/internal/util/SalmonMath.hpp"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
extern "C" {
void* ref_fld_new(double alpha, uint64_t max_val, double mu, double sigma, uint64_t kernel_n, double kernel_p) { return new FragmentLengthDistribution(alpha, max_val, mu, sigma, kernel_n, kernel_p, 1); }
void ref_fld_free(void* h) { delete (FragmentLengthDistribution*)h; }
void ref_fld_add(void* h, const uint32_t* lens, uint64_t n, double log_mass) { auto* f = (FragmentLengthDistribution*)h; for (uint64_t i = 0; i < n; ++i) f->addVal(lens[i], log_mass); }
void ref_fld_cache(void* h) { ((FragmentLengthDistribution*)h)->cacheCMF(); }
void ref_fld_pmf(void* h, double* out, uint32_t n) { auto* f = (FragmentLengthDistribution*)h; for (uint32_t i = 0; i < n; ++i) out[i] = f->pmf(i); }
void ref_fld_cmf(void* h, double* out, uint32_t n) { auto* f = (FragmentLengthDistribution*)h; for (uint32_t i = 0; i < n; ++i) out[i] = f->cmf(i); }
uint64_t ref_fld_min(void* h) { return ((FragmentLengthDistribution*)h)->minVal(); }
uint64_t ref_fld_max(void* h) { return ((FragmentLengthDistribution*)h)->maxVal(); }
double ref_fld_mean(void* h) { return ((FragmentLengthDistribution*)h)->mean(); }
double ref_fld_tot(void* h) { return ((FragmentLengthDistribution*)h)->totMass(); }
// ReadExperiment::updateTranscriptLengthsAtomic (include/salmon/internal/quant/ReadExperiment.inl:62-94): the calls it makes, in its order, on the
// reference's own functions — dumpPMF, renormalise, 100 * exp, correctionFactorsFromMass, computeSmoothedEffectiveLengths(LOG)
void ref_eff_lengths(void* h, const uint32_t* ref_len, uint32_t n, double* log_eff_len) {
  auto* fld = (FragmentLengthDistribution*)h;
  std::vector<double> logPMF; size_t minVal, maxVal; fld->dumpPMF(logPMF, minVal, maxVal);
  double sum = salmon::math::LOG_0; for (auto v : logPMF) sum = salmon::math::logAdd(sum, v);
  for (auto& v : logPMF) v -= sum;
  std::vector<double> pmf(maxVal + 1, 0.0); for (size_t i = minVal; i < maxVal; ++i) pmf[i] = 100.0 * std::exp(logPMF[i - minVal]);
  using distribution_utils::DistributionSpace;
  auto cf = distribution_utils::correctionFactorsFromMass(pmf, DistributionSpace::LINEAR);
  std::vector<Transcript> ts(n); for (uint32_t i = 0; i < n; ++i) ts[i].RefLength = ref_len[i];
  distribution_utils::computeSmoothedEffectiveLengths(pmf.size(), ts, cf, DistributionSpace::LOG);
  for (uint32_t i = 0; i < n; ++i) log_eff_len[i] = ts[i].cachedLogEffLen;
}
void ref_fld_summary(void* h, double* mean, double* sd, uint32_t* support) {
  auto ds = distribution_utils::samplesFromLogPMF((FragmentLengthDistribution*)h, 0); *mean = ds.mean; *sd = ds.sd; *support = (uint32_t)ds.samples.size();
}
uint32_t ref_eval_log_cmf(void* h, double* out, uint32_t cap) {
  auto v = distribution_utils::evaluateLogCMF((FragmentLengthDistribution*)h); const uint32_t n = (uint32_t)std::min<size_t>(v.size(), cap); memcpy(out, v.data(), (size_t)n * 8); return (uint32_t)v.size();
}
// LogCMFCache as processMiniBatch uses it (SalmonQuantify.cpp:560-575, :640-650): refreshed at the start of a mini-batch, then asked per alignment
double ref_ambig_prob(void* h, int single_end_lib, int burned_in, int fwd, int32_t pos, int32_t rlen, int32_t tlen) {
  distribution_utils::LogCMFCache c((FragmentLengthDistribution*)h, single_end_lib != 0, 10000);
  c.refresh(1, burned_in != 0);
  return c.getAmbigFragLengthProb(fwd != 0, pos, rlen, tlen, burned_in != 0);
}
}

