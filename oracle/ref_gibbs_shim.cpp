// oracle/ref_gibbs_shim.cpp — TEST INFRASTRUCTURE.  A C entry point around the reference's Gibbs sampler, compiled from where the source lies under
// /root/reference (never copied) into oracle/_ref/libgibbs_ref.so by oracle/Makefile:
//   src/inference/CollapsedGibbsSampler.cpp   CollapsedGibbsSampler::sample (:317-508: priors, chains, thinning, extrapolation of the counts) and
//                                             sampleRoundNonCollapsedMultithreaded_ (:92-278: the Gamma draw of every active transcript's abundance,
//                                             the multinomial of every class's reads) — row a17, the inference of configs[4]
// The file is #included; TBB (one thread's copy of everything: a legal schedule), Boost's irange / filesystem, spdlog, Eigen, ReadExperiment / Transcript /
// TranscriptGroup / SalmonOpts are stood in for by oracle/_stub/vbem as for the optimiser's pin.  The sampler seeds its generators from std::random_device:
// so that the pin is a deterministic test, the name is redirected to a counter-based source with the same interface, seeded by the caller.
// The checker draws from its own counter-based streams (SPEC §a17), so the comparison is one of DISTRIBUTIONS — per-transcript means and spreads over the
// samples — not of bits: tests/test_gibbs_pin.py.
#include <random>
#include <string>
#include <cstdint>
namespace std { struct sq_fixed_random_device { typedef unsigned int result_type; static uint64_t& state() { static uint64_t s = 1; return s; }
  sq_fixed_random_device() {} explicit sq_fixed_random_device(const std::string&) {} explicit sq_fixed_random_device(const char*) {}
  result_type operator()() { uint64_t z = (state() += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return (result_type)((z ^ (z >> 31)) >> 16); }
  static constexpr result_type min() { return 0; } static constexpr result_type max() { return 0xFFFFFFFFu; } double entropy() const { return 32.0; } }; }
#define random_device sq_fixed_random_device
#include "salmon/internal/quant/ReadExperiment.hpp"
#include "salmon/internal/util/SalmonUtils.hpp"
#include "src/inference/CollapsedGibbsSampler.cpp"
#include <cmath>
#include <functional>
#include <vector>
extern "C" int ref_gibbs(uint64_t E, const uint64_t* off, const uint32_t* tid, const double* w, const uint64_t* count, uint32_t M, const double* alpha_init, const double* eff_len,
                         uint64_t num_mapped, int use_vbem, int per_transcript_prior, double vb_prior, uint32_t thinning, int no_gamma_draw, int dont_extrapolate, uint32_t S, uint64_t seed, double* out) {
  std::sq_fixed_random_device::state() = seed;
  using ExpT = ReadExperiment<EquivalenceClassBuilder<TGValue>>;
  ExpT exp; exp.txps.resize(M); exp.mapped = num_mapped;
  for (uint32_t i = 0; i < M; ++i) { Transcript& t = exp.txps[i]; t.RefLength = (uint32_t)eff_len[i]; t.EffectiveLength = eff_len[i]; t.projectedCounts = alpha_init[i]; t.cachedLogEffLen = std::log(eff_len[i]); }
  auto& vec = exp.builder.vec; vec.reserve(E);
  for (uint64_t c = 0; c < E; ++c) { TranscriptGroup g; g.txps.assign(tid + off[c], tid + off[c + 1]); TGValue v; v.weights.assign(w + off[c], w + off[c + 1]); v.combinedWeights = v.weights; v.count = count[c]; vec.emplace_back(std::move(g), std::move(v)); }
  SalmonOpts so; so.useVBOpt = use_vbem != 0; so.perTranscriptPrior = per_transcript_prior != 0; so.vbPrior = vb_prior; so.thinningFactor = thinning; so.noGammaDraw = no_gamma_draw != 0; so.dontExtrapolateCounts = dont_extrapolate != 0;
  so.quiet = true; so.numThreads = 1; so.useQuasi = true; so.allowOrphans = true;
  uint32_t sid = 0;
  std::function<bool(const std::vector<double>&)> sink = [&](const std::vector<double>& a) { for (uint32_t i = 0; i < M; ++i) out[(size_t)sid * M + i] = a[i]; ++sid; return true; };
  CollapsedGibbsSampler sampler; const bool ok = sampler.sample(exp, so, sink, S);
  return ok && sid == S ? 0 : 1;
}
