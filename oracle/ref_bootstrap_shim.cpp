// oracle/ref_bootstrap_shim.cpp — TEST INFRASTRUCTURE.  A C entry point around the reference's bootstrap, compiled from where the source lies under /root/reference
// (never copied) into oracle/_ref/libbootstrap_ref.so by oracle/Makefile:
//   src/inference/CollapsedEMOptimizer.cpp   gatherBootstraps (:554-690) + doBootstrap (:398-552): the multinomial resample of the class counts, the uniform start over
//                                            the active transcripts, the serial EM / VBEM of at least 50 iterations, the truncation — row a16 — run as in a
//                                            `salmon quant --numBootstraps B`: after optimize(), which leaves the combined weights the replicates use
// The stand-ins are the optimiser's pin's (oracle/_stub/vbem); std::random_device, which seeds each worker's mt19937, is redirected to a counter seeded by the caller
// so that the pin is a deterministic test.  The checker resamples from its own counter-based streams (SPEC §a16): the comparison is one of DISTRIBUTIONS over the
// replicates — tests/test_bootstrap_pin.py.
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
#include "src/inference/CollapsedEMOptimizer.cpp"
#include <cmath>
#include <functional>
#include <vector>
extern "C" int ref_bootstrap(uint64_t E, const uint64_t* off, const uint32_t* tid, const double* w, const uint64_t* count, uint32_t M, const double* eff_len, uint64_t num_mapped,
                             int use_vbem, int per_transcript_prior, double vb_prior, double tol, uint32_t max_iter, uint32_t B, uint64_t seed, double* out) {
  std::sq_fixed_random_device::state() = seed;
  using ExpT = ReadExperiment<EquivalenceClassBuilder<TGValue>>;
  ExpT exp; exp.txps.resize(M); exp.mapped = num_mapped;
  for (uint32_t i = 0; i < M; ++i) { Transcript& t = exp.txps[i]; t.RefLength = (uint32_t)eff_len[i]; t.EffectiveLength = eff_len[i]; t.cachedLogEffLen = std::log(eff_len[i]); }
  auto& vec = exp.builder.vec; vec.reserve(E);
  for (uint64_t c = 0; c < E; ++c) { TranscriptGroup g; g.txps.assign(tid + off[c], tid + off[c + 1]); TGValue v; v.weights.assign(w + off[c], w + off[c + 1]); v.count = count[c]; vec.emplace_back(std::move(g), std::move(v)); }
  SalmonOpts so; so.useVBOpt = use_vbem != 0; so.perTranscriptPrior = per_transcript_prior != 0; so.vbPrior = vb_prior; so.initUniform = true; so.numThreads = 1; so.numBootstraps = B; so.useQuasi = true; so.allowOrphans = true;
  CollapsedEMOptimizer opt; if (!opt.optimize(exp, so, tol, max_iter)) return 2;      // leaves the combined weights (and the active flags) the replicates work with
  uint32_t b = 0;
  std::function<bool(const std::vector<double>&)> sink = [&](const std::vector<double>& a) { if (b < B) for (uint32_t i = 0; i < M; ++i) out[(size_t)b * M + i] = a[i]; ++b; return true; };
  const bool ok = opt.gatherBootstraps(exp, so, sink, tol, max_iter);
  return ok && b == B ? 0 : 1;
}
