// oracle/ref_vbem_shim.cpp — TEST INFRASTRUCTURE.  C entry points around the reference's default optimiser, compiled from where the source lies under
// /root/reference (never copied) into oracle/_ref/libvbem_ref.so by oracle/Makefile:
//   src/inference/CollapsedEMOptimizer.cpp   VBEMUpdate_ (serial, :104-171 — the update of the bootstrap replicates), and the whole of
//                                            CollapsedEMOptimizer::optimize (:732-1035: initialisation from the projected counts, combined weights,
//                                            markDegenerateClasses, the VBEM / EM update over the class vector (:241-328 / :178-234), the convergence
//                                            rule, truncateCountVector) — the code every default `salmon quant` run ends with.
// The file is #included so that its file-local templates are reachable; TBB (run in place), Boost's irange / digamma (= the checker's sq_digamma: the
// pin is about the update rule), spdlog, ReadExperiment / Transcript / TranscriptGroup / SalmonOpts are stood in for by oracle/_stub/vbem, each stub
// saying what it replaces.  Pins the checker's em_step / em_optimize (oracle.cpp) and through them the HIP kernels that are bit-exact with the
// checker — tests/test_vbem_pin.py.
#include "salmon/internal/quant/ReadExperiment.hpp"
#include "salmon/internal/util/SalmonUtils.hpp"
#include "src/inference/CollapsedEMOptimizer.cpp"
#include <cmath>
#include <cstdint>
#include <vector>
extern "C" {
// CSR in (off[E + 1], tid[L], combined weights cw[L], count[E]), prior[M]; one serial VBEMUpdate_ from alpha_in
void ref_vbem_update(uint64_t E, const uint64_t* off, const uint32_t* tid, const double* cw, const uint64_t* count, uint32_t M, const double* prior,
                     const double* alpha_in, double* alpha_out, double* exp_theta) {
  std::vector<std::vector<uint32_t>> labels(E); std::vector<std::vector<double>> weights(E); std::vector<uint64_t> counts(count, count + E);
  for (uint64_t c = 0; c < E; ++c) { labels[c].assign(tid + off[c], tid + off[c + 1]); weights[c].assign(cw + off[c], cw + off[c + 1]); }
  std::vector<double> pr(prior, prior + M), in(alpha_in, alpha_in + M), out(M, 0.0), th(M, 0.0);
  VBEMUpdate_(labels, weights, counts, pr, in, out, th);
  for (uint32_t i = 0; i < M; ++i) { alpha_out[i] = out[i]; exp_theta[i] = th[i]; }
}
// CollapsedEMOptimizer::optimize on a class table (off, tid, auxiliary weights w, count) and per-transcript inputs (projected counts, unique counts,
// effective lengths); alphas (Transcript::sharedCount) out; returns the iteration count the optimiser logs at its end (< 0: optimize() said false)
int64_t ref_optimize(uint64_t E, const uint64_t* off, const uint32_t* tid, const double* w, const uint64_t* count, uint32_t M, const double* projected,
                     const uint64_t* unique, const double* eff_len, int use_vbem, int per_transcript_prior, int init_uniform, int eq_class_mode, int no_rich,
                     int alt_init, double vb_prior, double num_required_fragments, double tol, uint32_t max_iter, double* alpha_out, uint64_t* num_valid_out) {
  using ExpT = ReadExperiment<EquivalenceClassBuilder<TGValue>>;
  ExpT exp; exp.txps.resize(M);
  for (uint32_t i = 0; i < M; ++i) { Transcript& t = exp.txps[i]; t.RefLength = (uint32_t)eff_len[i]; t.projectedCounts = projected ? projected[i] : 0.0; t.uniq = unique ? unique[i] : 0; t.cachedLogEffLen = std::log(eff_len[i]); }
  auto& vec = exp.builder.vec; vec.reserve(E);
  for (uint64_t c = 0; c < E; ++c) {
    TranscriptGroup g; g.txps.assign(tid + off[c], tid + off[c + 1]); TGValue v; v.weights.assign(w + off[c], w + off[c + 1]); v.count = count[c];
    vec.emplace_back(std::move(g), std::move(v));
  }
  SalmonOpts so; so.useVBOpt = use_vbem != 0; so.perTranscriptPrior = per_transcript_prior != 0; so.initUniform = init_uniform != 0; so.eqClassMode = eq_class_mode != 0;
  so.noRichEqClasses = no_rich != 0; so.alternativeInitMode = alt_init != 0; so.vbPrior = vb_prior; so.numRequiredFragments = (uint32_t)num_required_fragments;
  CollapsedEMOptimizer opt; const bool ok = opt.optimize(exp, so, tol, max_iter);
  for (uint32_t i = 0; i < M; ++i) alpha_out[i] = exp.txps[i].sharedCount_;
  if (num_valid_out) { uint64_t nv = 0; for (auto& kv : vec) nv += kv.first.valid; *num_valid_out = nv; }
  return ok ? (int64_t)so.jointLog->last_iter : -1;
}
}
