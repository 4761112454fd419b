// oracle/ref_models_shim.cpp — TEST INFRASTRUCTURE.  C entry points around two bias models of the reference, compiled from where their sources
// lie under /root/reference (never copied) into oracle/_ref/libmodels_ref.so by oracle/Makefile:
//   src/model/SBModel.cpp + include/salmon/internal/model/SBModel.hpp    the read-start context model of --seqBias (addSequence, normalize, evaluateLog)
//   include/salmon/internal/model/GCFragModel.hpp                        the fragment-GC model of --gcBias (inc, normalize, ratio), header-only, Eigen vendored
// Two headers the reference takes from outside this tree are stood in for by oracle/_stub (Kmer.hpp: pufferfish's k-mer word; UtilityFunctions.hpp:
// one constexpr).  They pin the checker's restatement — sb_cell / sb_normalize / sb_eval and the GC normalize + ratio of oracle.cpp — in
// tests/test_models_pin.py.
#include "salmon/internal/model/SBModel.hpp"
#include "salmon/internal/model/GCFragModel.hpp"
#include <cmath>
#include <cstdint>
#include <cstring>
extern "C" {
// n contexts of 9 characters, each added with its weight (reverse-complemented first where rc[i]); normalize(); the 64 x 9 log-probability
// matrix comes back position-major [9][64] (Eigen's column-major storage of _probs), the marginals [9][4]; then evaluateLog of nq query contexts
void ref_sb_train_eval(const char* ctx, const double* w, const uint8_t* rc, int n, double* logp_9x64, double* marg_9x4, const char* q, int nq, double* qout) {
  SBModel m;
  for (int i = 0; i < n; ++i) m.addSequence(ctx + 9 * i, rc && rc[i], w[i]);
  m.normalize();
  memcpy(logp_9x64, m.counts().data(), sizeof(double) * 64 * 9);
  memcpy(marg_9x4, m.marginals().data(), sizeof(double) * 4 * 9);
  for (int i = 0; i < nq; ++i) qout[i] = m.evaluateLog(q + 9 * i);
}
// the observed model as the mapping loop fills it (LOG space, GCFragModel::inc with a log weight — SalmonQuantify.cpp:948,969), the expected model
// as updateEffectiveLengths fills it (LINEAR, normalize(); SalmonUtils.cpp:1269,1695,1714), then gcCounts.ratio(transcriptGCDist, 1000) (:1720).
// obs_mass / exp_mass: [3 context classes][25 GC bins], linear masses (0 = nothing added); out: the clamped ratios
void ref_gc_ratio(const double* obs_mass, const double* exp_mass, double* out) {
  GCFragModel obs(3, 25, distribution_utils::DistributionSpace::LOG), ex(3, 25, distribution_utils::DistributionSpace::LINEAR);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 25; ++c) {
    GCDesc d{4 * c + 1, 34 * r + 1};            // fragBin(25) = int(fragFrac / 4), contextBin(3) = int(contextFrac / 33.3)
    if (obs_mass[r * 25 + c] > 0.0) obs.inc(d, std::log(obs_mass[r * 25 + c]));
    if (exp_mass[r * 25 + c] > 0.0) ex.inc(d, exp_mass[r * 25 + c]);
  }
  ex.normalize();
  GCFragModel rat = obs.ratio(ex, 1000.0);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 25; ++c) { GCDesc d{4 * c + 1, 34 * r + 1}; out[r * 25 + c] = rat.get(d); }
}
int ref_gc_bins(int frag_frac, int ctx_frac, int* ctx_bin) { GCDesc d{frag_frac, ctx_frac}; *ctx_bin = d.contextBin(3); return d.fragBin(25); }
}
