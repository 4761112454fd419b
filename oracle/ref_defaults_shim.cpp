// oracle/ref_defaults_shim.cpp — TEST INFRASTRUCTURE.  C entry points around two header-only files of the reference, compiled from where
// they lie under /root/reference (never copied) into oracle/_ref/libdefaults_ref.so by oracle/Makefile:
//   include/salmon/internal/config/SalmonDefaults.hpp   the default value of every option (needs <thread> only)
//   include/salmon/internal/util/SalmonMath.hpp          LOG_0 / LOG_EPSILON / ..., logAdd, logSub (needs an empty boost/config.hpp: oracle/_stub)
//   include/salmon/internal/quant/ForgettingMassCalculator.hpp   the learning-rate schedule of the online phase
// They pin what the product and the checker take as defaults (sq_quant_opts_default, sq_em_opts_default, the CLI) and the log-space
// helpers of include/sq_math.h to the reference's own values (tests/test_defaults_pin.py).
#include <string>   // the header uses std::string without including it
#include "salmon/internal/config/SalmonDefaults.hpp"
#include "salmon/internal/util/SalmonMath.hpp"
#include <mutex>
#include <vector>
#include <cmath>
#include <cstdint>
#include "salmon/internal/quant/ForgettingMassCalculator.hpp"   // header-only too (its spdlog include is vendored under the reference's include/)
#include <cstring>
namespace d = salmon::defaults;
extern "C" int ref_default(const char* name, double* out) {
  struct E { const char* n; double v; };
  static const E tab[] = {
    {"incompatPrior", d::incompatPrior}, {"consensusSlack", (double)d::consensusSlack}, {"minScoreFraction", d::minScoreFraction},
    {"pre_merge_chain_sub_thresh", d::pre_merge_chain_sub_thresh}, {"post_merge_chain_sub_thresh", d::post_merge_chain_sub_thresh},
    {"orphan_chain_sub_thresh", d::orphan_chain_sub_thresh}, {"scoreExp", d::scoreExp}, {"matchScore", d::matchScore},
    {"mismatchPenalty", d::mismatchPenalty}, {"gapOpenPenalty", d::gapOpenPenalty}, {"gapExtendPenalty", d::gapExtendPenalty},
    {"dpBandwidth", d::dpBandwidth}, {"mismatchSeedSkip", d::mismatchSeedSkip}, {"disableChainingHeuristic", d::disableChainingHeuristic},
    {"hardFilter", d::hardFilter}, {"allowDovetail", d::allowDovetail}, {"recoverOrphans", d::recoverOrphans}, {"discardOrphansQuasi", d::discardOrphansQuasi},
    {"minAssignedFrags", d::minAssignedFrags}, {"biasSpeedSamp", d::biasSpeedSamp}, {"maxFragLength", d::maxFragLength},
    {"fragLenPriorMean", d::fragLenPriorMean}, {"fragLenPriorSD", d::fragLenPriorSD}, {"ffactor", d::ffactor}, {"maxReadOccs", d::maxReadOccs},
    {"maxRecoverReadOccs", d::maxRecoverReadOccs}, {"maxOccsPerHit", d::maxOccsPerHit}, {"noLengthCorrection", d::noLengthCorrection},
    {"noEffectiveLengthCorrection", d::noEffectiveLengthCorrection}, {"noFragLengthDist", d::noFragLengthDist}, {"noSingleFragProb", d::noSingleFragProb},
    {"numBiasSamples", d::numBiasSamples}, {"numBurninFrags", d::numBurninFrags}, {"numPreBurninFrags", d::numPreBurninFrags}, {"useEM", d::useEM},
    {"useVBOpt", d::useVBOpt}, {"sigDigits", d::sigDigits}, {"rangeFactorizationBins", d::rangeFactorizationBins}, {"thinningFactor", d::thinningFactor},
    {"noGammaDraw", d::noGammaDraw}, {"perTranscriptPrior", d::perTranscriptPrior}, {"perNucleotidePrior", d::perNucleotidePrior}, {"vbPrior", d::vbPrior},
    {"decoyThreshold", d::decoyThreshold}, {"minAlnProb", d::minAlnProb}, {"numFragGCBins", (double)d::numFragGCBins},
    {"numConditionalGCBins", (double)d::numConditionalGCBins}, {"initUniform", d::initUniform}, {"alternativeInitMode", d::alternativeInitMode},
    {"numThreads", d::numThreads}, {"validateMappings", d::validateMappings},
  };
  for (const E& e : tab) if (!strcmp(e.n, name)) { *out = e.v; return 1; }
  return 0;
}
extern "C" const char* ref_default_aux_dir() { return d::auxDir; }
extern "C" double ref_log_add(double x, double y) { return salmon::math::logAdd(x, y); }
extern "C" double ref_log_sub(double x, double y) { return salmon::math::logSub(x, y); }
extern "C" double ref_math_const(int which) {   // 0 LOG_0, 1 LOG_1, 2 LOG_ONEHALF, 3 LOG_ORPHAN_PROB, 4 EPSILON, 5 LOG_EPSILON
  switch (which) { case 0: return salmon::math::LOG_0; case 1: return salmon::math::LOG_1; case 2: return salmon::math::LOG_ONEHALF;
    case 3: return salmon::math::LOG_ORPHAN_PROB; case 4: return salmon::math::EPSILON; default: return salmon::math::LOG_EPSILON; }
}
// the schedule as salmon quant consumes it (SalmonQuantify.cpp:2544-2545 prefill, :515 getLogMassAndTimestep): out[b] = log forgetting mass of mini-batch b
extern "C" void ref_forgetting_masses(double ff, uint32_t n, double* out) {
  ForgettingMassCalculator fm(ff); fm.prefill((uint64_t)n + 8);
  for (uint32_t b = 0; b < n; ++b) { double m; uint64_t t; fm.getLogMassAndTimestep(m, t); out[b] = m; }
}
