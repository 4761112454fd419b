// oracle/ref_minibatch_shim.cpp — TEST INFRASTRUCTURE.  C entry points around the reference's own online stage (row a10 and what it feeds), compiled from where the
// sources lie under /root/reference (never copied into the repository) into oracle/_ref/libminibatch_ref.so by oracle/Makefile:
//   src/quant/SalmonQuantify.cpp:279-1023            MiniBatchScratch / MiniBatchHotConfig / MiniBatchHotState and processMiniBatch<AlnT> itself.  The file as a whole
//                                                     needs pufferfish's mapper, FQFeeder and Boost.ProgramOptions; the Makefile cuts the lines of these definitions
//                                                     out of it into oracle/_ref/minibatch_slice.inc (git-ignored build output) and this file #includes the cut.
//   src/util/SalmonUtils.cpp:138-148,195-298          isCompatible, compatibleHit (both overloads)        — cut the same way (utils_slice.inc)
//   src/util/SalmonUtils.cpp:461-529                  normalizeAlphas                                       — "
//   include/salmon/internal/quant/ReadExperiment.inl:62-94   updateTranscriptLengthsAtomic (burn-in)        — " (readexp_slice.inc)
//   include/salmon/internal/quant/EquivalenceClassBuilder.hpp  addGroup / finish, TGValue, over the vendored libcuckoo map (include/salmon/vendor/cuckoohash_map.hh)
//   src/model/TranscriptGroup.cpp, src/model/LibraryFormat.cpp, src/model/FragmentLengthDistribution.cpp, src/util/DistributionUtils.cpp (LogCMFCache),
//   src/model/SimplePosBias.cpp; headers Transcript.hpp, ClusterForest.hpp, TranscriptCluster.hpp, ForgettingMassCalculator.hpp, ReadLibrary.hpp,
//   LibraryTypeDetector.hpp, GCFragModel.hpp, SalmonOpts.hpp, spdlog and core/range.hpp (vendored in the reference tree) — all as they lie.
// Stood in for (oracle/_stub/mb, each file says what it replaces): pufferfish's QuasiAlignment / MateStatus / LibraryFormat.hpp / compact_vector / rank9b, Boost
// (filesystem::path, disjoint_sets, dynamic_bitset, normal / binomial), RapMap's SpinLock, sparsepp, TBB's concurrent_vector, and the headers SalmonUtils.hpp,
// AlignmentGroup.hpp, ReadExperiment.hpp (declarations only / plain containers of the reference's classes).
// One thread, one mini-batch in flight: the function is then deterministic given the engine it draws from.  The engine is std::default_random_engine seeded by the
// caller; the uniform draws a mini-batch WILL make are reported (a copy of the engine is run ahead with the function's own distribution object, :458-459), so the
// checker can be run on the same draws (tests/test_minibatch_pin.py).
#include <algorithm>
#include <atomic>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>
#include <mutex>
#include <numeric>
#include <random>
#include <sstream>
#include <thread>
#include <vector>
#include <spdlog/spdlog.h>
#include <spdlog/sinks/null_sink.h>
#include "core/range.hpp"
#include "Util.hpp"
#include "salmon/internal/alignment/AlignmentGroup.hpp"
#include "salmon/internal/config/SalmonOpts.hpp"
#include "salmon/internal/config/SalmonDefaults.hpp"
#include "salmon/internal/model/LibraryFormat.hpp"
#include "salmon/internal/quant/ReadLibrary.hpp"
#include "salmon/internal/util/SalmonMath.hpp"
#include "salmon/internal/util/SalmonUtils.hpp"
#include "salmon/internal/model/Transcript.hpp"
#include "salmon/internal/quant/ClusterForest.hpp"
#include "salmon/internal/quant/EquivalenceClassBuilder.hpp"
#include "salmon/internal/quant/ForgettingMassCalculator.hpp"
#include "salmon/internal/model/FragmentLengthDistribution.hpp"
#include "salmon/internal/model/FragmentStartPositionDistribution.hpp"
#include "salmon/internal/model/GCFragModel.hpp"
#include "salmon/internal/model/SimplePosBias.hpp"
#include "salmon/internal/util/DistributionUtils.hpp"
#include "salmon/internal/quant/ReadExperiment.hpp"

using MateStatus = pufferfish::util::MateStatus;
using QuasiAlignment = pufferfish::util::QuasiAlignment;
#include "_ref/readexp_slice.inc"        // ReadExperiment<EQBuilderT>::updateTranscriptLengthsAtomic
namespace salmon { namespace utils {
#include "_ref/utils_slice.inc"          // isCompatible, compatibleHit x 2, normalizeAlphas
} }
#include "_ref/minibatch_slice.inc"      // AlnGroupVec, MiniBatchScratch, MiniBatchHotConfig, MiniBatchHotState, ReadExperimentT, processMiniBatch

namespace {
struct RefMB {
  std::shared_ptr<spdlog::logger> log; SalmonOpts sopt; std::unique_ptr<ReadExperimentT> exp; std::unique_ptr<ReadLibrary> lib; LibraryFormat fmt{ReadType::PAIRED_END, ReadOrientation::TOWARD, ReadStrandedness::U};
  std::unique_ptr<ForgettingMassCalculator> fm; uint64_t firstTimestep = 0; std::atomic<uint64_t> numAssigned{0}; std::atomic<bool> burnedIn{false}; double maxZeroFrac = 0.0;
  std::default_random_engine eng; std::unique_ptr<distribution_utils::LogCMFCache> cmfCache; MiniBatchScratch scratch; MiniBatchHotConfig hot;
  std::vector<FragmentStartPositionDistribution> fsd; std::unique_ptr<GCFragModel> gc; double massFwd = salmon::math::LOG_0, massRC = salmon::math::LOG_0; std::vector<SimplePosBias> posFW, posRC;
  bool finished = false;
};
}
// the input record: sq_aln of include/salmon_hip.h, field for field (40 bytes)
struct mb_aln { uint32_t tid; int32_t pos, mate_pos, score, mate_score; uint32_t frag_len; uint16_t read_len, mate_len; uint8_t fwd, mate_fwd, mate_status, format_id; double est_aln_prob; };
struct mb_opts { uint32_t lib_type, lib_orientation, lib_strand; uint32_t range_factorization_bins; uint64_t num_burnin_frags, num_pre_burnin_frags; double incompat_prior, forgetting_factor;
  uint32_t fld_max, fld_mean, fld_sd; uint8_t ignore_incompat, no_eff_length_correction, no_length_correction, no_frag_length_dist, no_single_frag_prob, rank_eq_classes, _pad[2]; uint64_t engine_seed; };

extern "C" {
void* ref_mb_create(uint32_t M, const uint32_t* ref_len, const uint32_t* complete_len, const mb_opts* o) {
  RefMB* R = new RefMB();
  R->log = std::make_shared<spdlog::logger>("mb", std::make_shared<spdlog::sinks::null_sink_mt>());
  SalmonOpts& s = R->sopt; s.jointLog = R->log;
  s.fragLenDistMax = o->fld_max; s.fragLenDistPriorMean = o->fld_mean; s.fragLenDistPriorSD = o->fld_sd; s.noEffectiveLengthCorrection = o->no_eff_length_correction; s.ignoreIncompat = o->ignore_incompat;
  s.incompatPrior = o->incompat_prior; s.forgettingFactor = o->forgetting_factor; s.numBurninFrags = o->num_burnin_frags; s.numPreBurninFrags = o->num_pre_burnin_frags;
  s.rangeFactorizationBins = o->range_factorization_bins; s.noLengthCorrection = o->no_length_correction; s.noFragLengthDist = o->no_frag_length_dist; s.noSingleFragProb = o->no_single_frag_prob;
  s.rankEqClasses = o->rank_eq_classes; s.posBiasCorrect = false; s.gcBiasCorrect = false; s.noFragLenFactor = false;
  R->exp.reset(new ReadExperimentT(R->log, o->fld_max, o->fld_mean, o->fld_sd));
  auto& T = R->exp->transcripts_; T.reserve(M);
  for (uint32_t i = 0; i < M; ++i) { T.emplace_back(i, "t", ref_len[i], 0.005); T.back().setCompleteLength(complete_len[i]); }    // ReadExperiment.inl:114-134 (alpha = 0.005)
  R->exp->clusters_.reset(new ClusterForest(T.size(), T));                                                                              // :59
  R->fmt = LibraryFormat((ReadType)o->lib_type, (ReadOrientation)o->lib_orientation, (ReadStrandedness)o->lib_strand);
  R->lib.reset(new ReadLibrary(R->fmt));
  R->fm.reset(new ForgettingMassCalculator(o->forgetting_factor)); R->fm->prefill(1000000000 / 5000);                                   // SalmonQuantify.cpp:2543-2545
  R->firstTimestep = R->fm->getCurrentTimestep();                                                                                       // :1108
  R->eng.seed((std::default_random_engine::result_type)o->engine_seed);
  R->cmfCache.reset(new distribution_utils::LogCMFCache(R->exp->fragmentLengthDistribution(), o->lib_type == 0));                       // :1060 (singleEndLib)
  R->hot = MiniBatchHotConfig{s.numBurninFrags, s.posBiasCorrect, s.gcBiasCorrect, !s.noFragLengthDist, s.noFragLenFactor, s.rankEqClasses, s.rangeFactorizationBins,
                              s.noLengthCorrection, !s.noSingleFragProb, (uint32_t)s.numPreBurninFrags, s.incompatPrior};                  // :1063-1074
  R->gc.reset(new GCFragModel());
  R->exp->equivalenceClassBuilder().start();
  return R;
}
void ref_mb_free(void* h) { delete (RefMB*)h; }
// one call of processMiniBatch<QuasiAlignment> on the fragments [0, n) (CSR offsets into alns); draws_out (capacity = number of alignments) receives the uniform
// numbers the call's `uni(randEng)` produces, in order; returns how many alignments got a log probability other than LOG_0 (= draws consumed)
uint64_t ref_mb_process(void* h, uint32_t n, const uint64_t* off, const mb_aln* alns, double* draws_out, double* log_prob_out) {
  RefMB* R = (RefMB*)h;
  AlnGroupVec<QuasiAlignment> groups(n);
  for (uint32_t r = 0; r < n; ++r) for (uint64_t i = off[r]; i < off[r + 1]; ++i) {
    const mb_aln& a = alns[i]; QuasiAlignment q; q.tid = a.tid; q.pos = a.pos; q.matePos = a.mate_pos; q.fwd = a.fwd; q.mateIsFwd = a.mate_fwd; q.readLen = a.read_len; q.mateLen = a.mate_len;
    q.fragLen = a.frag_len; q.mateStatus = (MateStatus)a.mate_status; q.isPaired = a.mate_status == 3; q.formatID_ = a.format_id; q.estAlnProb_ = a.est_aln_prob; q.score_ = a.score; q.mateScore_ = a.mate_score;
    groups[r].alns.push_back(q);
  }
  { std::default_random_engine ahead = R->eng; std::uniform_real_distribution<> uni(0.0, 1.0 + std::numeric_limits<double>::min()); for (uint64_t i = 0; i < off[n]; ++i) draws_out[i] = uni(ahead); }
  AlnGroupVecRange<QuasiAlignment> range(groups.begin(), groups.end());
  MiniBatchHotState hs{&R->fsd, R->gc.get(), &R->massFwd, &R->massRC, &R->posFW, &R->posRC};
  processMiniBatch<QuasiAlignment>(*R->exp, *R->fm, R->firstTimestep, *R->lib, R->sopt, R->hot, hs, range, R->exp->transcripts(), R->exp->clusterForest(), *R->exp->fragmentLengthDistribution(),
                                   R->numAssigned, R->eng, true, R->burnedIn, R->maxZeroFrac, *R->cmfCache, R->scratch);
  uint64_t used = 0;
  for (uint32_t r = 0; r < n; ++r) for (size_t k = 0; k < groups[r].alns.size(); ++k) { const double lp = groups[r].alns[k].logProb; if (log_prob_out) log_prob_out[off[r] + k] = lp; if (std::abs(lp) != salmon::math::LOG_0) ++used; }
  return used;
}
void ref_mb_state(void* h, double* log_mass, uint64_t* uniq, uint64_t* total, double* log_eff_len, double* fld_log_pmf /* [fld_max + 1] */, uint64_t* num_assigned, int* burned_in, uint64_t* num_compat) {
  RefMB* R = (RefMB*)h; auto& T = R->exp->transcripts();
  for (size_t t = 0; t < T.size(); ++t) { log_mass[t] = T[t].mass(false); uniq[t] = T[t].uniqueCount(); total[t] = T[t].totalCount(); log_eff_len[t] = T[t].getCachedLogEffectiveLength(); }
  FragmentLengthDistribution* f = R->exp->fragmentLengthDistribution();
  for (size_t l = 0; l <= f->maxVal(); ++l) fld_log_pmf[l] = f->pmf(l);
  *num_assigned = R->numAssigned.load(); *burned_in = R->burnedIn.load() ? 1 : 0; *num_compat = R->lib->numCompat();
}
// EquivalenceClassBuilder::finish() (:165-181): normalises every class's weights; then the classes are handed out one by one
uint64_t ref_mb_eq_finish(void* h, uint64_t* total_label_len) {
  RefMB* R = (RefMB*)h; if (!R->finished) { R->exp->equivalenceClassBuilder().finish(); R->finished = true; }
  auto& v = R->exp->equivalenceClassBuilder().eqVec(); uint64_t L = 0; for (auto& kv : v) L += kv.first.txps.size(); *total_label_len = L; return v.size();
}
void ref_mb_eq_fetch(void* h, uint64_t* off /* [E + 1] offsets into label */, uint32_t* label /* tids then bins, as the key holds them */, uint64_t* count, double* weights /* [sum of n per class] in class order */, uint64_t* woff) {
  RefMB* R = (RefMB*)h; auto& v = R->exp->equivalenceClassBuilder().eqVec(); uint64_t p = 0, w = 0; size_t c = 0;
  for (auto& kv : v) { off[c] = p; woff[c] = w; for (uint32_t x : kv.first.txps) label[p++] = x; for (double x : kv.second.weights) weights[w++] = x; count[c] = kv.second.count; ++c; }
  off[c] = p; woff[c] = w;
}
// salmon::utils::normalizeAlphas (SalmonUtils.cpp:461-529) on the experiment as the mini-batches left it; projectedCounts per transcript
void ref_mb_normalize_alphas(void* h, uint64_t num_mapped, double* projected) {
  RefMB* R = (RefMB*)h; R->exp->numMapped_ = num_mapped;
  salmon::utils::normalizeAlphas(R->sopt, *R->exp);
  auto& T = R->exp->transcripts(); for (size_t t = 0; t < T.size(); ++t) projected[t] = T[t].projectedCounts;
}
}
