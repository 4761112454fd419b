// oracle/_stub/mb/.../ReadExperiment.hpp — TEST INFRASTRUCTURE.  The reference's ReadExperiment owns the index (pufferfish) and the input files; processMiniBatch and
// normalizeAlphas reach through it to the equivalence-class builder, the transcripts, the cluster forest, the fragment-length distribution and the conditional means.
// Those members here are the REFERENCE'S classes (EquivalenceClassBuilder.hpp, Transcript.hpp, ClusterForest.hpp, FragmentLengthDistribution.hpp compiled as they lie);
// updateTranscriptLengthsAtomic is the reference's function body, cut out of include/salmon/internal/quant/ReadExperiment.inl:62-94 by oracle/Makefile.
#pragma once
#include <atomic>
#include <memory>
#include <vector>
#include "SpinLock.hpp"
#include "salmon/internal/quant/ClusterForest.hpp"
#include "salmon/internal/util/DistributionUtils.hpp"
#include "salmon/internal/quant/EquivalenceClassBuilder.hpp"
#include "salmon/internal/model/FragmentLengthDistribution.hpp"
#include "salmon/internal/model/Transcript.hpp"
template <typename EQBuilderT> class ReadExperiment {
public:
  ReadExperiment(std::shared_ptr<spdlog::logger> log, uint32_t fldMax, uint32_t fldMean, uint32_t fldSD) : eqBuilder_(log, 1) {
    fragLengthDist_.reset(new FragmentLengthDistribution(1.0, fldMax, fldMean, fldSD, 4, 0.5, 1));     // ReadExperiment.inl:19-24: alpha 1, kernel n 4, p 0.5, bin size 1
  }
  EQBuilderT& equivalenceClassBuilder() { return eqBuilder_; }
  std::vector<Transcript>& transcripts() { return transcripts_; }
  ClusterForest& clusterForest() { return *clusters_.get(); }
  FragmentLengthDistribution* fragmentLengthDistribution() { return fragLengthDist_.get(); }
  std::vector<double>& condMeans() { return conditionalMeans_; }
  uint64_t numMappedFragments() const { return numMapped_; }
  void updateTranscriptLengthsAtomic(std::atomic<bool>& done);
  std::vector<Transcript> transcripts_; std::unique_ptr<ClusterForest> clusters_; std::unique_ptr<FragmentLengthDistribution> fragLengthDist_; std::vector<double> conditionalMeans_;
  EQBuilderT eqBuilder_; SpinLock sl_; uint64_t numMapped_ = 0;
};
