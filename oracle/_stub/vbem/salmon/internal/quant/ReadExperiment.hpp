// oracle/_stub/vbem — TEST INFRASTRUCTURE.  Stand-ins on the include path of the VBEM pin only (oracle/Makefile, ref_vbem_shim.cpp): they let
// /root/reference/src/inference/CollapsedEMOptimizer.cpp compile where it lies, without TBB / Boost / spdlog / pufferfish.
// ReadExperiment / EquivalenceClassBuilder / TGValue: the accessors the optimiser calls, over plain vectors the shim fills
// (include/salmon/internal/quant/ReadExperiment.hpp, EquivalenceClassBuilder.hpp:78-132).
#pragma once
#include <cstdint>
#include <utility>
#include <vector>
#include <unordered_set>
#include <thread>
#include <random>
#include <iostream>
#include "boost/range/irange.hpp"
#include "salmon/internal/model/Transcript.hpp"
#include "salmon/internal/model/TranscriptGroup.hpp"
#include "salmon/internal/config/SalmonOpts.hpp"
#include "salmon/internal/util/SalmonUtils.hpp"
struct TGValue { mutable std::vector<double> weights; mutable std::vector<double> combinedWeights; uint64_t count = 0; };
struct SCTGValue : TGValue {};
template <class V> class EquivalenceClassBuilder { public: std::vector<std::pair<const TranscriptGroup, V>> vec; std::vector<std::pair<const TranscriptGroup, V>>& eqVec() { return vec; }
  size_t getNumTranscriptsForClass(size_t c) const { return vec[c].first.txps.size(); } };
struct StubFragStartDists {};
template <class B> class ReadExperiment { public: std::vector<Transcript> txps; B builder; StubFragStartDists fsd; uint64_t mapped = 0;
  std::vector<Transcript>& transcripts() { return txps; } B& equivalenceClassBuilder() { return builder; } StubFragStartDists& fragmentStartPositionDistributions() { return fsd; }
  uint64_t numMappedFragments() const { return mapped; } uint64_t upperBoundHits() const { return mapped; } };
