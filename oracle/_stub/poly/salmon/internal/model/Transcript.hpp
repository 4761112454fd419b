// oracle/_stub/poly — TEST INFRASTRUCTURE.  Stand-ins on the include path of the polytope pin only (oracle/Makefile, ref_polytope_shim.cpp): they let
// /root/reference/include/salmon/internal/quant/{TranscriptCluster,ClusterForest}.hpp compile where they lie, without Boost.
// Transcript: what ClusterForest, projectToPolytope and normalizeAlphas read and set (include/salmon/internal/model/Transcript.hpp).
#pragma once
#include <cstdint>
#include "salmon/internal/util/SalmonMath.hpp"
class Transcript { public: double logMass_ = salmon::math::LOG_0; uint64_t uniq_ = 0, total_ = 0; uint64_t uniqueCounts = 0, totalCounts = 0; double projectedCounts = 0.0;
  double mass(bool = true) const { return logMass_; } uint64_t uniqueCount() const { return uniq_; } uint64_t totalCount() const { return total_; } void setMass(double) {} };
