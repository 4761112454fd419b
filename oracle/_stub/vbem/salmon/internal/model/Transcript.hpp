// oracle/_stub/vbem — TEST INFRASTRUCTURE.  Stand-ins on the include path of the VBEM pin only (oracle/Makefile, ref_vbem_shim.cpp): they let
// /root/reference/src/inference/CollapsedEMOptimizer.cpp compile where it lies, without TBB / Boost / spdlog / pufferfish.
// Transcript: what the optimiser reads and sets (include/salmon/internal/model/Transcript.hpp).
#pragma once
#include <cstdint>
#include "salmon/internal/util/SalmonMath.hpp"
class Transcript {
public:
  uint32_t RefLength = 0; double EffectiveLength = 0.0; double projectedCounts = 0.0; double cachedLogEffLen = 0.0; uint64_t uniq = 0; bool active = false; double sharedCount_ = 0.0, mass_ = 0.0;
  double getCachedLogEffectiveLength() const { return cachedLogEffLen; } uint64_t uniqueCount() const { return uniq; }
  void setSharedCount(double v) { sharedCount_ = v; } void setMass(double v) { mass_ = v; } void setActive() { active = true; } bool getActive() const { return active; }
};
