// oracle/_stub/fld/.../Transcript.hpp — TEST INFRASTRUCTURE.  The reference's Transcript.hpp pulls in pufferfish, Boost and the sequence
// machinery; src/util/DistributionUtils.cpp touches three members of it (computeSmoothedEffectiveLengths, :31-55).  Only on the include path of the
// FLD pin (oracle/Makefile).
#pragma once
#include <cstdint>
#include "salmon/internal/util/SalmonMath.hpp"
class Transcript {
public:
  uint32_t RefLength = 0; double EffectiveLength = 0.0; double cachedLogEffLen = 0.0;
  void setCachedLogEffectiveLength(double v) { cachedLogEffLen = v; }
};
