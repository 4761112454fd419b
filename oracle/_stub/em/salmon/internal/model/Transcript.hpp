// oracle/_stub/em/.../Transcript.hpp — TEST INFRASTRUCTURE.  The reference's Transcript.hpp pulls in pufferfish, Boost and the sequence
// machinery; include/salmon/internal/inference/EMUtils.hpp includes it and uses nothing of it.  Only on the include path of the
// EM pin (oracle/Makefile).
#pragma once
#include <cstdint>
#include "salmon/internal/util/SalmonMath.hpp"
class Transcript {
public:
  uint32_t RefLength = 0; double EffectiveLength = 0.0; double cachedLogEffLen = 0.0;
  void setCachedLogEffectiveLength(double v) { cachedLogEffLen = v; }
};
