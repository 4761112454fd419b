// oracle/_stub/vbem — TEST INFRASTRUCTURE.  Stand-ins on the include path of the VBEM pin only (oracle/Makefile, ref_vbem_shim.cpp): they let
// /root/reference/src/inference/CollapsedEMOptimizer.cpp compile where it lies, without TBB / Boost / spdlog / pufferfish.
// TranscriptGroup: label + the validity flag markDegenerateClasses clears (include/salmon/internal/model/TranscriptGroup.hpp).
#pragma once
#include <cstdint>
#include <vector>
class TranscriptGroup { public: std::vector<uint32_t> txps; size_t hash = 0; double totalMass = 0.0; mutable bool valid = true; void setValid(bool v) const { valid = v; } };
