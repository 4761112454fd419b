// oracle/_stub/aln — TEST INFRASTRUCTURE.  Stand-ins on the include path of the alignment-model pin only (oracle/Makefile, ref_alnmodel_shim.cpp): they let
// /root/reference/src/alignment/AlignmentModel.cpp and AlignmentCommon.cpp compile where they lie, without htslib / spdlog / TBB / Boost / pufferfish.
// Transcript: length, name, id and baseAt (include/salmon/internal/model/Transcript.hpp:185-198: the sequence as SAM codes, two per byte, first base high)
#pragma once
#include <cstdint>
#include <limits>
#include <string>
#include <vector>
#include "salmon/internal/util/SalmonStringUtils.hpp"
class Transcript { public: uint32_t RefLength = 0; std::string RefName; uint32_t id = 0; std::vector<uint8_t> SAMSequence_;
  inline uint8_t baseAt(size_t idx, salmon::stringtools::strand dir = salmon::stringtools::strand::forward) {
    using salmon::stringtools::strand; using salmon::stringtools::encodedRevComp;
    const size_t byte = idx >> 1; const size_t nibble = (!(idx & 0x1)) << 2; const uint8_t base = (SAMSequence_[byte] >> nibble) & 0x0F;
    switch (dir) { case strand::forward: return base; case strand::reverse: return encodedRevComp[base]; }
    return std::numeric_limits<uint8_t>::max(); } };
