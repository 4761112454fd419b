// oracle/_stub/aln — TEST INFRASTRUCTURE.  Stand-ins on the include path of the alignment-model pin only (oracle/Makefile, ref_alnmodel_shim.cpp): they let
// /root/reference/src/alignment/AlignmentModel.cpp and AlignmentCommon.cpp compile where they lie, without htslib / spdlog / TBB / Boost / pufferfish.
// SalmonUtils.hpp: what AtomicMatrix.hpp uses — "add to a double" as compare-exchange loops, plain and in log space (include/salmon/internal/util/SalmonUtils.hpp:131-158)
#pragma once
#include <atomic>
#include <cstdint>
#include <iostream>
#include <ostream>
#include <sstream>
#include "salmon/internal/util/SalmonMath.hpp"
namespace salmon { namespace utils {
inline void incLoop(std::atomic<double>& val, double inc) { double seen = val.load(); while (!val.compare_exchange_strong(seen, seen + inc)) {} }
inline void incLoopLog(std::atomic<double>& val, double inc) { double seen = val.load(); while (!val.compare_exchange_strong(seen, salmon::math::logAdd(seen, inc))) {} }
enum class OrphanStatus : uint8_t { LeftOrphan = 0, RightOrphan = 1, Paired = 2 };
inline std::ostream& operator<<(std::ostream& os, OrphanStatus s) { return os << (int)s; }
} }
