// oracle/_stub/fld/.../SalmonUtils.hpp — TEST INFRASTRUCTURE.  The reference's header of this name pulls in Boost, spdlog, TBB and pufferfish;
// FragmentLengthDistribution.cpp needs one thing from it: the compare-exchange loop that adds a log-space increment to an atomic double
// (include/salmon/internal/util/SalmonUtils.hpp:147-153 in the reference).  Only on the include path of the FLD pin (oracle/Makefile).
#pragma once
#include <atomic>
#include <sstream>
#include "salmon/internal/util/SalmonMath.hpp"
namespace salmon { namespace utils {
inline void incLoopLog(std::atomic<double>& val, double inc) {
  double seen = val.load();
  while (!val.compare_exchange_strong(seen, salmon::math::logAdd(seen, inc))) {}
}
} }
