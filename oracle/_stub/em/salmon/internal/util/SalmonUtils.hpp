// oracle/_stub/em/.../SalmonUtils.hpp — TEST INFRASTRUCTURE.  The reference's header of this name pulls in Boost, spdlog, TBB and pufferfish;
// src/inference/EMUtils.cpp needs one thing from it: "add to a double", plain and as a compare-exchange loop on an atomic
// (include/salmon/internal/util/SalmonUtils.hpp:131-158 in the reference).  Only on the include path of the EM pin (oracle/Makefile).
#pragma once
#include <atomic>
#include <cassert>
#include <cmath>
#include <limits>
#include "salmon/internal/util/SalmonMath.hpp"
namespace salmon { namespace utils {
inline void incLoop(double& val, double inc) { val += inc; }
inline void incLoop(std::atomic<double>& val, double inc) { double seen = val.load(); while (!val.compare_exchange_strong(seen, seen + inc)) {} }
} }
