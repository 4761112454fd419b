// oracle/_stub/vbem — TEST INFRASTRUCTURE.  Stand-ins on the include path of the VBEM pin only (oracle/Makefile, ref_vbem_shim.cpp): they let
// /root/reference/src/inference/CollapsedEMOptimizer.cpp compile where it lies, without TBB / Boost / spdlog / pufferfish.
#pragma once
#include <cstddef>
namespace oneapi { namespace tbb { template <class T> class blocked_range { T b_, e_; public: blocked_range(T b, T e, size_t = 1) : b_(b), e_(e) {} T begin() const { return b_; } T end() const { return e_; } }; } }
