// oracle/_stub/vbem — TEST INFRASTRUCTURE.  Stand-ins on the include path of the VBEM pin only (oracle/Makefile, ref_vbem_shim.cpp): they let
// /root/reference/src/inference/CollapsedEMOptimizer.cpp compile where it lies, without TBB / Boost / spdlog / pufferfish.
// parallel_for over a blocked_range: ONE call with the whole range (a serial schedule is one of TBB's legal schedules; the reference's
// sums into alphaOut then run class after class, which is the order the pin compares with to rounding).
#pragma once
namespace oneapi { namespace tbb { template <class R, class F> void parallel_for(const R& r, const F& f) { f(r); } } }
