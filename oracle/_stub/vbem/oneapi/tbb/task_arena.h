// oracle/_stub/vbem — TEST INFRASTRUCTURE.  Stand-ins on the include path of the VBEM pin only (oracle/Makefile, ref_vbem_shim.cpp): they let
// /root/reference/src/inference/CollapsedEMOptimizer.cpp compile where it lies, without TBB / Boost / spdlog / pufferfish.
// oneapi::tbb::task_arena: the arena runs the callable in place.
#pragma once
namespace oneapi { namespace tbb { class task_arena { public: explicit task_arena(int = 1) {} template <class F> void execute(F&& f) { f(); } }; } }
