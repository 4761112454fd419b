// oracle/_stub/vbem — TEST INFRASTRUCTURE.  Stand-ins on the include path of the VBEM pin only (oracle/Makefile, ref_vbem_shim.cpp): they let
// /root/reference/src/inference/CollapsedEMOptimizer.cpp compile where it lies, without TBB / Boost / spdlog / pufferfish.
// SalmonUtils.hpp: incLoop (include/salmon/internal/util/SalmonUtils.hpp:131-158) and the declaration of updateEffectiveLengths (the bias hook; the
// pin runs without bias correction, the definition in the shim aborts).
#pragma once
#include <atomic>
#include <cstdlib>
#include <vector>
#include "Eigen/Dense"
#include "oneapi/tbb/task_arena.h"
#include "salmon/internal/util/SalmonMath.hpp"
#include "salmon/internal/config/SalmonOpts.hpp"
namespace salmon { namespace utils {
inline void incLoop(double& val, double inc) { val += inc; }
inline void incLoop(std::atomic<double>& val, double inc) { double seen = val.load(); while (!val.compare_exchange_strong(seen, seen + inc)) {} }
template <class AbundanceVecT, class ReadExpT>
Eigen::VectorXd updateEffectiveLengths(oneapi::tbb::task_arena&, SalmonOpts&, ReadExpT&, Eigen::VectorXd& effLens, AbundanceVecT&, std::vector<bool>&, bool) { std::abort(); return effLens; }
} }
