// oracle/_stub/vbem — TEST INFRASTRUCTURE.  Stand-ins on the include path of the VBEM pin only (oracle/Makefile, ref_vbem_shim.cpp): they let
// /root/reference/src/inference/CollapsedEMOptimizer.cpp compile where it lies, without TBB / Boost / spdlog / pufferfish.
// boost::math::digamma is not in the tree; the pin is about the UPDATE RULE (what is done with digamma's values), so the function is the
// checker's own (include/sq_math.h, held to scipy in tests/test_math.py).
#pragma once
#include "sq_math.h"
namespace boost { namespace math { inline double digamma(double x) { return sq_digamma(x); } } }
