// oracle/_stub/mb/.../SalmonUtils.hpp — TEST INFRASTRUCTURE.  The reference's header of this name pulls in htslib, Boost.ProgramOptions, TBB, cereal and
// pufferfish.  What the files of the mini-batch pin use of it, with the reference's own declarations (include/salmon/internal/util/SalmonUtils.hpp:46,
// :147-172, :268-280); the DEFINITIONS of isCompatible / compatibleHit / normalizeAlphas are the reference's, cut out of src/util/SalmonUtils.cpp by
// oracle/Makefile into oracle/_ref/ and compiled with the shim.  On the include path of the mini-batch pin only.
#pragma once
#include <atomic>
#include <cstdint>
#include <sstream>
#include <string>
#include <vector>
#include "salmon/internal/util/SalmonMath.hpp"
#include "salmon/internal/model/LibraryFormat.hpp"
#include "salmon/internal/config/SalmonOpts.hpp"
#include "Util.hpp"
namespace salmon { namespace utils {
using MateStatus = pufferfish::util::MateStatus;
inline void incLoopLog(std::atomic<double>& val, double inc) { double o = val.load(), n; do { n = salmon::math::logAdd(o, inc); } while (!val.compare_exchange_strong(o, n)); }
inline void incLoop(double& val, double inc) { val += inc; }
inline void incLoop(std::atomic<double>& val, double inc) { double o = val.load(), n; do { n = o + inc; } while (!val.compare_exchange_strong(o, n)); }
template <typename AlnLibT> void normalizeAlphas(const SalmonOpts& sopt, AlnLibT& alnLib);
bool isCompatible(const LibraryFormat observed, const LibraryFormat expected, int32_t start, bool isForward, MateStatus ms);
bool compatibleHit(const LibraryFormat expected, int32_t start, bool isForward, MateStatus ms);
bool compatibleHit(const LibraryFormat expected, const LibraryFormat observed);
} }
