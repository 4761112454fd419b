// oracle/_stub/vbem — TEST INFRASTRUCTURE.  Stand-ins on the include path of the VBEM pin only (oracle/Makefile, ref_vbem_shim.cpp): they let
// /root/reference/src/inference/CollapsedEMOptimizer.cpp compile where it lies, without TBB / Boost / spdlog / pufferfish.
// SalmonOpts: the members CollapsedEMOptimizer.cpp reads, with the reference's names (include/salmon/internal/config/SalmonOpts.hpp).
#pragma once
#include <cstdint>
#include <memory>
#include "spdlog/fmt/fmt.h"
#include <boost/filesystem.hpp>
struct SalmonOpts {
  uint32_t numThreads = 1; bool biasCorrect = false, gcBiasCorrect = false, posBiasCorrect = false, meta = false, alternativeInitMode = false, noRichEqClasses = false;
  bool useVBOpt = true, perTranscriptPrior = false, noEffectiveLengthCorrection = false, noLengthCorrection = false, initUniform = false, eqClassMode = false;
  bool useQuasi = false, allowOrphans = true, bootstrapReproject = false; double vbPrior = 1e-2; uint32_t numBootstraps = 0; uint32_t numRequiredFragments = 50000000;
  // (CollapsedGibbsSampler.cpp, the Gibbs pin)
  uint32_t thinningFactor = 16; bool quiet = true, noGammaDraw = false, dontExtrapolateCounts = false; boost::filesystem::path outputDirectory;
  std::shared_ptr<spdlog::logger> jointLog = std::make_shared<spdlog::logger>();
};
