// oracle/_stub/mapu/Util.hpp — TEST INFRASTRUCTURE.  Stand-ins for the pufferfish types that the reference's own include/salmon/internal/quant/SalmonMappingUtils.hpp
// names (pufferfish itself — COMBINE-lab/pufferfish @ ace68c1c — is absent from /root/reference).  Only the members that header touches: what a chain
// (MemCluster), a candidate (JointMems) and an emitted alignment (QuasiAlignment) expose to updateRefMappings / filterAndCollectAlignments[Decoy].
// Used by oracle/ref_mapping_utils_shim.cpp only.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <memory>
#include <limits>
namespace pufferfish { namespace util {
enum class MateStatus : uint8_t { SINGLE_END = 0, PAIRED_END_LEFT = 1, PAIRED_END_RIGHT = 2, PAIRED_END_PAIRED = 3 };
enum class PuffAlignmentMode : uint8_t { SCORE_ONLY, APPROXIMATE_CIGAR, EXACT_CIGAR };
enum class HitFilterPolicy : uint8_t { FILTER_AFTER_CHAINING, FILTER_BEFORE_CHAINING, FILTER_BEFORE_AND_AFTER_CHAINING, DO_NOT_FILTER };
struct CIGARGenerator { std::string s; void clear() { s.clear(); } };
struct MemCluster { int32_t first_pos = 0; bool isFw = true; CIGARGenerator cigar; double coverage = 0; int32_t getTrFirstHitPos() const { return first_pos; } };
struct JointMems {
  uint32_t tid = 0; MemCluster* leftClust = nullptr; MemCluster* rightClust = nullptr; int32_t fragmentLen = 0; int32_t alignmentScore = 0, mateAlignmentScore = 0; MateStatus mateStatus = MateStatus::PAIRED_END_PAIRED;
  bool isOrphan() const { return mateStatus != MateStatus::PAIRED_END_PAIRED; }
  bool isLeftAvailable() const { return mateStatus == MateStatus::PAIRED_END_PAIRED || mateStatus == MateStatus::PAIRED_END_LEFT || mateStatus == MateStatus::SINGLE_END; }
  bool isRightAvailable() const { return mateStatus == MateStatus::PAIRED_END_PAIRED || mateStatus == MateStatus::PAIRED_END_RIGHT; }
  MemCluster* orphanClust() const { return isLeftAvailable() ? leftClust : rightClust; }
};
struct QuasiAlignment {
  QuasiAlignment(uint32_t tidIn, int32_t posIn, bool fwdIn, uint32_t readLenIn, CIGARGenerator& cigarIn, uint32_t fragLenIn, bool isPairedIn)
      : tid(tidIn), pos(posIn), fwd(fwdIn), readLen(readLenIn), cigar(cigarIn), fragLen(fragLenIn), isPaired(isPairedIn) {}
  uint32_t tid; int32_t pos; bool fwd; uint32_t readLen; CIGARGenerator cigar; uint32_t fragLen; bool isPaired;
  uint32_t mateLen = 0; CIGARGenerator mateCigar; int32_t matePos = 0; bool mateIsFwd = false; int32_t score = 0, mateScore = 0; uint32_t numHits = 0; MateStatus mateStatus = MateStatus::PAIRED_END_PAIRED;
  double estAlnProb_ = 0.0; void estAlnProb(double p) { estAlnProb_ = p; } double estAlnProb() const { return estAlnProb_; }
};
struct AlignmentConfig { int32_t refExtendLength; bool fullAlignment; int16_t mismatchPenalty; bool bestStrata, decoyPresent; int16_t matchScore, gapExtendPenalty, gapOpenPenalty; double minScoreFraction; bool mimicBT2, mimicBT2Strict, allowOverhangSoftclip, allowSoftclip, useAlignmentCache, noDovetail; PuffAlignmentMode alignmentMode; };
struct MappingConstraintPolicy { bool noOrphans, noDiscordant, noDovetail; double post = 0.9, orph = 0.95; void setPostMergeChainSubThresh(double v) { post = v; } void setOrphanChainSubThresh(double v) { orph = v; } };
struct HitCounters {};
struct QueryCache {};
} }
