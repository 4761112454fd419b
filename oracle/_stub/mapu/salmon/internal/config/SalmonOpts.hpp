// stand-in: the options initMapperSettings reads (a template that is never instantiated here)
#pragma once
#include <string>
#include "Util.hpp"
struct SalmonOpts { uint32_t maxOccsPerHit; double consensusSlack; pufferfish::util::HitFilterPolicy hitFilterPolicy; uint32_t mismatchSeedSkip; double pre_merge_chain_sub_thresh, post_merge_chain_sub_thresh, orphan_chain_sub_thresh;
  int16_t gapOpenPenalty, gapExtendPenalty, matchScore, mismatchPenalty; int32_t dpBandwidth; bool fullLengthAlignment; double minScoreFraction; bool mimicBT2, mimicStrictBT2, softclipOverhangs, softclip, disableAlignmentCache, allowOrphans, allowDovetail; std::string qmFileName; };
