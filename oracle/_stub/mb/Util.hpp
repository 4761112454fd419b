// oracle/_stub/mb/Util.hpp — TEST INFRASTRUCTURE.  Stand-ins for the pufferfish types (COMBINE-lab/pufferfish @ ace68c1c, absent from /root/reference) that
// processMiniBatch<QuasiAlignment> (src/quant/SalmonQuantify.cpp:426-1023) touches: MateStatus and the emitted alignment record with the accessors the function
// calls.  The accessors are pufferfish code that cannot be read here (row a6 stays "parity unpinned"); their bodies follow the reference's own alignment-mode twin,
// ReadPair (include/salmon/internal/alignment/ReadPair.hpp:117-168: fragLen / fragLengthPedantic) and the constructor call sites
// (include/salmon/internal/quant/SalmonMappingUtils.hpp:349-383).  On the include path of the mini-batch pin only.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <string>
#include "LibraryFormat.hpp"
namespace pufferfish { namespace util {
enum class MateStatus : uint8_t { SINGLE_END = 0, PAIRED_END_LEFT = 1, PAIRED_END_RIGHT = 2, PAIRED_END_PAIRED = 3 };
enum class HitFilterPolicy : uint8_t { FILTER_AFTER_CHAINING, FILTER_BEFORE_CHAINING, FILTER_BEFORE_AND_AFTER_CHAINING, DO_NOT_FILTER };
struct QuasiAlignment {
  uint32_t tid = 0; int32_t pos = 0; bool fwd = true; uint32_t readLen = 0; uint32_t fragLen = 0; bool isPaired = false;
  uint32_t mateLen = 0; int32_t matePos = 0; bool mateIsFwd = false; int32_t score_ = 0, mateScore_ = 0; MateStatus mateStatus = MateStatus::PAIRED_END_PAIRED;
  uint8_t formatID_ = 0;            // the observed library format of the hit, as the mapping stage recorded it (sq_aln.format_id)
  double logProb = 0.0; double estAlnProb_ = 0.0;
  inline uint32_t transcriptID() const { return tid; }
  inline double estAlnProb() const { return estAlnProb_; }
  inline int32_t hitPos() const { return pos < matePos ? pos : matePos; }
  inline uint32_t fragLength() const { return fragLen; }
  inline uint32_t fragLengthPedantic(uint32_t txpLen) const {       // ReadPair.hpp:149-168: both ends clamped into [0, txpLen]
    if (mateStatus != MateStatus::PAIRED_END_PAIRED || fwd == mateIsFwd) return 0;
    int32_t p1 = fwd ? pos : matePos; p1 = p1 < 0 ? 0 : p1; p1 = p1 > (int32_t)txpLen ? (int32_t)txpLen : p1;
    int32_t p2 = fwd ? matePos + (int32_t)mateLen : pos + (int32_t)readLen; p2 = p2 < 0 ? 0 : p2; p2 = p2 > (int32_t)txpLen ? (int32_t)txpLen : p2;
    return (uint32_t)(p1 > p2 ? p1 - p2 : p2 - p1);
  }
  inline LibraryFormat libFormat() const { return LibraryFormat::formatFromID(formatID_); }
};
} }
