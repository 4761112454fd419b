// oracle/ref_mapping_utils_shim.cpp — TEST INFRASTRUCTURE.  A C entry point around three functions of the reference's own header, compiled from where it lies
// under /root/reference (never copied) into oracle/_ref/libmappingutils_ref.so by oracle/Makefile:
//   include/salmon/internal/quant/SalmonMappingUtils.hpp   MappingScoreInfo (:82-151), updateRefMappings (:225-281), filterAndCollectAlignments (:283-405),
//                                                           filterAndCollectAlignmentsDecoy (:407-485)
// The pufferfish types that header names (JointMems, MemCluster, QuasiAlignment, ...; pufferfish itself is absent from the tree) are stood in for by
// oracle/_stub/mapu with exactly the members the header touches.  The driver below is the call site's loop (src/quant/SalmonQuantify.cpp:1457-1631):
// every scored candidate goes through updateRefMappings in order, then haveOnlyDecoyMappings decides which collector runs.  `lag` = 1 reproduces the call
// site literally — the slot index is NOT advanced when a candidate is skipped as incompatible (:1521-1523) — so that tests can show what that does; lag = 0
// gives every candidate its own slot (what the checker and the kernels do, SPEC section a7).
// Pins the checker's select_hits (rows a7 / a8) — tests/test_selection_pin.py.
#include "salmon/internal/quant/SalmonMappingUtils.hpp"
#include <cstdint>
#include <limits>
#include <vector>
extern "C" uint32_t ref_select_hits(uint32_t n, const uint32_t* tid, const int32_t* score, const uint8_t* compat, const uint8_t* skipped_incompat, const uint8_t* failed,
                                    uint32_t first_decoy, uint32_t num_targets, double decoy_threshold, int hard_filter, double score_exp, double min_aln_prob, int lag,
                                    uint32_t* kept_slot_out /* the jointHits slot a record was built from */, uint32_t* kept_tid_out, double* prob_out, int32_t* info) {
  using namespace pufferfish::util;
  const int32_t invalidScore = std::numeric_limits<int32_t>::min();
  std::vector<Transcript> transcripts(num_targets); for (uint32_t t = first_decoy; t < num_targets; ++t) transcripts[t].decoy = true;
  std::vector<MemCluster> clusters(n); std::vector<JointMems> jointHits(n);
  for (uint32_t i = 0; i < n; ++i) { clusters[i].first_pos = (int32_t)i; jointHits[i].tid = tid[i]; jointHits[i].leftClust = &clusters[i]; jointHits[i].rightClust = &clusters[i];
    jointHits[i].fragmentLen = (int32_t)i;        // the record carries its slot back to the caller as its fragment length
    jointHits[i].alignmentScore = score[i]; jointHits[i].mateStatus = MateStatus::PAIRED_END_LEFT; }
  salmon::mapping_utils::MappingScoreInfo msi(decoy_threshold); msi.collect_decoys(true); msi.clear(n);
  size_t idx = 0;
  for (uint32_t i = 0; i < n; ++i) {      // SalmonQuantify.cpp:1458-1581
    if (skipped_incompat[i]) { if (!lag) ++idx; continue; }
    if (failed[i]) { ++idx; continue; }
    salmon::mapping_utils::updateRefMappings(tid[i], score[i], compat[i] != 0, idx, transcripts, invalidScore, msi);
    ++idx;
  }
  std::vector<QuasiAlignment> out; const bool bestHitDecoy = msi.haveOnlyDecoyMappings();
  info[0] = msi.bestScore; info[1] = msi.bestDecoyScore; info[2] = bestHitDecoy ? 1 : 0;
  if (msi.bestScore > invalidScore && !bestHitDecoy) salmon::mapping_utils::filterAndCollectAlignments(jointHits, 100, 100, false, true, hard_filter != 0, score_exp, min_aln_prob, msi, out);
  else if (bestHitDecoy) { salmon::mapping_utils::filterAndCollectAlignmentsDecoy(jointHits, 100, 100, false, true, hard_filter != 0, score_exp, min_aln_prob, msi, out); info[2] = 2 + (int32_t)out.size() * 4; out.clear(); }   // (decoy records are written to the mapping file only: their number is reported, they are not alignments of the fragment)
  for (size_t k = 0; k < out.size(); ++k) { kept_slot_out[k] = out[k].fragLen; kept_tid_out[k] = out[k].tid; prob_out[k] = out[k].estAlnProb(); }
  return (uint32_t)out.size();
}
