// oracle/_stub/aln — TEST INFRASTRUCTURE.  Stand-ins on the include path of the alignment-model pin only (oracle/Makefile, ref_alnmodel_shim.cpp): they let
// /root/reference/src/alignment/AlignmentModel.cpp and AlignmentCommon.cpp compile where they lie, without htslib / spdlog / TBB / Boost / pufferfish.
// ReadPair: the two records and the orphan status (include/salmon/internal/alignment/ReadPair.hpp:14-75)
#pragma once
#include "salmon/internal/io/AlignmentIO.hpp"
#include "salmon/internal/util/SalmonUtils.hpp"
struct ReadPair { bam_seq_t* read1 = nullptr; bam_seq_t* read2 = nullptr; salmon::utils::OrphanStatus orphanStatus = salmon::utils::OrphanStatus::Paired;
  inline bool isPaired() const { return orphanStatus == salmon::utils::OrphanStatus::Paired; }
  inline bool isLeftOrphan() const { return orphanStatus == salmon::utils::OrphanStatus::LeftOrphan; }
  inline bool isRightOrphan() const { return orphanStatus == salmon::utils::OrphanStatus::RightOrphan; }
  inline int32_t readLen() const { return read1 ? bam_seq_len(read1) : 0; } };
