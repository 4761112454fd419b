// oracle/_stub/aln — TEST INFRASTRUCTURE.  Stand-ins on the include path of the alignment-model pin only (oracle/Makefile, ref_alnmodel_shim.cpp): they let
// /root/reference/src/alignment/AlignmentModel.cpp and AlignmentCommon.cpp compile where they lie, without htslib / spdlog / TBB / Boost / pufferfish.
#pragma once
