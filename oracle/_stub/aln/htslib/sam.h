// oracle/_stub/aln — TEST INFRASTRUCTURE.  Stand-ins on the include path of the alignment-model pin only (oracle/Makefile, ref_alnmodel_shim.cpp): they let
// /root/reference/src/alignment/AlignmentModel.cpp and AlignmentCommon.cpp compile where they lie, without htslib / spdlog / TBB / Boost / pufferfish.
// htslib's bam1_t as far as include/salmon/internal/io/AlignmentIO.hpp and the two source files use it: the core fields and the variable part
// (name, CIGAR as length << 4 | op, bases two per byte with the first in the high nibble, qualities), bam_cigar_type's table (0x3C1A7: bit 0 = the
// operation consumes the read, bit 1 = the reference), the flag bits and the CIGAR operation numbers of the SAM specification.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
struct bam1_core_t { int32_t pos = 0, tid = 0, mtid = 0; uint16_t flag = 0; uint8_t qual = 0; int32_t l_qseq = 0; uint32_t n_cigar = 0; uint16_t l_qname = 0; };
struct bam1_t { bam1_core_t core; uint8_t* data = nullptr; };
struct sam_hdr_t {}; struct samFile {};
#define BAM_CMATCH 0
#define BAM_CINS 1
#define BAM_CDEL 2
#define BAM_CREF_SKIP 3
#define BAM_CSOFT_CLIP 4
#define BAM_CHARD_CLIP 5
#define BAM_CPAD 6
#define BAM_CEQUAL 7
#define BAM_CDIFF 8
#define BAM_CIGAR_SHIFT 4
#define BAM_CIGAR_MASK 0xf
#define BAM_CIGAR_TYPE 0x3C1A7
#define bam_cigar_type(o) (BAM_CIGAR_TYPE >> ((o) << 1) & 3)
#define BAM_FPAIRED 1
#define BAM_FPROPER_PAIR 2
#define BAM_FUNMAP 4
#define BAM_FMUNMAP 8
#define BAM_FREVERSE 16
#define BAM_FMREVERSE 32
#define BAM_FREAD1 64
#define BAM_FREAD2 128
#define bam_is_rev(b) (((b)->core.flag & BAM_FREVERSE) != 0)
#define bam_get_qname(b) ((char*)(b)->data)
#define bam_get_cigar(b) ((uint32_t*)((b)->data + (b)->core.l_qname))
#define bam_get_seq(b) ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname)
#define bam_get_qual(b) ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname + (((b)->core.l_qseq + 1) >> 1))
#define bam_seqi(s, i) ((s)[(i) >> 1] >> ((~(i) & 1) << 2) & 0xf)
inline bam1_t* bam_init1() { return new bam1_t(); }
inline void bam_destroy1(bam1_t* b) { if (b) { free(b->data); delete b; } }
inline bam1_t* bam_dup1(const bam1_t* b) { bam1_t* d = new bam1_t(*b); d->data = nullptr; return d; }
inline uint8_t* bam_aux_get(const bam1_t*, const char*) { return nullptr; }
inline int64_t bam_aux2i(const uint8_t*) { return 0; }
