/* salmon_hip.h — C ABI of libsalmon_hip.so: the MI355X-native `salmon quant` hot path.
 *
 * The reference (COMBINE-lab/salmon v1.11.4) has no FFI; it is one statically linked binary.  This
 * header therefore defines the drop-in boundary at the reference's own in-process seams
 * (SURVEY.md §8b).  Each entry point names the reference call site it replaces (paths relative to
 * the reference tree).  Plain pointers and sizes only; no C++ or torch types.
 *
 * All functions return 0 on success or a negative sq_status; they never abort the process
 * (the reference's loop calls std::exit(1) on fatal paths — SalmonQuantify.cpp:1202-1210).
 * sq_last_error() returns a thread-local message for the last failing call.
 */
#ifndef SALMON_HIP_H
#define SALMON_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  SQ_OK = 0,
  SQ_ERR_ARG = -1,      /* bad argument */
  SQ_ERR_IO = -2,       /* file missing / unreadable / bad format (SalmonIndex.hpp:120-160 throws) */
  SQ_ERR_NOMEM = -3,
  SQ_ERR_DEVICE = -4,   /* HIP runtime failure or no gfx950 device (there is NO CPU fallback) */
  SQ_ERR_STATE = -5,    /* call order violated */
  SQ_ERR_OVERFLOW = -6  /* a device work buffer would overflow; batch must be split */
} sq_status;

const char* sq_last_error(void);
const char* sq_version(void); /* "salmon-hip x.y (salmon 1.11.4 semantics)" */

/* ------------------------------------------------------------------------------------------------
 * B0  index handle — replaces checkLoadIndex()/SalmonIndex::load (src/index/BuildSalmonIndex.cpp:264-284,
 *     include/salmon/internal/index/SalmonIndex.hpp:33-67,120-205) and `salmon index`
 *     (BuildSalmonIndex.cpp:49-262 -> pufferfishIndex(IndexOptions&)).
 * ---------------------------------------------------------------------------------------------- */
typedef struct sq_index sq_index;

typedef struct {
  uint32_t k;               /* odd, <= 31 (BuildSalmonIndex.cpp:204-210); 0 -> 31 */
  uint32_t m;               /* minimizer length; 0 -> min(20, max(4, k-4)) (BuildSalmonIndex.cpp:78-81) */
  uint32_t keep_duplicates; /* --keepDuplicates */
  uint32_t no_clip_polya;   /* --no-clip */
  uint32_t threads;         /* -p */
  uint32_t gencode;         /* --gencode: split names at first '|' */
} sq_index_opts;

/* Build from a FASTA file (plain or .gz). decoys_path may be NULL; outdir receives index.bin,
 * info.json, versionInfo.json, duplicate_clusters.tsv. */
int sq_index_build(const sq_index_opts* opts, const char* fasta_path, const char* decoys_path,
                   const char* outdir);
/* [r5] The same from a FASTA file, kept in memory and written nowhere (the -t targets of alignment-based mode: SalmonQuantifyAlignments.cpp reads them
 * into its AlignmentLibrary without an index directory). */
int sq_index_build_fasta_mem(const sq_index_opts* opts, const char* fasta_path, const char* decoys_path, sq_index** out);
/* Build from in-memory sequences (ASCII, not NUL-terminated; lens in nt). Names NUL-terminated.
 * first_decoy = index of the first decoy sequence (== nrefs when none). Either writes outdir (if
 * non-NULL) and/or returns a host-resident handle in *out (if non-NULL). */
int sq_index_build_mem(const sq_index_opts* opts, uint32_t nrefs, const char* const* names,
                       const char* const* seqs, const uint32_t* lens, uint32_t first_decoy,
                       const char* outdir, sq_index** out);
/* [r6] Where index construction builds its k-mer table (the phase of BuildSalmonIndex.cpp:49-262 -> pufferfishIndex that decides where unitigs end): -2 = on GPU 0 when
 * there is one and the input has >= 2*10^7 k-mer positions (the default), -1 = on the host, >= 0 = on that device (a build then fails if it cannot).  The index is
 * byte for byte the same either way.  Process-wide; call before sq_index_build*. */
int sq_index_build_set_device(int device);
/* Load index.bin from dir. device >= 0 uploads the query structures to that GPU's HBM;
 * device < 0 keeps a host-only handle (metadata queries, tests without a GPU). */
int sq_index_load(const char* dir, int device, sq_index** out);
/* Upload an already-built host handle to a device (idempotent). */
int sq_index_to_device(sq_index* idx, int device);
void sq_index_free(sq_index* idx);

uint32_t sq_index_k(const sq_index*);
uint32_t sq_index_m(const sq_index*);
uint32_t sq_index_num_refs(const sq_index*);
uint32_t sq_index_first_decoy(const sq_index*);
const char* sq_index_ref_name(const sq_index*, uint32_t tid);
uint32_t sq_index_ref_len(const sq_index*, uint32_t tid);          /* RefLength (post-clipping) */
uint32_t sq_index_ref_complete_len(const sq_index*, uint32_t tid); /* CompleteLength */
int sq_index_is_decoy(const sq_index*, uint32_t tid);
/* [r4] SHA-2 digests of the records the index was built from, lower-case hex (SalmonIndex.hpp:94-98 <- info.json's SeqHash, NameHash, SeqHash512,
 * NameHash512, DecoySeqHash, DecoyNameHash): which = 0..5 in that order; "" for an index written before they were kept. */
const char* sq_index_hash(const sq_index*, int which);
int sq_index_keeps_duplicates(const sq_index*);   /* --keepDuplicates at build time (info.json "keep_duplicates") */
uint64_t sq_index_num_unitigs(const sq_index*);
uint64_t sq_index_num_kmers(const sq_index*);
uint64_t sq_index_device_bytes(const sq_index*);
/* Raw views of host-side sections (for the checker and tests; do not free). */
typedef struct {
  uint32_t k, m;
  uint32_t num_refs, first_decoy;
  uint64_t num_unitigs, total_unitig_nt, num_kmers, total_ref_nt, num_occ;
  const uint64_t* ref_accum;   /* [num_refs+1] start of each reference in refseq (nt) */
  const uint32_t* ref_len;     /* [num_refs] */
  const uint32_t* ref_clen;    /* [num_refs] complete length */
  const uint64_t* refseq;      /* 2-bit packed, 32 nt per word, little-endian in the word */
  const uint64_t* useq;        /* unitig pool, same packing */
  const uint64_t* uoff;        /* [num_unitigs+1] unitig start (nt) in the pool */
  const uint64_t* ctab_off;    /* [num_unitigs+1] */
  const uint64_t* ctab;        /* occurrences: tid<<32 | ori<<31 | pos */
} sq_index_view;
int sq_index_get_view(const sq_index*, sq_index_view* out);

/* Single k-mer dictionary query on the HOST copy of the SSHash-style dictionary (used by tests to
 * check the dictionary against brute force for every k-mer; the device kernel is the product
 * path).  kmer = 2-bit packed, base i in bits [2i,2i+1].  Returns 1 if found. */
int sq_index_lookup_host(const sq_index*, uint64_t kmer, uint64_t* unitig, uint32_t* offset,
                         int* is_fw);

/* ------------------------------------------------------------------------------------------------
 * Options that define hot-path behaviour (subset of SalmonOpts, include/salmon/internal/config/
 * SalmonOpts.hpp:23-306; defaults from SalmonDefaults.hpp:8-127).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  /* library format (LibraryFormat; src/util/LibraryTypeUtils.cpp:22-46) */
  uint8_t lib_type;        /* 0 single-end, 1 paired-end */
  uint8_t lib_orientation; /* 0 SAME(M), 1 AWAY(O), 2 TOWARD(I), 3 NONE */
  uint8_t lib_strand;      /* 0 SA(SF for PE), 1 AS(SR for PE), 2 S, 3 A, 4 U */
  uint8_t _pad0;
  /* mapping (ProgramOptionsGenerator.cpp:103-289) */
  int32_t match_score;        /* --ma 2 */
  int32_t mismatch_penalty;   /* --mp -4 */
  int32_t gap_open;           /* --go 6 */
  int32_t gap_extend;         /* --ge 2 */
  int32_t bandwidth;          /* --bandwidth 15 */
  uint32_t mismatch_seed_skip;/* 3 */
  uint32_t max_occs_per_hit;  /* 1000 */
  uint32_t max_read_occs;     /* 200 (flag only; see SPEC) */
  uint32_t frag_len_max;      /* --fldMax 1000 */
  double consensus_slack;     /* 0.35 */
  double min_score_fraction;  /* 0.65 */
  double pre_merge_chain_sub_thresh;  /* 0.75 */
  double post_merge_chain_sub_thresh; /* 0.9 */
  double orphan_chain_sub_thresh;     /* 0.95 */
  double score_exp;           /* 1.0 */
  double decoy_threshold;     /* 1.0 */
  double min_aln_prob;        /* 1e-5 */
  uint8_t hard_filter;        /* 0 */
  uint8_t allow_dovetail;     /* 0 */
  uint8_t allow_orphans;      /* 1 (discardOrphansQuasi=false) */
  uint8_t disable_chaining_heuristic; /* 0 */
  uint8_t ignore_incompat;    /* 1 (incompatPrior == 0) */
  uint8_t recover_orphans;    /* 0 (--recoverOrphans, SalmonQuantify.cpp:1356-1364; SPEC §a5) */
  uint8_t lib_autodetect;     /* 0; 1 = `-l A` (LibraryTypeDetector.hpp, SalmonQuantify.cpp:496-501,692-704): lib_* above hold the starting format
                                 (IU paired / U single, nothing is penalised as incompatible meanwhile); once 50 000 alignments of the
                                 library's read type have been seen the most likely format replaces it for the online model (SPEC §D8) */
  uint8_t gc_bias;            /* 0; 1 = --gcBias: collect the observed fragment-GC model while mapping (SalmonQuantify.cpp:938-972; paired-end
                                 libraries); the expected model and the bias-corrected effective lengths come from sq_bias_gc_eff_lengths */
  /* online model (SalmonQuantify.cpp:426-1023) */
  uint32_t mini_batch_size;   /* 5000 (SalmonQuantify.cpp:150) */
  uint32_t num_pre_burnin_frags; /* 5000 */
  uint64_t num_burnin_frags;  /* 5,000,000 */
  double fld_mean, fld_sd;    /* 250, 25 */
  double forgetting_factor;   /* 0.65 */
  double incompat_prior;      /* 0.0 -> ignore_incompat */
  uint32_t range_factorization_bins; /* 4 */
  uint8_t use_frag_len_dist;  /* 1 (!noFragLengthDist) */
  uint8_t model_single_frag_prob; /* 1 (!noSingleFragProb) */
  uint8_t no_length_correction;   /* 0 */
  uint8_t no_eff_length_correction; /* 0 */
  uint64_t seed;              /* seed of the counter-based RNG for FLD sampling (reference: random_device) */
  uint8_t seq_bias;           /* 0; 1 = --seqBias: collect the observed read-start context models (SBModel) from one sampled alignment per paired-end
                                 fragment (SalmonQuantify.cpp:1668-1747); the expected models and the corrected lengths come from sq_bias_eff_lengths */
  uint8_t pos_bias;           /* 0; 1 = --posBias: collect the observed read-start position models by transcript length class (SimplePosBias,
                                 SalmonQuantify.cpp:895-934); the expected models and the corrected lengths come from sq_bias_eff_lengths */
  uint8_t error_model;        /* [r5] alignment-based input only: 1 = the CIGAR-based alignment error model (AlignmentModel.cpp; the reference's default there:
                                 --noErrorModel switches it off): the reads travel with their alignments (sq_aln_inject_reads) */
  uint8_t num_error_bins;     /* 6 (--numErrorBins): read-position bins of the error model's transition matrices */
  uint32_t num_bias_samples;  /* 2,000,000 (SalmonDefaults.hpp numBiasSamples): fragments that contribute to the observed sequence-bias models */
  uint32_t mini_batches_in_flight; /* 8 = the reference's default numThreads (SalmonDefaults.hpp:15): W worker threads each run a mini-batch
                                      against the shared model (SalmonQuantify.cpp:2390-2403); here W consecutive mini-batches read
                                      one model snapshot and their increments are applied in order (SPEC §D1).  1..64; 1 = strictly serial */
} sq_quant_opts;
void sq_quant_opts_default(sq_quant_opts* o); /* -l IU defaults */
/* --mimicBT2 (strict = 0) / --mimicStrictBT2 (strict = 1): the presets processQuantOptions applies on top of whatever else was set
 * (src/util/QuantOptionsUtils.cpp:256-294): maxReadOccs 1000, consensusSlack 0.5, orphans discarded, then ma 2 / mp -4 / go 5 / ge 3, or
 * minScoreFraction 0.8 with ma 1 / mp 0 / go 25 / ge 25.  (Both also switch soft-clipping of overhangs off, which this path never does.)
 * Returns SQ_ERR_ARG for another value of `strict`. */
int sq_quant_opts_mimic_bt2(sq_quant_opts* o, int strict);

/* ------------------------------------------------------------------------------------------------
 * B1  mapping — replaces the worker-loop body between parser->refill(rg) and processMiniBatch
 *     (src/quant/SalmonQuantify.cpp:1199-1854 paired, :2032-2314 single): MemCollector::operator(),
 *     findChains, joinReadsAndFilter, PuffAligner::calculateAlignments [external pufferfish@ace68c1c]
 *     + updateRefMappings / filterAndCollectAlignments (SalmonMappingUtils.hpp:225-485).
 * ---------------------------------------------------------------------------------------------- */
typedef struct sq_ctx sq_ctx;
int sq_ctx_create(sq_index* idx, const sq_quant_opts* opts, int device, uint32_t max_batch_reads,
                  sq_ctx** out);
void sq_ctx_free(sq_ctx*);
/* Forget the online model and the eq-class table (fresh ReadExperiment), keeping work buffers. */
int sq_ctx_reset(sq_ctx*);
/* Pre-size what the END of a job allocates (eq-class export buffers + their page-locked staging area, the EM
 * workspace) for up to max_classes equivalence classes with max_labels label entries; (0, 0) = 10^6 classes, the size
 * the reference gives its eq-class map up front (countMap_.reserve(1000000), EquivalenceClassBuilder.hpp:140), with 6
 * labels each.  Optional: larger jobs grow the buffers as before. */
int sq_ctx_reserve(sq_ctx*, uint64_t max_classes, uint64_t max_labels);

typedef struct {
  uint32_t n;              /* fragments (pairs or single reads) */
  uint32_t paired;         /* 1: seq holds 2n records, record 2i = mate1, 2i+1 = mate2 */
  const uint8_t* seq;      /* concatenated ASCII bases */
  const uint64_t* seq_off; /* [nrec+1] byte offsets into seq */
  int on_device;           /* 0: host pointers (copied H2D); 1: pointers are already in HBM */
} sq_read_batch;

/* mate_status values follow pufferfish::util::MateStatus as used in SalmonMappingUtils.hpp:349-383 */
enum { SQ_MS_SINGLE_END = 0, SQ_MS_PAIRED_END_LEFT = 1, SQ_MS_PAIRED_END_RIGHT = 2,
       SQ_MS_PAIRED_END_PAIRED = 3 };
/* mapping type per fragment (salmon::utils::MappingType, SalmonQuantify.cpp:1604-1630) */
enum { SQ_MT_UNMAPPED = 0, SQ_MT_LEFT_ORPHAN = 1, SQ_MT_RIGHT_ORPHAN = 2, SQ_MT_BOTH_ORPHAN = 3,
       SQ_MT_PAIRED_MAPPED = 4, SQ_MT_SINGLE_MAPPED = 5, SQ_MT_DECOY = 6 };

typedef struct {           /* one QuasiAlignment (fields salmon reads: SURVEY.md §8a row a6) */
  uint32_t tid;
  int32_t pos;             /* implied start of (left/orphan) read on the transcript */
  int32_t mate_pos;
  int32_t score, mate_score;
  uint32_t frag_len;       /* jointHit.fragmentLen (0 for orphans) */
  uint16_t read_len, mate_len;
  uint8_t fwd, mate_fwd, mate_status, format_id; /* LibraryFormat::formatID of the observed hit */
  double est_aln_prob;
} sq_aln;

typedef struct {
  uint32_t n;              /* fragments */
  uint64_t* read_off;      /* [n+1] CSR offsets into aln (caller-owned, n+1 entries) */
  sq_aln* aln;             /* caller-owned, capacity aln_cap */
  uint64_t aln_cap;
  uint8_t* map_type;       /* [n] SQ_MT_* (caller-owned, may be NULL) */
} sq_aln_batch;

typedef struct {           /* HitCounters / MappingStatistics subset (SalmonQuantify.cpp:1861-1865) */
  uint64_t num_reads, num_mapped_at_least_a_kmer, num_with_joint_hits /* upperBoundHits */,
      num_mapped /* >=1 kept alignment */, num_alignments /* validHits */,
      num_mappings_filtered, num_fragments_filtered, num_dovetails, num_decoy_fragments,
      num_seeds, num_lookups, num_mems, num_chains, num_candidates, num_dp_alignments,
      num_orphans_rescued /* fragments with a recovered mate (mstats.numOrphansRescued) */,
      num_truncated_ends /* [r4] always 0: nothing is cut — a batch with a read of more than 1000 bases is refused (SQ_ERR_ARG), longer strides and uni-MEM slabs are taken as needed (SPEC §I, §a1) */;
} sq_map_stats;

/* Map one batch. Results stay resident on the device for sq_eq_accumulate(); if out != NULL they
 * are also copied to the caller's buffers (SQ_ERR_OVERFLOW if aln_cap is too small). */
int sq_map_batch(sq_ctx*, const sq_read_batch* in, sq_aln_batch* out, sq_map_stats* stats);
/* Pipelined form of sq_map_batch: submit queues a batch on the next mapping lane (a worker thread with
 * its own HIP stream and work buffers; two lanes by default) and returns at once; wait returns the
 * batches in submission order, after which sq_eq_accumulate / sq_debug_tap refer to that batch, exactly
 * as after sq_map_batch.  Keeping two batches in flight lets their kernels fill each other's stalls.
 * `in` (and `out`, if given) must stay valid until the matching sq_map_wait returns.  This replaces
 * the reference's N worker threads pulling chunks from the parser (SalmonQuantify.cpp:2390-2403). */
int sq_ctx_set_lanes(sq_ctx*, int lanes /* 1..4, default 2 */);
int sq_map_submit(sq_ctx*, const sq_read_batch* in, sq_aln_batch* out /* may be NULL */);
int sq_map_wait(sq_ctx*, sq_aln_batch* out /* may be NULL */, sq_map_stats* stats);
/* Alignments of the batch sq_map_wait / sq_map_batch returned last, copied out of HBM on demand (they stay in that lane's
 * buffers until the lane maps again).  out->read_off == out->aln == NULL: only out->n and out->aln_cap (= alignments
 * held) are filled — the size query the SAM writer makes before it sizes its arrays. */
int sq_map_fetch(sq_ctx*, sq_aln_batch* out);   /* out->map_type alone (read_off == aln == NULL): only the per-fragment mapping types */
/* [r4] alignment-based mode (`salmon quant -a`, src/alignment/SalmonQuantifyAlignments.cpp:125-937): a batch of alignments that came from a SAM file
 * (sq_sam_* below) takes the place of a mapped batch — in->n fragments, in->read_off[n + 1] a prefix sum into in->aln — and sq_eq_accumulate
 * runs the same online model / equivalence-class stage on it.  num_with_joint_hits = fragments with at least one alignment record. */
int sq_aln_inject(sq_ctx* ctx, const sq_aln_batch* in, uint64_t num_with_joint_hits);
/* The record source of alignment-based mode: a name-collated SAM text file (plain or gzip) or BAM file (decoded here: no htslib), read as the reference's
 * BAMQueue reads it (include/salmon/internal/alignment/BAMQueue.tpp:288-600): proper pairs on one target -> pair alignments, a mapped read whose mate
 * is not -> an orphan, consecutive alignments of one read name -> one fragment ordered by transcript.  paired_library: 1 = ReadPair rules, 0 = every
 * mapped record is a single-end alignment.  sq_sam_set_tid_map: SAM target i -> transcript id of the index (0xFFFFFFFF: skip its alignments); the
 * default is the identity.  sq_sam_next: up to max_frags fragments into arrays the reader owns (valid until the next call); out->n == 0 at the end.
 * use_as_scores: est_aln_prob = exp(-score_exp (bestAS - AS)) within a fragment (--useASWithoutCIGAR, SalmonQuantifyAlignments.cpp:516-521), else 1
 * (--noErrorModel).  The CIGAR-based error model of alignment mode (AlignmentModel.hpp) is not built. */
typedef struct sq_sam sq_sam;
typedef struct { uint64_t num_records, num_fragments, num_alignments, num_unaligned, num_suspicious_pairs, num_skipped_unknown_target, num_frags_without_as; } sq_sam_counts;
int sq_sam_open(const char* path, int paired_library, sq_sam** out);
int sq_sam_first_flag(const char* path, int* flag);   /* [r5] FLAG of the first record (SAM, gzip, BAM): -l A decides paired / single-end from it (the reference peeks at the file too: SalmonQuantifyAlignments.cpp, AlignmentLibrary's constructor) */
uint32_t sq_sam_num_refs(const sq_sam*);
const char* sq_sam_ref_name(const sq_sam*, uint32_t i);
uint32_t sq_sam_ref_len(const sq_sam*, uint32_t i);
int sq_sam_set_tid_map(sq_sam*, const uint32_t* map, uint32_t n);
int sq_sam_next(sq_sam*, uint32_t max_frags, int use_as_scores, double score_exp, sq_aln_batch* out, sq_sam_counts* counts);
void sq_sam_close(sq_sam*);
/* [r5] The reads behind the alignments of a batch, for the CIGAR-based error model (src/alignment/AlignmentModel.cpp): per alignment a two records — [2a] the one
 * AlignmentModel scores with its "left" transition matrices (a pair's record with the smaller position — on a tie the file's second —, a left orphan, a
 * single-end read), [2a + 1] the one scored with the "right" ones (empty where there is none).  A record = its leftmost reference position, its CIGAR
 * operations in BAM encoding (length << 4 | op) and its bases as the file stores them (reference strand), one byte each, 0..3 = ACGT (anything else: 0,
 * as the reference's samToTwoBit maps it).  aligner_score: the sum of the fragment's AS tags when the file's @PG line names bowtie2, else 0 — the weight p
 * of AlignmentModel::update (SalmonQuantifyAlignments.cpp:265-285).  sq_sam_keep_reads(reader, 1) before the first sq_sam_next; sq_sam_reads returns the
 * arrays of the batch sq_sam_next returned last (valid until the next call). */
typedef struct {
  uint64_t num_alignments;
  const uint64_t* cig_off;       /* [2 * num_alignments + 1] */
  const uint32_t* cigar;
  const uint64_t* seq_off;       /* [2 * num_alignments + 1] */
  const uint8_t* seq;
  const int32_t* pos;            /* [2 * num_alignments] */
  const int32_t* aligner_score;  /* [num_alignments] */
} sq_aln_reads;
int sq_sam_keep_reads(sq_sam*, int on);
int sq_sam_reads(sq_sam*, sq_aln_reads* out);
/* sq_aln_inject with the reads: required when sq_quant_opts.error_model is set (the online stage then weighs every alignment by
 * AlignmentModel::logLikelihood and, until the burn-in ends, learns the transition matrices from the sampled alignments: SalmonQuantifyAlignments.cpp:516-523, :861-864) */
int sq_aln_inject_reads(sq_ctx* ctx, const sq_aln_batch* in, const sq_aln_reads* reads, uint64_t num_with_joint_hits);

/* ------------------------------------------------------------------------------------------------
 * B2  equivalence classes — replaces processMiniBatch (SalmonQuantify.cpp:426-1023) +
 *     EquivalenceClassBuilder::addGroup/finish (EquivalenceClassBuilder.hpp:165-181,237-250).
 * ---------------------------------------------------------------------------------------------- */
/* Host read pipeline in front of the mapping seam (replaces the FQFeeder parser threads and chunk queues,
 * SalmonQuantify.cpp:2419-2443): one producer thread per mate stream inflates (.gz or plain) and splits
 * FASTQ/FASTA records; sq_reader_next interleaves the mates into a read batch in one of `num_slots`
 * rotating page-locked host buffers (layout = sq_read_batch, on_device = 0).  A batch stays valid until
 * sq_reader_release(reader, slot).  files2 == NULL / n2 == 0: single-end.  End of input: batch->n == 0. */
typedef struct sq_reader sq_reader;
int sq_reader_open(const char* const* files1, uint32_t n1, const char* const* files2, uint32_t n2,
                   uint32_t batch_reads, uint32_t num_slots, sq_reader** out);
/* flags: SQ_READER_KEEP_NAMES also keeps the read names of the mate-1 stream (header up to the first blank) for
 * sq_reader_names — the SAM writer (--writeMappings) and --writeUnmappedNames ask for them. */
#define SQ_READER_KEEP_NAMES 1u
int sq_reader_open_ex(const char* const* files1, uint32_t n1, const char* const* files2, uint32_t n2,
                      uint32_t batch_reads, uint32_t num_slots, uint32_t flags, sq_reader** out);
int sq_reader_next(sq_reader*, sq_read_batch* batch, int* slot);
/* names of the batch held in `slot`: name i = names[name_off[i] .. name_off[i+1]); valid until the slot is released */
int sq_reader_names(const sq_reader*, int slot, const char** names, const uint64_t** name_off);
void sq_reader_release(sq_reader*, int slot);
uint64_t sq_reader_total(const sq_reader*);
void sq_reader_close(sq_reader*);

/* Run the online model over the batch last mapped by sq_map_batch (mini-batches of
 * opts.mini_batch_size in input order) and add its fragments to the device eq-class table. */
int sq_eq_accumulate(sq_ctx*);

typedef struct {
  uint64_t num_classes;   /* E */
  uint64_t num_labels;    /* L = sum of class sizes (transcripts only) */
  uint64_t* off;          /* [E+1] */
  uint32_t* tid;          /* [L] */
  double* w;              /* [L] normalised aux weights (TGValue::normalizeAux) */
  uint64_t* wq;           /* [L] raw fixed-point weight sums (SQ_WFRAC_BITS) — exact reduction form */
  uint64_t* count;        /* [E] */
  uint32_t* bins;         /* [L] range-factorization bin ids (label tail), or NULL */
  uint64_t* h1;           /* [E] label hash (canonical class order = ascending (first tid, h1, h2)) */
  uint64_t* h2;
} sq_eq_table;
/* Query sizes (out arrays NULL) then fetch into caller buffers. Classes come out in canonical order. */
int sq_eq_finish(sq_ctx*, sq_eq_table* out);
/* Merge an externally provided table (e.g. all-gathered from other GPUs) into this ctx's table:
 * counts and fixed-point weight sums add exactly, so any merge order gives identical bits. */
int sq_eq_merge(sq_ctx*, const sq_eq_table* other);
/* Device-resident forms for the multi-GPU reduction: sq_eq_export_device fills `out` with DEVICE pointers to
 * the canonical-order export (valid until the next accumulate / merge / reset); sq_eq_merge_device merges a table
 * whose arrays are device pointers on this ctx's GPU (e.g. the receive buffers of an RCCL all_gather), so the
 * tables never bounce through host memory. */
int sq_eq_export_device(sq_ctx*, sq_eq_table* out);
int sq_eq_merge_device(sq_ctx*, const sq_eq_table* t);

typedef struct {          /* online-model state needed downstream (Transcript, FLD, counters) */
  uint64_t num_observed, num_assigned, num_mapped_ub;
  int burned_in;
  uint64_t num_compatible;   /* assigned fragments with at least one library-compatible alignment (SalmonQuantify.cpp:811-815) */
  uint32_t lib_format_id;    /* format the online model expects now: type | orientation << 1 | strandedness << 3 (LibraryFormat::formatID) */
  uint32_t lib_detected;     /* 1 once `-l A` auto-detection has replaced the starting format */
} sq_model_summary;
int sq_model_summary_get(sq_ctx*, sq_model_summary* out);
/* [r4] multi-GPU, the shared burn-in prefix (oracle/SPEC.md §MG): every rank maps and accumulates the same batches until the model is burned in
 * (sq_model_summary.burned_in), so all ranks hold the same fragment-length model and effective lengths — the reference learns them ONCE too; the ranks
 * other than 0 then call this: the classes, counts and observed bias masses the prefix added are forgotten (rank 0 keeps them), the model is kept, the
 * transcript masses move into the prior term.  After it the rank's table holds only what it maps from here on, and the merged table of an N-rank job
 * is the one-rank job's table bit for bit. */
int sq_model_drop_counts(sq_ctx*);
/* per-transcript state after the online phase: log-mass (LOG_0 = +inf when none), unique/total
 * counts, log effective length (Transcript.hpp:136-141,210-283). Arrays of length num_refs. */
int sq_model_fetch(sq_ctx*, double* log_mass, uint64_t* unique_count, uint64_t* total_count,
                   double* log_eff_len);
int sq_model_fetch_fld(sq_ctx*, double* log_pmf_1001); /* log PMF bins 0..1000 (flenDist) */
int sq_model_fld_min(sq_ctx*, uint32_t* min_len);      /* [r4] FragmentLengthDistribution::minVal(): the smallest fragment length added so far, 1 when none (FragmentLengthDistribution.cpp:78-83) */
/* fragments per observed library format id (type | orientation << 1 | strandedness << 3), 64 slots
 * (ReadLibrary::libTypeCounts, SalmonQuantify.cpp:1000-1021). */
int sq_model_fetch_lib_counts(sq_ctx*, uint64_t* counts64);
/* lib_format_counts.json (ReadExperiment::summarizeLibraryTypeCounts, ReadExperiment.inl:219-348). */
int sq_write_lib_format_counts(const char* path, const char* read_files, uint8_t lib_type, uint8_t lib_orientation, uint8_t lib_strand,
                               const uint64_t* counts64, uint64_t num_assigned, uint64_t num_compatible);

/* ------------------------------------------------------------------------------------------------
 * B3  inference — replaces CollapsedEMOptimizer::optimize (src/inference/CollapsedEMOptimizer.cpp:
 *     732-1035), ::gatherBootstraps (:554-690), CollapsedGibbsSampler::sample
 *     (src/inference/CollapsedGibbsSampler.cpp:317-508).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  uint8_t use_vbem;             /* useVBOpt = true */
  uint8_t per_transcript_prior; /* true */
  uint8_t init_uniform;         /* false; forced true in -e mode */
  uint8_t eq_class_mode;        /* combined weight = file weight verbatim (:862) */
  uint8_t no_rich_eq_classes;
  uint8_t alt_init_mode;        /* --alternativeInitMode / --meta: mix the online estimate with (uniqueCount + 0.5) * 1e-3 * effLen instead
                                   of the uniform abundance (CollapsedEMOptimizer.cpp:790-792, 817-818); needs sq_txp_in.unique_count */
  uint8_t _pad[2];
  double vb_prior;              /* 1e-2 */
  double rel_diff_tolerance;    /* 0.01 */
  uint32_t max_iter;            /* 10000 */
  uint32_t min_iter;            /* 100 */
  double num_required_fragments;/* 5e7 (deprecated knob, still used for init mixing :790) */
} sq_em_opts;
void sq_em_opts_default(sq_em_opts*);

typedef struct {
  uint32_t num_txp;             /* M */
  const double* projected_counts; /* [M] alphas from normalizeAlphas (ignored with init_uniform) */
  const uint64_t* unique_count; /* [M] (altInitMode only; may be NULL) */
  const double* eff_len;        /* [M] linear-space effective lengths */
} sq_txp_in;

typedef struct {
  uint32_t iters;
  int converged;
  double max_rel_diff;
  double alpha_sum;
  double device_ms;             /* HIP-event time of the iteration loop */
  double ms_per_iter;
  uint32_t num_degenerate;      /* classes dropped by markDegenerateClasses (CollapsedEMOptimizer.cpp:330-394): sum_i alpha0[tid_i] * w_i <= DBL_MIN;
                                   they take no part in the optimisation ("Marked {} weighted equivalence classes as degenerate") */
  uint32_t _pad;
} sq_em_report;

/* Full optimisation. eq arrays are host pointers (copied) — tid/w/count/off as in sq_eq_table.
 * eq == NULL: optimise over the classes the ctx itself has accumulated (the reference passes the
 * experiment, which owns its eq-classes: optimizer.optimize(experiment, ...)); they are read from the
 * canonical-order export already resident in HBM, nothing is re-uploaded. */
int sq_em_optimize(sq_ctx*, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* opts,
                   double* alpha_out /*[M]*/, sq_em_report* report);
/* Standalone variant that needs no index/ctx (the `salmon quant -e` seam,
 * SalmonQuantifyAlignments.cpp:1407-1441). */
int sq_em_optimize_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp,
                       const sq_em_opts* opts, double* alpha_out, sq_em_report* report);
/* Run exactly `iters` update steps from the given alpha (benchmarking / parity of single steps). */
int sq_em_steps_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* opts,
                    const double* alpha_in, uint32_t iters, double* alpha_out, sq_em_report* report);

/* ------------------------------------------------------------------------------------------------
 * Bias-corrected effective lengths (row f-3; --gcBias).  The observed fragment-GC model is collected by the online stage when
 * sq_quant_opts.gc_bias is set (observedGCMass, SalmonQuantify.cpp:938-972); the expected model and the corrected lengths replace
 * salmon::utils::updateEffectiveLengths (src/util/SalmonUtils.cpp:1208-1985, gcBiasCorrect branches), which the optimizer calls at
 * iteration > 10 (CollapsedEMOptimizer.cpp:901-928): sq_em_optimize_bias does that call through `cb`.
 * ---------------------------------------------------------------------------------------------- */
int sq_model_fetch_gc_observed(sq_ctx*, double* out /*[3][25] conditional bin x fragment-GC bin, linear-space masses*/);
typedef struct { uint32_t num_processed; int32_t fld_low, fld_high; uint32_t _pad; double gc_bias_row0[25]; } sq_bias_report;
/* idx must be on a device: the sweep over (transcript, fragment start, sampled length) runs there. */
int sq_bias_gc_eff_lengths(sq_index* idx, const double* gc_observed /*[75]*/, const double* log_pmf_1001, uint32_t num_txp,
                           const double* alphas, const double* eff_len_in, double* eff_len_out, sq_bias_report* report);
/* --seqBias, alone (use_gc = 0) or with --gcBias: the expected read-start context models (SBModel), the expected GC model with its context
 * bins, and the corrected lengths (SalmonUtils.cpp:1576-1600, 1810-1960).  seq_fw / seq_rc: the observed context counts collected by the online
 * stage (sq_model_fetch_seq_observed).  models_out (may be NULL) receives the four normalised log-probability tables [expected fw, expected rc,
 * observed fw, observed rc][9 positions x 64 contexts]. */
int sq_model_fetch_seq_observed(sq_ctx*, uint64_t* fw576, uint64_t* rc576, uint64_t* num_samples);
int sq_bias_seq_eff_lengths(sq_index* idx, int use_gc, const double* gc_observed /*[75] or NULL*/, const uint64_t* seq_fw /*[576]*/, const uint64_t* seq_rc,
                            const double* log_pmf_1001, uint32_t num_txp, const double* alphas, const double* eff_len_in, double* eff_len_out,
                            double* models_out, sq_bias_report* report);
/* --posBias and every combination of the three corrections (SalmonUtils.cpp:1639-1652, 1708-1712, 1815-1835, 1941-1944; SimplePosBias.cpp).
 * pos_observed: sq_model_fetch_pos_observed's [5' model, 3' model][5 length classes][20 bins]; threads = the reference's numThreads (every
 * bin of the shared model and of each worker's local copy starts with mass 1, so a model bin starts at 1 + threads).  NULL members switch a
 * correction off; --gcBias alone takes sq_bias_gc_eff_lengths' path.  seq_models_out as in sq_bias_seq_eff_lengths; pos_models_out (may be
 * NULL) receives the normalised bin masses [observed 5', observed 3', expected 5', expected 3'][5][20] (SimplePosBias::writeBinary's values). */
typedef struct { const double* gc_observed; const uint64_t* seq_fw; const uint64_t* seq_rc; const double* pos_observed; uint32_t threads; uint32_t _pad; } sq_bias_models;
int sq_model_fetch_pos_observed(sq_ctx*, double* out200);
/* [r5] the CIGAR error model's transition matrices as learned so far (AlignmentModel::transitionProbsLeft_ / Right_, AtomicMatrix storage_ and rowsums_, log space):
 * cells [2][bins][82][82], rows [2][bins][82] (either may be NULL); *bins = the number of read-position bins */
int sq_model_fetch_error_model(sq_ctx*, double* cells, double* rows, uint32_t* bins);
/* [r4] the expected fragment-GC masses [3 context classes][25 bins] (linear) of the calling thread's last sq_bias_*eff_lengths sweep with a GC model:
 * what aux_info/exp_gc.gz holds (GZipWriter.cpp:405-413) */
int sq_bias_last_gc_expected(double* out75);
int sq_bias_eff_lengths(sq_index* idx, const sq_bias_models* models, const double* log_pmf_1001, uint32_t num_txp, const double* alphas, const double* eff_len_in,
                        double* eff_len_out, double* seq_models_out, double* pos_models_out, sq_bias_report* report);
/* Transcript::lengthClassIndex (ReadExperiment.inl:352-388): the length quantiles of the non-decoy references and the class of every reference;
 * returns the number of classes (5, or the number of references when there are no more than 5), < 0 on error. */
int sq_index_length_classes(const sq_index* idx, uint32_t* quantiles5, uint8_t* cls /* [num_refs] or NULL */);
/* updateEffectiveLengths as a callback: alphas and current effective lengths in, new effective lengths out; non-zero aborts. */
typedef int (*sq_efflen_cb)(const double* alphas, const double* eff_len_in, double* eff_len_out, uint32_t m, void* user);
/* sq_em_optimize with the bias hook: after 11 updates (itNum > 10) — or at the first update after which the convergence test holds, if that
 * comes earlier (CollapsedEMOptimizer.cpp:901: `itNum > targetIt or converged`) — `cb` is called once, priors and combined class weights are rebuilt
 * from the new effective lengths (updateEqClassWeights, CollapsedEMOptimizer.cpp:160-176) and the iteration goes on.
 * eff_len_out (may be NULL) receives the lengths the optimisation ended with (Transcript::EffectiveLength, :1024-1027). */
int sq_em_optimize_bias(sq_ctx*, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* opts, sq_efflen_cb cb, void* user,
                        double* alpha_out, double* eff_len_out, sq_em_report* report);

/* The learning-rate schedule of the online phase (ForgettingMassCalculator.hpp:23-40: prefill; :64-90 getLogMassAndTimestep as called at
 * SalmonQuantify.cpp:515): out[b] = log forgetting mass of mini-batch b for b < n.  sq_eq_accumulate uses exactly these values. */
int sq_forgetting_masses(double forgetting_factor, uint64_t n, double* out);

/* Host-side, once per run: salmon::utils::normalizeAlphas (src/util/SalmonUtils.cpp:461-529) with
 * TranscriptCluster::projectToPolytope — online masses -> projectedCounts used to initialise EM. */
int sq_normalize_alphas(uint32_t num_txp, const sq_eq_table* eq, const double* log_mass, const uint64_t* unique_count,
                        const uint64_t* total_count, double* projected_out);
/* B4 output files: quant.sf (GZipWriter.cpp:684-739; num_mapped_frags <= 0 -> explicit sum) and
 * aux_info/eq_classes.txt.gz (GZipWriter.cpp:64-168). */
int sq_write_quant_sf(const char* path, const sq_index* idx, const double* eff_len, const double* num_reads, double num_mapped_frags);
/* the same with --sigDigits decimals for EffectiveLength and NumReads (GZipWriter.cpp:734-736; default 3) */
int sq_write_quant_sf_digits(const char* path, const sq_index* idx, const double* eff_len, const double* num_reads, double num_mapped_frags, int sig_digits);
int sq_write_eq_classes(const char* path, const sq_index* idx, const sq_eq_table* eq, int with_weights);

/* [r4] the rest of aux_info (GZipWriter::writeMeta, src/output/GZipWriter.cpp:294-599).
 * sq_write_fld_samples = aux_info/fld.gz: distribution_utils::samplesFromLogPMF (src/util/DistributionUtils.cpp:57-102) over the log PMF's bins
 *   [min_len, max_len] (FragmentLengthDistribution::minVal / maxVal; sq_model_fld_min gives the former) — `num_samples` (the reference: 10 000) draws
 *   histogrammed into int32[max_len + 1], gzipped raw; also returns the summary's mean / sd / support (meta_info's frag_length_mean, frag_length_sd,
 *   frag_dist_length).  path may be NULL (summary only).  The reference draws from a randomly seeded Mersenne twister; here draw j is a pure function of (seed, j).
 * sq_write_legacy_bias = expected_bias.gz, observed_bias.gz, observed_bias_3p.gz (:335-351): the 4^6-entry tables of the retired k-mer bias model at their
 *   initial values (nothing updates them in the reference either); *num_bias_bins = 4096.
 * sq_write_gc_model / sq_write_seq_model / sq_write_pos_models = GCFragModel::writeBinary (GCFragModel.hpp:63-79), SBModel::writeBinary (SBModel.cpp:77-116)
 *   and the positional-model file of GZipWriter.cpp:424-447 (SimplePosBias::writeBinary, SimplePosBias.cpp:84-101), gzipped. */
int sq_write_fld_samples(const char* path, const double* log_pmf, uint32_t min_len, uint32_t max_len, uint32_t num_samples, uint64_t seed,
                         double* mean_out, double* sd_out, uint32_t* support_out);
int sq_write_legacy_bias(const char* aux_dir, uint32_t* num_bias_bins);
int sq_write_gc_model(const char* path, int32_t dtype /* 0 linear, 1 log */, uint32_t rows, uint32_t cols, const double* totals /*[rows]*/, const double* counts /*[rows][cols]*/);
int sq_write_seq_model(const char* path, const double* log_probs_9x64 /* position-major: sq_bias_eff_lengths' seq_models_out rows */);
int sq_write_pos_models(const char* path, uint32_t num_models, const uint32_t* len_bounds, uint32_t model_len, const double* masses /*[num_models][model_len]*/);
/* aux_info/meta_info.json with GZipWriter::writeMeta's keys (:497-597) in its order.  Strings may be NULL (written as ""). */
typedef struct {
  const char* salmon_version;   /* NULL -> "1.11.4" (the semantics this library follows) */
  const char* samp_type;        /* "none" | "gibbs" | "bootstrap" */
  const char* opt_type;         /* "vb" | "em" | "none" */
  const char* quant_errors;     /* NULL / "" -> []; else one entry (writeEmptyMeta, :180-290) */
  uint32_t num_libraries; uint32_t frag_dist_length;
  const char* const* library_types;   /* [num_libraries], LibraryFormat::toString() */
  double frag_length_mean, frag_length_sd;
  int32_t seq_bias_correct, gc_bias_correct, pos_bias_correct;
  uint32_t num_bias_bins;
  const char* mapping_type;     /* "mapping" | "alignment" */
  int32_t keep_duplicates;      /* 1 / 0; < 0 = unknown: no key (:518-529) */
  int32_t serialized_eq_classes, range_factorized, scalar_weights;
  uint64_t num_valid_targets, num_decoy_targets, num_eq_classes;
  uint32_t num_length_classes; uint32_t _pad0;
  const uint32_t* length_classes;     /* ReadExperiment::getLengthQuantiles (sq_index_length_classes) */
  const char *index_seq_hash, *index_name_hash, *index_seq_hash512, *index_name_hash512, *index_decoy_seq_hash, *index_decoy_name_hash;
  uint64_t num_bootstraps, num_processed, num_mapped, num_decoy_fragments, num_dovetail_fragments, num_fragments_filtered_vm, num_alignments_below_threshold_vm;
  double percent_mapped;
  const char *start_time, *end_time;  /* asctime-style, as SalmonOpts::runStartTime */
  /* not reference keys: written last, under one "salmon_hip" object, when backend != NULL */
  const char* backend; uint32_t num_em_iterations, num_degenerate_eq_classes; double runtime_s;
} sq_meta_info;
int sq_write_meta_info(const char* path, const sq_meta_info* m);

/* aux_info/ambig_info.tsv (GZipWriter.cpp:601-638): UniqueCount / AmbigCount per transcript from the eq-classes. */
int sq_write_ambig_info(const char* path, uint32_t m, const sq_eq_table* eq);

/* quant.sf from plain name / length arrays (`salmon quant -e` has no index; lens may be NULL). */
int sq_write_quant_sf_names(const char* path, uint32_t m, const char* const* names, const uint32_t* lens,
                            const double* eff_len, const double* num_reads, double num_mapped_frags);

/* `salmon quant -e eq_classes.txt[.gz]` input (salmon::utils::readEquivCounts, SalmonUtils.cpp:1026-1122;
 * consumed by processEqClasses, SalmonQuantifyAlignments.cpp:1407-1441): the file written with
 * --dumpEqWeights, optionally followed by "name effLen" pairs (missing ones default to 100).
 * sq_eq_file_table fills off/tid/w/count with pointers owned by the file object. */
typedef struct sq_eq_file sq_eq_file;
int sq_eq_file_read(const char* path, sq_eq_file** out);
void sq_eq_file_free(sq_eq_file*);
uint32_t sq_eq_file_num_txp(const sq_eq_file*);
const char* sq_eq_file_name(const sq_eq_file*, uint32_t i);
const double* sq_eq_file_eff_lens(const sq_eq_file*);
int sq_eq_file_table(const sq_eq_file*, sq_eq_table* out);

/* aux_info/bootstrap/names.tsv.gz + bootstraps.gz (raw f64[M] per replicate, GZipWriter.cpp:306-326,765-788);
 * sq_boot_writer_append has the signature shape of the replicate callback below. */
typedef struct sq_boot_writer sq_boot_writer;
int sq_boot_writer_open(const char* aux_dir, uint32_t m, const char* const* names, sq_boot_writer** out);
int sq_boot_writer_append(sq_boot_writer*, const double* alphas, uint32_t m);
uint64_t sq_boot_writer_close(sq_boot_writer*);   /* returns the number of replicates written */

typedef int (*sq_replicate_cb)(const double* alphas, uint32_t m, void* user);
int sq_bootstrap_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* opts,
                     uint32_t num_bootstraps, uint64_t seed, uint64_t num_mapped,
                     sq_replicate_cb cb, void* user);
/* Replicates [first, first + count) of num_bootstraps: the same bytes whichever GPU computes them (counter RNG keyed by the replicate). */
int sq_bootstrap_range_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_em_opts* opts,
                           uint32_t num_bootstraps, uint32_t first, uint32_t count, uint64_t seed, uint64_t num_mapped,
                           sq_replicate_cb cb, void* user);
typedef struct {
  uint32_t thinning_factor;     /* 16 */
  uint8_t no_gamma_draw;
  uint8_t use_vbem;
  uint8_t per_transcript_prior;
  uint8_t _pad;
  double vb_prior;
} sq_gibbs_opts;
int sq_gibbs_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_gibbs_opts* opts,
                 const double* alpha_init, uint32_t num_samples, uint64_t seed, uint64_t num_mapped,
                 sq_replicate_cb cb, void* user);
/* Samples [first, first + count) of num_samples; `first` must start a chain (a multiple of sq_gibbs_chain_step(num_samples)). */
uint32_t sq_gibbs_chain_step(uint32_t num_samples);
int sq_gibbs_range_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_gibbs_opts* opts,
                       const double* alpha_init, uint32_t num_samples, uint32_t first, uint32_t count, uint64_t seed,
                       uint64_t num_mapped, sq_replicate_cb cb, void* user);

/* The same with the device time of the sampling rounds (HIP events around each sample's thinning rounds): measurement for configs[4]. */
typedef struct { uint64_t rounds; double device_ms, ms_per_round; uint64_t draws_per_round; /* categorical draws of one round = fragments in multi-label classes */
                 uint32_t items[3]; /* work items (<= 256 draws of one class) by class size: <= 8, <= 16, larger */ uint32_t _pad; } sq_gibbs_report;
int sq_gibbs_range_report_dev(int device, const sq_eq_table* eq, const sq_txp_in* txp, const sq_gibbs_opts* opts,
                              const double* alpha_init, uint32_t num_samples, uint32_t first, uint32_t count, uint64_t seed,
                              uint64_t num_mapped, sq_replicate_cb cb, void* user, sq_gibbs_report* report);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU (oracle/SPEC.md §MG): one process per GPU.  Reads shard by rank — there is no collective on the mapping path;
 * after mapping ONE exchange of the equivalence-class tables over RCCL (xGMI inside a node), the per-transcript model state reduced
 * by a defined rule, the inference tail replicated, posterior replicates sharded by rank.  The reference has no multi-process mode;
 * this seam sits where its worker threads join (SalmonQuantify.cpp:2445-2480) and where doBootstrap / the Gibbs chains fan out
 * (CollapsedEMOptimizer.cpp:554-690, CollapsedGibbsSampler.cpp:425-470).
 * ---------------------------------------------------------------------------------------------- */
#define SQ_DIST_ID_BYTES 128
typedef struct sq_dist sq_dist;
int sq_dist_make_id(uint8_t* id128);              /* rank 0: an RCCL unique id; the launcher hands it to every rank (file, pipe, MPI, torch) */
int sq_dist_init(const uint8_t* id128, int rank, int world, int device, sq_dist** out);   /* collective: all ranks call it */
void sq_dist_free(sq_dist*);
int sq_dist_rank(const sq_dist*);
int sq_dist_world(const sq_dist*);
/* Collective. All-gathers every rank's canonical-order class table (HBM -> xGMI -> HBM) and merges the others' into this ctx: every
 * rank ends with the same table; counts and fixed-point weight sums add exactly, so the bits do not depend on the gather order. */
int sq_dist_merge_eq(sq_dist*, sq_ctx*);
/* The same exchange when the box has fewer GPUs than ranks (tests, bring-up): `n` contexts on the communicator's device stand for the ranks of an
 * n-rank job.  Needs a communicator of ONE rank; every context's table is packed, sent through the size and payload all-gathers of that
 * communicator (RCCL executes them) into a receive slot, and every context merges the other contexts' slots as sq_dist_merge_eq does. */
int sq_dist_merge_eq_loopback(sq_dist*, sq_ctx* const* ctxs, uint32_t n);
/* Collective. SPEC §MG: unique / total counts add; masses combine by logAdd in rank order; effective lengths are rank 0's. */
int sq_dist_reduce_model(sq_dist*, uint32_t num_txp, double* log_mass, uint64_t* unique_count, uint64_t* total_count, double* log_eff_len);
/* The mass rule on its own (host arithmetic, no device): row r of all_log_mass = rank r's log-masses; out[t] = logAdd over r = 0..R-1. */
int sq_merge_log_masses(uint32_t num_txp, uint32_t num_ranks, const double* all_log_mass /*[R][M]*/, double* out /*[M]*/);
int sq_dist_allreduce_u64(sq_dist*, uint64_t* host, size_t n);                  /* element-wise sums (mapping statistics, fragment counts) */
int sq_dist_bcast(sq_dist*, void* host, size_t bytes, int root);
int sq_dist_allgather(sq_dist*, const void* host_in, size_t bytes, void* host_out /* world * bytes */);
int sq_dist_barrier(sq_dist*);
/* This rank's contiguous share of `total` replicates, cut at multiples of `unit` (1 for bootstraps, sq_gibbs_chain_step(total) for Gibbs). */
void sq_dist_share(const sq_dist*, uint32_t total, uint32_t unit, uint32_t* first, uint32_t* count);

/* Per-stage HIP-event timing of the mapping / eq pipeline (measurement).  When enabled,
 * every stage kernel of sq_map_batch / sq_eq_accumulate is bracketed by hipEventRecord on the ctx
 * stream; sq_ctx_stage_times returns accumulated milliseconds and launch counts per stage. */
int sq_ctx_set_profiling(sq_ctx*, int on);
int sq_ctx_num_stages(void);
const char* sq_ctx_stage_name(int stage);
int sq_ctx_stage_times(sq_ctx*, double* ms /*[num_stages]*/, uint64_t* calls /*[num_stages]*/, int reset);
/* [r5] measurement: 64-byte filter sectors the seeding kernel fetched since the last reset (a probe whose minimizer's filter block is already in the
 * lane's LDS column fetches none) — with num_lookups and num_seeds of sq_map_stats the kernel's own algorithmic bytes. */
uint64_t sq_ctx_seed_filter_fills(sq_ctx*, int reset);

/* ------------------------------------------------------------------------------------------------
 * Debug / parity taps: copy an intermediate stage of the LAST sq_map_batch to the host.
 * ---------------------------------------------------------------------------------------------- */
enum { SQ_TAP_UNIMEMS = 1, SQ_TAP_MEMS = 2, SQ_TAP_CHAINS = 3, SQ_TAP_CANDIDATES = 4,
       SQ_TAP_PACKED = 5 /* the packed read ends as 64-bit words, per end: length | (has a non-base << 32), then the 2-bit words (32 bases each) and the non-base masks (64 bases each) of the context's packing stride; sq_debug_tap with buf = NULL returns the word count */ };
typedef struct { uint32_t end; uint16_t qpos, len; uint64_t unitig; uint32_t uoff; uint8_t fw; uint8_t _p[3]; } sq_unimem;
typedef struct { uint32_t end; uint32_t tid; int32_t rpos; uint16_t qpos, len; uint8_t fw; uint8_t _p[3]; } sq_mem;
typedef struct { uint32_t end; uint32_t tid; int32_t pos; int32_t last_end; uint8_t fw; uint8_t _p[3];
                 uint32_t n_mems; double score; } sq_chain;
typedef struct { uint32_t frag; uint32_t tid; int32_t lpos, rpos; uint8_t lfw, rfw, mate_status, valid;
                 int32_t lscore, rscore; uint32_t frag_len; } sq_cand;
/* Returns number of records (or negative status). buf may be NULL to query the count. */
int64_t sq_debug_tap(sq_ctx*, int what, void* buf, uint64_t cap_records);

/* Parity tap for orphan recovery's infix aligner (row a5; reference src/edlib.cpp:290-372 called with
 * {k, EDLIB_MODE_HW, EDLIB_TASK_LOC}): runs the DEVICE aligner on ncases (query, window) pairs.  Queries: ASCII, at most
 * 256 bases, any non-ACGT byte matches nothing; windows: ASCII ACGT only.  out[4i..4i+3] = found, edit distance,
 * startLocations[0], endLocations[0] (-1 when not found). */
int sq_debug_infix_align(int device, uint32_t ncases, const uint8_t* queries, const uint64_t* q_off /*[ncases+1]*/,
                         const uint8_t* windows, const uint64_t* w_off /*[ncases+1]*/, const int32_t* k /*[ncases]*/,
                         int32_t* out /*[4*ncases]*/);

/* [r5] test hooks of the device BGZF inflater (hip/inflate_dev.hip).  sq_debug_bgzf_inflate: members (raw deflate streams in `comp`, descriptors
 * {coff, voff, csize, isize, crc, 0} of 32 bytes each) through the device kernel; status2[0] = 0xFFFFFFFF or index + 1 of the first damaged member, status2[1] = why.
 * sq_debug_inflate_core_host: the decoder's source compiled for the CPU (tests check it against zlib where there is no GPU; 0 = sound). */
int sq_debug_bgzf_inflate(int device, const uint8_t* comp, uint64_t comp_bytes, const void* members, uint32_t nmem, uint8_t* text, uint64_t text_bytes, uint32_t* status2);
int sq_debug_inflate_core_host(const uint8_t* comp, uint64_t csize, uint8_t* out, uint32_t isize, uint32_t* crc_out);

#ifdef __cplusplus
}
#endif
#endif /* SALMON_HIP_H */
