// hip/ctx.h — per-device quantification context: HBM work buffers of the mapping pipeline, the
// online model and the equivalence-class table.  One sq_ctx per GPU (one process per GPU).
#pragma once
#include <cstdlib>
#include <cstddef>
#include <memory>
#include <thread>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <string>
#include <hip/hip_runtime.h>
#include <vector>
#include "device_index.h"

// [r4] a read end of up to SQ_MAX_READ_LEN bases is mapped whole (the 10-bit position / length fields of a MEM record hold 0..1023); a longer one is
// refused with an error, not cut.  The packed reads' stride is a property of the context: 8 words (256 bases) until a batch brings a longer read,
// then 16 or 32 (the batch is packed again; sticky).  N-mask words = stride / 2.
#define SQ_MAX_READ_LEN 1000u
#define SQ_READ_WORDS_MIN 8u      // 2-bit words per read end (the default stride: 256 bases)
#define SQ_READ_WORDS_MAX 32u
#define SQ_MAX_UNIMEMS 32u        // uni-MEM slots per read end by default, and the most a size class of k_mems keeps in LDS; [r4] an end that needs more makes the
                                  // context widen its slab (uni_slots: 64, 128, ... 1024) and takes the large-end path — nothing is dropped (SPEC §a1)
#define SQ_MAX_UNI_SLOTS 1024u
#define SQ_MAX_CHAIN_GAP 200      // SPEC §a2
#define SQ_REF_EXTEND 20          // aconf.refExtendLength (SalmonMappingUtils.hpp:184)
#define SQ_MAX_BAND 15            // DP band kept in registers (runtime bandwidth must be <= this)
#define SQ_INVALID_SCORE INT32_MIN
#define SQ_NEG_INF (-(1 << 29))

struct sq_unimem_dev {   // 32 B: the uni-MEM, and what its projection needs of the contig table and the unitig (k_seed has them at hand)
  uint32_t unitig, ustart; uint16_t qpos, len; uint8_t fw, pad[3];
  uint64_t ctab_a;      // start of the unitig's run in the contig table
  uint32_t cnt, ulen;   // occurrences (0: more than maxOccsPerHit, not projected); unitig length
};
static_assert(sizeof(sq_unimem_dev) == 32, "sq_unimem_dev layout");

struct sq_chain_dev {   // 40 B.  Bytes 16..31 are everything the scorer reads of a chain (its transcript is in the candidate): ONE 16-byte load
  double score; uint32_t tid; int32_t last_end;
  int32_t pos; uint32_t first; uint32_t pad2; uint16_t n_mems; uint8_t fw, pad[3];   // pad[0]: MEMs given by the bit mask pad2 (else linked through mnext)
  uint16_t read_len; uint32_t spare;
};
static_assert(sizeof(sq_chain_dev) == 40 && offsetof(sq_chain_dev, pos) == 16 && offsetof(sq_chain_dev, n_mems) == 28 && offsetof(sq_chain_dev, pad) == 31,
              "sq_chain_dev layout");
struct alignas(16) sq_cand_dev {    // 48 B
  double cov;
  uint32_t tid;
  uint32_t lc, rc;
  uint32_t frag_len;
  int32_t lscore, rscore;
  uint8_t mate_status, valid, compat, lfail, rfail, pad[3];
  uint32_t pad2;
};

struct sq_dp_item {     // one banded-DP region queued by the fast scorer (40 B)
  // region scores below `budget` cannot yield a valid alignment
  int64_t tstart;
  uint32_t cand;
  uint8_t end, mode, rc, pad;
  int32_t qstart, qdir, n;
  int32_t tdir, tl;
  int32_t budget;
};
static_assert(sizeof(sq_dp_item) == 40, "sq_dp_item layout");

struct sq_map_params {
  int32_t ma, mp, go, ge, bw;
  uint32_t k, alt_skip, max_occ, frag_len_max, first_decoy, max_read_occs;
  double pre_thr, post_thr, orphan_thr, consensus_frac, min_score_fraction, score_exp, decoy_threshold, min_aln_prob;
  uint8_t lib_type, lib_orient, lib_strand, hard_filter, allow_dovetail, allow_orphans, no_heuristic, ignore_incompat, recover_orphans;
};

template <class T>
struct sq_dbuf {
  T* p = nullptr; size_t n = 0;
  // grow-only with 25 % headroom: per-batch totals (MEMs, candidates) drift by a few percent, and a
  // hipFree + hipMalloc of a multi-GB buffer inside the hot loop costs milliseconds
  int ensure(size_t want) {
    if (want <= n && p) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
    size_t cap = want + want / 4 + 64;
    if (hipMalloc((void**)&p, cap * sizeof(T)) != hipSuccess) {
      cap = want ? want : 1;
      if (hipMalloc((void**)&p, cap * sizeof(T)) != hipSuccess) return -1;
    }
    n = cap;
    // SQ_POISON=1 (tests): fresh device memory is usually zero, memory handed back by another allocation or another process is not.
    // Filling every new buffer with a pattern makes any read-before-write show up as a parity failure instead of hiding behind zeros.
    static const bool poison = getenv("SQ_POISON") != nullptr;
    if (poison) (void)hipMemset(p, 0xA5, cap * sizeof(T));
    return 0;
  }
  void free_() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

struct sq_online_dev;  // online.hip
struct sq_eq_dev;      // eq.hip

struct sq_ctx {
  sq_index* idx = nullptr; sq_device_index* di = nullptr; int device = 0;
  sq_quant_opts opts; sq_map_params mp; uint32_t max_reads = 0;
  uint32_t uni_slots = SQ_MAX_UNIMEMS;       // stride of the uni-MEM slab; raised (and the batch seeded again) when an end produces more
  uint32_t seed_lw = 4;                      // [r5] words of a read end k_seed2 keeps in LDS: 4 (reads of up to 128 bases) until a batch brings a longer one, then [r6] 5 (up to 160: 2 x 150 keeps six blocks per CU) or 8 (sticky)
  uint32_t read_words = SQ_READ_WORDS_MIN;   // stride of rpack (rnmask: half of it); raised when a batch holds reads of more than 32 * read_words bases
  hipStream_t stream = nullptr;
  // reads
  sq_dbuf<uint8_t> seq; sq_dbuf<uint64_t> seq_off; sq_dbuf<uint64_t> rpack; sq_dbuf<uint64_t> rnmask; sq_dbuf<uint16_t> rlen; sq_dbuf<uint8_t> rany;   // rany [r6]: one byte per end, "its N-mask has a bit set" (k_seed2 asks this instead of reading the mask)
  // seeds / MEMs
  sq_dbuf<sq_unimem_dev> unimems; sq_dbuf<uint32_t> n_uni; sq_dbuf<uint32_t> n_proj; sq_dbuf<uint64_t> mem_off;
  sq_dbuf<uint64_t> mkey, mval, mkey2, mval2; sq_dbuf<uint8_t> sort_tmp; uint64_t mem_cap = 0;
  sq_dbuf<uint4> mlinfo;   // list entries of the MEM size classes (mem_kernels.h)
  sq_dbuf<uint32_t> dp_bh, dp_perm; sq_dbuf<uint64_t> dp_off;   // DP queue order (k_dp_hist / k_dp_scatter)
  sq_dbuf<double> cf; sq_dbuf<int32_t> cp; sq_dbuf<uint32_t> mnext; sq_dbuf<uint8_t> mused;
  sq_dbuf<uint64_t> lg_a, lg_b; sq_dbuf<uint32_t> lg_c, lg_d, lg_first, lg_cnt; sq_dbuf<uint8_t> lg_flags;   // [r4] flat chaining of the large class (mem_kernels.h: k_lg_*)
  sq_dbuf<uint32_t> mlist, mlbase; sq_dbuf<uint64_t> lkey, lval;   // read ends by MEM-count class (mem_kernels.h); sorted compact buffer of the large class
  // chains
  sq_dbuf<sq_chain_dev> chains; sq_dbuf<uint32_t> n_chains; uint64_t last_total_chains = 0;
  // candidates / alignments
  sq_dbuf<uint32_t> n_cand; sq_dbuf<uint64_t> cand_off; sq_dbuf<sq_cand_dev> cands; uint64_t cand_cap = 0;
  sq_dbuf<uint32_t> cand_frag, tid_arr; sq_dbuf<int32_t> hs_arr;
  sq_dbuf<sq_dp_item> dpq; sq_dbuf<uint32_t> counters; sq_dbuf<uint8_t> frag_flags;
  sq_dbuf<uint32_t> n_aln; sq_dbuf<uint64_t> aln_off; sq_dbuf<sq_aln> aln_slots; sq_dbuf<uint64_t> sel_desc /* [r5] k_select: ticket + one look-back descriptor per block */; sq_dbuf<sq_aln> aln; sq_dbuf<uint8_t> map_type;
  sq_dbuf<double> gapcost; sq_dbuf<unsigned long long> stats;
  // last batch bookkeeping
  uint32_t last_n = 0;
  uint32_t last_paired = 0;
  uint64_t seed_fills = 0;   // [r5] sum of ST_FILLS over this lane's batches (sq_ctx_seed_filter_fills)
  uint64_t last_total_aln = 0, last_total_mems = 0, last_total_cands = 0, last_joint = 0, last_chain_slots = 0;
  void* em_arena = nullptr;   // persistent EM workspace (em.hip: EmArena), grown by sq_ctx_reserve / sq_em_optimize(ctx, ...)
  bool have_batch = false;
  // online model + eq table
  sq_online_dev* online = nullptr; sq_eq_dev* eq = nullptr;
  uint64_t reads_seen = 0;
  // eq stage runs on its own stream so the online model of batch b overlaps the mapping of batch b+1;
  // alignments are double-buffered (alnb[2], aln_offb[2]) and handed over with events
  // With a CU partition (eq_cus > 0) stream/stream2 are CU-masked to disjoint sets: the eq stage's chain of small
  // dependent kernels then never queues behind the mapping kernels' workgroups.  stream3 is unmasked: an eq job that
  // starts while no mapping is in flight (the last batch of a run) takes the whole GPU instead.
  hipStream_t stream3 = nullptr;
  hipStream_t eq_stream_cur = nullptr;
  int eq_cus = 0, ncu = 0;
  std::atomic<int> map_active{0};
  hipEvent_t ev_eq_last = nullptr;
  hipStream_t stream2 = nullptr;
  std::vector<hipEvent_t> prof_ev3; std::vector<int> prof_stage3;
  hipEvent_t ev_map_done[2] = {nullptr, nullptr}, ev_eq_done[2] = {nullptr, nullptr};
  int cur_buf = 0, last_buf = 0;
  bool eq_pending[2] = {false, false};
  sq_dbuf<sq_aln> aln_b1; sq_dbuf<uint64_t> aln_off_b1;
  // [r5] the reads behind an injected batch (sq_aln_inject_reads), one set per alignment buffer: CIGARs, bases, positions, aligner scores
  sq_dbuf<uint64_t> rd_cig_off[2], rd_seq_off[2]; sq_dbuf<uint32_t> rd_cig[2]; sq_dbuf<uint8_t> rd_seq[2]; sq_dbuf<int32_t> rd_pos[2], rd_score[2]; bool rd_have[2] = {false, false};
  sq_aln* aln_ptr(int b) { return b ? aln_b1.p : aln.p; }
  uint64_t* aln_off_ptr(int b) { return b ? aln_off_b1.p : aln_off.p; }
  std::vector<hipEvent_t> prof_ev2; std::vector<int> prof_stage2;
  // The eq stage of a batch is several hundred small launches (three per mini-batch of 5000 fragments):
  // a worker thread enqueues them on stream2 so the caller can go straight on to mapping the next batch.
  struct eq_job { uint32_t n; int buf; uint64_t total_aln, joint; sq_ctx* src; };
  std::thread eq_thread; std::mutex eq_mu; std::condition_variable eq_cv, eq_cv_done; std::deque<eq_job> eq_q;
  uint64_t eq_submitted = 0, eq_enqueued = 0;
  uint64_t eq_job_of_buf[2] = {0, 0};
  bool eq_stop = false;
  int eq_err = 0;
  std::string eq_errmsg;
  // Mapping lanes: sq_map_submit / sq_map_wait run batches on alternating lanes, each a worker thread with its own
  // stream and work buffers (lane 0 = this ctx, further lanes = shadow ctxs that own buffers only).  The mapping
  // kernels are latency- or issue-bound one at a time; two batches in flight fill each other's stalls.  Results come
  // back in submission order; the online/eq stage stays strictly ordered on its own stream.
  struct map_job { sq_read_batch in; sq_aln_batch out; bool has_out = false; int rc = 0; sq_map_stats st; bool done = false; std::string err; uint32_t n = 0; int buf = 0; uint64_t total_aln = 0,
      joint = 0; };
  // the batch sq_eq_accumulate will take (set by sq_map_batch / sq_map_wait)
  uint32_t acc_n = 0;
  int acc_buf = 0;
  uint64_t acc_total_aln = 0, acc_joint = 0;
  sq_ctx* owner = nullptr;                 // set in a shadow ctx: the ctx that owns the online model and the eq worker
  std::vector<sq_ctx*> shadows;            // lanes 1.. (owned by lane 0)
  int n_lanes = 0;                         // 0 = not chosen yet (SQ_MAP_LANES or 2 at the first submit)
  sq_ctx* last_src = nullptr;              // lane whose batch the next sq_eq_accumulate / sq_debug_tap refers to
  bool api_have = false;                   // a mapped batch has been returned to the caller and not yet accumulated
  std::thread lane_thread;
  std::mutex lane_mu;
  std::condition_variable lane_cv, lane_cv_done;
  std::deque<std::shared_ptr<map_job>> lane_q;
  bool lane_stop = false;
  std::deque<std::pair<sq_ctx*, std::shared_ptr<map_job>>> tickets; uint64_t submitted = 0;
  // stage profiling
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;
  std::vector<int> prof_stage;
  double stage_ms[32] = {0};
  uint64_t stage_calls[32] = {0};
  uint64_t eq_groups = 0;   // groups of mini-batches run while profiling (launch pairs of the online chain)
};

enum { SG_PACK = 0, SG_SEED, SG_SCAN_MEMS, SG_PROJECT, SG_SORT, SG_CHAIN, SG_JOIN_COUNT, SG_SCAN_CANDS, SG_JOIN_FILL, SG_SCORE, SG_DP,
    SG_SELECT, SG_COMPACT,
       SG_EQ_FLAGS, SG_EQ_MINIBATCH, SG_EQ_TABLE, SG_FINALIZE, SG_EQ_STATIC, SG_NUM };
void sq_prof_mark(sq_ctx* c, int stage, int which = 0);   // records an event: time since the previous mark is charged to `stage` (which: 0 map stream, 1 eq stream, 2 the chain's stream when the eq stage is split)
void sq_prof_begin(sq_ctx* c, int which = 0);
void sq_prof_end(sq_ctx* c, int which = 0);               // call after the stream has been synchronised
int sq_eq_sync(sq_ctx* c);
void sq_eq_wait_enqueued(sq_ctx* c, uint64_t id);        // block until the eq worker has enqueued job `id`
void sq_eq_worker_stop(sq_ctx* c);                                // wait for outstanding eq-stage work, collect its timings, report table overflow

// stats slots (device array of unsigned long long, same order as sq_map_stats)
enum { ST_READS = 0, ST_KMER, ST_JOINT, ST_MAPPED, ST_ALNS, ST_MAPFILT, ST_FRAGFILT, ST_DOVETAIL, ST_DECOY, ST_SEEDS, ST_LOOKUPS, ST_MEMS,
    ST_CHAINS, ST_CANDS, ST_DP,
    ST_RESCUED, ST_TRUNC, ST_MAXLEN /* longest read end of the batch when it did not fit the packing stride (k_pack) */,
    ST_UNIOVER /* read ends whose uni-MEMs did not fit the slab's stride (k_seed) */,
    ST_SEEDLW /* [r5] the longest read end that did not fit the LDS column of the k_seed2 instantiation that ran, 0 if all did (the host seeds again with a wider one) */,
    ST_FILLS /* [r5] filter blocks k_seed2 brought into LDS (+ the rare single words read past them): the filter's sectors per launch */, ST_N };

int sq_eq_export_dev(sq_ctx* c, sq_eq_dev_csr* out);     // runs the export if needed; pointers stay valid until the next accumulate / merge / reset
int sq_map_batch_impl(sq_ctx* c, const sq_read_batch* in, sq_aln_batch* out, sq_map_stats* stats);   // runs one batch on lane ctx `c`
extern "C" int sq_merge_log_masses(uint32_t M, uint32_t R, const double* all_log_mass, double* out);   // host/opts.cpp
void sq_detect_lib_format(uint8_t type, const uint64_t* counts64, uint8_t* out_type, uint8_t* out_orient, uint8_t* out_strand);   // host/opts.cpp
int sq_online_create(sq_ctx* c);
int sq_em_optimize_bias_impl(int device, const sq_eq_table* eq, const sq_eq_dev_csr* dv, const sq_txp_in* txp, const sq_em_opts* o, sq_efflen_cb cb, void* user,
    double* alpha_out, double* eff_len_out, sq_em_report* rep, void** arena_slot, void* lent_stream);   // em.hip
void sq_online_free(sq_ctx* c);
