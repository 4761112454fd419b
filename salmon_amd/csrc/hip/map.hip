// hip/map.hip — host orchestration of the mapping pipeline (seam B1): sq_ctx_create / sq_map_batch /
// sq_debug_tap.  Stage kernels live in map_kernels.h.  All intermediate data stay in HBM between
// stages; the host only reads back three totals (MEMs, candidates, DP regions) to size buffers.
#include "map_kernels.h"
#include "mem_kernels.h"
#include "scan_kernels.h"
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cstring>

using namespace sqk;

namespace {
const int TB = 256;
inline uint32_t nblk(uint64_t n) { return (uint32_t)((n + TB - 1) / TB); }

int exclusive_scan_u32(sq_ctx* c, const uint32_t* in, uint64_t* out, uint32_t n_plus_1) {   // scan_kernels.h
  const uint64_t n = (uint64_t)n_plus_1 - 1;
  if (c->sort_tmp.ensure((size_t)scan_tiles(n) * 8 + 256)) { sq_set_error("scan spine allocation failed"); return SQ_ERR_NOMEM; }
  exclusive_scan_u32_u64(in, out, n, (uint64_t*)c->sort_tmp.p, c->stream);
  SQ_HIP_CHECK(hipGetLastError());
  return SQ_OK;
}

void fill_params(sq_ctx* c) {
  const sq_quant_opts& o = c->opts; sq_map_params& P = c->mp;
  P.ma = o.match_score; P.mp = o.mismatch_penalty; P.go = o.gap_open; P.ge = o.gap_extend; P.bw = o.bandwidth;
  P.k = c->idx->k;
  P.alt_skip = o.mismatch_seed_skip;
  P.max_occ = o.max_occs_per_hit;
  P.frag_len_max = o.frag_len_max;
  P.first_decoy = c->idx->first_decoy;
  P.pre_thr = o.pre_merge_chain_sub_thresh; P.post_thr = o.post_merge_chain_sub_thresh; P.orphan_thr = o.orphan_chain_sub_thresh;
  P.consensus_frac = (o.consensus_slack == 0.0) ? 1.0 : (1.0 - o.consensus_slack);  // SalmonMappingUtils.hpp:160-162
  P.min_score_fraction = o.min_score_fraction;
  P.score_exp = o.score_exp;
  P.decoy_threshold = o.decoy_threshold;
  P.min_aln_prob = o.min_aln_prob;
  P.lib_type = o.lib_type;
  P.lib_orient = o.lib_orientation;
  P.lib_strand = o.lib_strand;
  P.hard_filter = o.hard_filter;
  P.allow_dovetail = o.allow_dovetail;
  P.allow_orphans = o.allow_orphans; P.no_heuristic = o.disable_chaining_heuristic; P.ignore_incompat = o.ignore_incompat;
  P.recover_orphans = o.recover_orphans; P.max_read_occs = o.max_read_occs;
}
}  // namespace

static const char* kStageNames[SG_NUM] = {"k_pack", "k_seed", "scan_mems", "k_mems", "large_ends", "count_kmer_frags", "k_join_count",
    "scan_cands", "k_join_fill", "k_score",
    "k_dp", "k_select", "compact_alns",
                                          "eq_flags_scan", "eq_mini_batches", "eq_table", "k_finalize", "eq_static"};
static hipStream_t prof_stream(sq_ctx* c, int which) { return which == 0 ? c->stream : which == 2 ? c->stream2 : (c->eq_stream_cur ? c->eq_stream_cur : c->stream2); }
static std::vector<hipEvent_t>& prof_evs(sq_ctx* c, int which) { return which == 0 ? c->prof_ev : which == 2 ? c->prof_ev3 : c->prof_ev2; }
static std::vector<int>& prof_stgs(sq_ctx* c, int which) { return which == 0 ? c->prof_stage : which == 2 ? c->prof_stage3 : c->prof_stage2; }
void sq_prof_begin(sq_ctx* c, int which) {
  if (!c->prof_on) return; auto& ev = prof_evs(c, which); auto& stg = prof_stgs(c, which); hipStream_t st = prof_stream(c, which);
  if (!(which && !stg.empty())) stg.clear();   // eq stages not collected yet keep their marks; a new origin event separates the stages
  size_t i = stg.size(); if (ev.size() <= i) { hipEvent_t e; hipEventCreate(&e); ev.push_back(e); } hipEventRecord(ev[i], st); stg.push_back(-1); }
void sq_prof_mark(sq_ctx* c, int stage, int which) {
  if (!c->prof_on) return; auto& ev = prof_evs(c, which); auto& stg = prof_stgs(c, which); hipStream_t st = prof_stream(c, which);
  size_t i = stg.size(); if (ev.size() <= i) { hipEvent_t e; hipEventCreate(&e); ev.push_back(e); } hipEventRecord(ev[i], st); stg.push_back(stage); }
void sq_prof_end(sq_ctx* c, int which) {
  if (!c->prof_on) return;
  for (int w = which; w <= (which == 1 ? 2 : which); ++w) {   // the eq stage's two sequences end together
    auto& ev = prof_evs(c, w); auto& stg = prof_stgs(c, w);
    for (size_t i = 1; i < stg.size(); ++i) { if (stg[i] < 0) continue; float ms = 0; if (hipEventElapsedTime(&ms, ev[i - 1], ev[i]) == hipSuccess) { c->stage_ms[stg[i]] += ms; c->stage_calls[stg[i]]++; } }
    stg.clear(); } }
extern "C" int sq_ctx_set_profiling(sq_ctx* c, int on) {
  if (!c) return SQ_ERR_ARG;
  c->prof_on = on != 0;
  for (sq_ctx* sh : c->shadows) sh->prof_on = c->prof_on;
  return SQ_OK;
}
extern "C" int sq_ctx_num_stages(void) { return SG_NUM; }
extern "C" const char* sq_ctx_stage_name(int s) { return (s >= 0 && s < SG_NUM) ? kStageNames[s] : nullptr; }
extern "C" uint64_t sq_ctx_seed_filter_fills(sq_ctx* c, int reset) {   // [r5] filter sectors k_seed2 fetched since the last reset (all lanes)
  if (!c) return 0;
  uint64_t v = c->seed_fills; if (reset) c->seed_fills = 0;
  for (sq_ctx* sh : c->shadows) { v += sh->seed_fills; if (reset) sh->seed_fills = 0; }
  return v;
}
extern "C" int sq_ctx_stage_times(sq_ctx* c, double* ms, uint64_t* calls, int reset) {
  if (!c) return SQ_ERR_ARG;
  (void)sq_eq_sync(c);
  for (int i = 0; i < SG_NUM; ++i) {
    double m = c->stage_ms[i]; uint64_t n = c->stage_calls[i];
    for (sq_ctx* sh : c->shadows) {
      m += sh->stage_ms[i];
      n += sh->stage_calls[i];
      if (reset) {
        sh->stage_ms[i] = 0;
        sh->stage_calls[i] = 0;
      }
    }
    if (ms) ms[i] = m; if (calls) calls[i] = n; if (reset) { c->stage_ms[i] = 0; c->stage_calls[i] = 0; } }
  // the online chain's row counts its launch pairs (one per group of mini-batches), not mapped batches
  if (calls && c->eq_groups) calls[SG_EQ_MINIBATCH] = c->eq_groups;
  if (reset) c->eq_groups = 0;
  return SQ_OK;
}

static int ctx_create_lane(sq_index* idx, const sq_quant_opts* opts, int device, uint32_t max_batch_reads, sq_ctx* owner, sq_ctx** out);
// [r6] The "late first synchronisation" of rounds 3-5, explained.  For a copy between the device and PAGEABLE host memory above a size threshold the HIP runtime pins the caller's
// pages for the transfer (a userptr buffer object) and keeps the pin cached; when the caller later frees or remaps that memory the kernel's MMU notifier fires and amdkfd EVICTS
// THE PROCESS'S QUEUES until it has rebuilt the mapping — the next submission, whatever it is, then starts 6-30 ms late (measured: the first 1.5 MB upload of the EM set-up began
// 15.9 ms after it was queued, the device idle, `evicted_ms` of the process counting up; profiles/r06_eviction_ab.txt: 11-15 ms with the default, 0.04 ms with the threshold beyond
// any copy, three times each in one session).  GPU_PINNED_MIN_XFER_SIZE (MiB) is that threshold: beyond any copy, pageable transfers go through the runtime's own staging buffers
// and nothing of the caller's memory is ever pinned behind its back.  The runtime reads it when it initialises, so it is set when this library is loaded — unless the host
// application has set it itself; an application that initialises HIP before loading the library sets it in its own environment (bench.py and salmon-hip do).
__attribute__((constructor)) static void sq_runtime_defaults() { setenv("GPU_PINNED_MIN_XFER_SIZE", "1048576", 0); }

extern "C" int sq_ctx_create(sq_index* idx, const sq_quant_opts* opts, int device, uint32_t max_batch_reads, sq_ctx** out) {
  return ctx_create_lane(idx, opts, device, max_batch_reads, nullptr, out);
}
// owner == nullptr: a full context (lane 0); else a shadow lane: work buffers and a stream only
static int ctx_create_lane(sq_index* idx, const sq_quant_opts* opts, int device, uint32_t max_batch_reads, sq_ctx* owner, sq_ctx** out) {
  if (!idx || !opts || !out || max_batch_reads == 0) { sq_set_error("sq_ctx_create: bad arguments"); return SQ_ERR_ARG; }
  if (max_batch_reads > (1u << 23)) {
    sq_set_error("max_batch_reads %u exceeds 2^23 (sort key layout)", max_batch_reads);
    return SQ_ERR_ARG;
  }
  if (opts->bandwidth > SQ_MAX_BAND || opts->bandwidth < 0) {
    sq_set_error("bandwidth %d not supported (max %d)", opts->bandwidth, SQ_MAX_BAND);
    return SQ_ERR_ARG;
  }
  int rc = sq_index_to_device(idx, device); if (rc) return rc;
  SQ_HIP_CHECK(hipSetDevice(device));
  sq_ctx* c = new sq_ctx();
  c->idx = idx;
  c->di = idx->dev;
  c->device = device;
  c->opts = *opts;
  c->max_reads = max_batch_reads;
  c->owner = owner;
  c->last_src = c;
  fill_params(c);
  { // CU partition: the eq stage gets `eq_cus` CUs of its own (default 64 of 256; SQ_EQ_CUS=0 disables) and the mapping stream keeps
    // off them, so the eq chain never waits behind mapping workgroups.  The mask is a multiple of 64 bits from the top: workgroups are dealt
    // round-robin to the XCDs, and an XCD with fewer CUs than the others still gets a full share of the blocks and sets the pace
    // (measured [r2], 4·10^6-pair batches: 32 / 40 / 48 / 64 CUs -> eq chain 44 / 44 / 44 / 28 ms; [r3] mask bit i is a CU of XCD i mod 8,
    // so 64 bits = 8 CUs in every XCD — see the note at the loop below).
    // With the mapping kernels on 192 CUs, the 64 keep the eq chain (22 ms per 8·10^6 pairs) off the critical path (24.7 ms).
    hipDeviceProp_t prop; SQ_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    const int ncu = prop.multiProcessorCount;
    int eq_cus = getenv("SQ_EQ_CUS") ? atoi(getenv("SQ_EQ_CUS")) : (ncu >= 128 ? ncu / 4 : 0);
    if (eq_cus < 0 || eq_cus >= ncu) eq_cus = 0;
    c->eq_cus = eq_cus; c->ncu = ncu;
    if (eq_cus > 0) {
      std::vector<uint32_t> m1((ncu + 31) / 32, 0), m2((ncu + 31) / 32, 0);
      // [r3] the driver deals the mask's bits round-robin to the XCDs (bit i -> XCD i mod 8: amdkfd's symmetric CU-mask mapping), so the top
      // eq_cus bits are eq_cus / 8 CUs in EVERY XCD (default: 8 of 32) — not whole XCDs, as the round-2 comment above assumed.
      for (int i = 0; i < ncu; ++i) { if (i >= ncu - eq_cus) m2[i / 32] |= 1u << (i % 32); else m1[i / 32] |= 1u << (i % 32); }
      // a partition is an optimisation: if the platform refuses CU masks, fall back to plain streams
      bool ok = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)m1.size(), m1.data()) == hipSuccess;
      if (ok && !owner) ok = hipExtStreamCreateWithCUMask(&c->stream2, (uint32_t)m2.size(), m2.data()) == hipSuccess &&
          hipStreamCreate(&c->stream3) == hipSuccess;
      if (!ok) {
        (void)hipGetLastError();
        if (c->stream) {
          (void)hipStreamDestroy(c->stream);
          c->stream = nullptr;
        }
        if (c->stream2) {
          (void)hipStreamDestroy(c->stream2);
          c->stream2 = nullptr;
        }
        if (c->stream3) {
          (void)hipStreamDestroy(c->stream3);
          c->stream3 = nullptr;
        }
        c->eq_cus = 0;
        SQ_HIP_CHECK(hipStreamCreate(&c->stream)); if (!owner) SQ_HIP_CHECK(hipStreamCreate(&c->stream2));
      }
    } else {
      SQ_HIP_CHECK(hipStreamCreate(&c->stream)); if (!owner) SQ_HIP_CHECK(hipStreamCreate(&c->stream2));
    }
    c->eq_stream_cur = c->stream2;
  }
  for (int b = 0; b < 2; ++b) {
    SQ_HIP_CHECK(hipEventCreateWithFlags(&c->ev_map_done[b], hipEventDisableTiming));
    SQ_HIP_CHECK(hipEventCreateWithFlags(&c->ev_eq_done[b], hipEventDisableTiming));
  }
  const uint32_t nends = 2 * max_batch_reads;
  bool bad = c->seq_off.ensure((size_t)nends + 2) || c->rpack.ensure((size_t)nends * c->read_words + 8) ||
      c->rnmask.ensure((size_t)nends * (c->read_words / 2) + 8) ||
      c->rlen.ensure(nends) || c->rany.ensure(nends) ||
             c->unimems.ensure((size_t)nends * c->uni_slots) || c->n_uni.ensure(nends + 1) || c->n_proj.ensure(nends + 1) ||
                 c->mem_off.ensure((size_t)nends + 2) ||
             c->n_chains.ensure(nends + 1) || c->n_cand.ensure(max_batch_reads + 1) ||
                 c->cand_off.ensure((size_t)max_batch_reads + 2) || c->counters.ensure(32) ||
             c->frag_flags.ensure(max_batch_reads) || c->n_aln.ensure(max_batch_reads + 1) ||
                 c->aln_off.ensure((size_t)max_batch_reads + 2) ||
                 c->aln_off_b1.ensure((size_t)max_batch_reads + 2) || c->map_type.ensure(max_batch_reads) ||
             c->stats.ensure(ST_N) || c->gapcost.ensure(SQ_MAX_CHAIN_GAP + 1);
  if (bad) { sq_set_error("device allocation failed in sq_ctx_create"); sq_ctx_free(c); return SQ_ERR_NOMEM; }
  // chaining gap-cost table: 0.01*avgSeed*l + 0.5*log2(l) (SPEC §a2), built with the shared deterministic log
  std::vector<double> gc(SQ_MAX_CHAIN_GAP + 1, 0.0); const double inv_ln2 = 1.0 / 0.6931471805599453;
  for (int l = 1; l <= SQ_MAX_CHAIN_GAP; ++l) gc[l] = 0.01 * 31.0 * (double)l + 0.5 * (sq_log((double)l) * inv_ln2);
  SQ_HIP_CHECK(hipMemcpy(c->gapcost.p, gc.data(), gc.size() * 8, hipMemcpyHostToDevice));
  if (!owner) { rc = sq_online_create(c); if (rc) { sq_ctx_free(c); return rc; } }
  *out = c;
  return SQ_OK;
}

extern "C" void sq_ctx_free(sq_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->lane_thread.joinable()) {
    {
      std::lock_guard<std::mutex> lk(c->lane_mu);
      c->lane_stop = true;
    }
    c->lane_cv.notify_all();
    c->lane_thread.join();
  }
  if (!c->owner) {
    for (sq_ctx* sh : c->shadows) if (sh->lane_thread.joinable()) {
      {
        std::lock_guard<std::mutex> lk(sh->lane_mu);
        sh->lane_stop = true;
      }
      sh->lane_cv.notify_all();
      sh->lane_thread.join();
    }
  }
  if (!c->owner) sq_eq_worker_stop(c);   // drains the queued eq-stage jobs first
  if (!c->owner) { for (sq_ctx* sh : c->shadows) sq_ctx_free(sh); c->shadows.clear(); }
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  if (c->stream3) (void)hipStreamSynchronize(c->stream3);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (!c->owner) sq_online_free(c);
  if (c->em_arena) { sq_em_arena_free(c->em_arena); c->em_arena = nullptr; }
  c->seq.free_();
  c->seq_off.free_();
  c->rpack.free_();
  c->rnmask.free_();
  c->rlen.free_(); c->rany.free_();
  c->unimems.free_();
  c->n_uni.free_();
  c->n_proj.free_();
  c->mem_off.free_();
  c->mkey.free_();
  c->mval.free_();
  c->mkey2.free_();
  c->mval2.free_();
  c->mlinfo.free_(); c->sort_tmp.free_(); c->dp_bh.free_(); c->dp_perm.free_(); c->dp_off.free_();
  c->cf.free_();
  c->cp.free_();
  c->mnext.free_();
  c->mused.free_();
  c->mlist.free_();
  c->mlbase.free_();
  c->lkey.free_(); c->lg_a.free_(); c->lg_b.free_(); c->lg_c.free_(); c->lg_d.free_(); c->lg_first.free_(); c->lg_cnt.free_(); c->lg_flags.free_();
  c->lval.free_();
  c->chains.free_();
  c->n_chains.free_();
  c->n_cand.free_();
  c->cand_off.free_();
  c->cands.free_();
  c->cand_frag.free_();
  c->hs_arr.free_();
  c->tid_arr.free_();
  c->dpq.free_();
  c->counters.free_();
  c->frag_flags.free_();
  c->n_aln.free_();
  c->aln_off.free_();
  c->aln_slots.free_(); c->sel_desc.free_();
  c->aln.free_();
  c->aln_b1.free_();
  c->aln_off_b1.free_();
  c->map_type.free_(); c->gapcost.free_(); c->stats.free_();
  if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
  if (c->stream3) { (void)hipStreamSynchronize(c->stream3); (void)hipStreamDestroy(c->stream3); }

  for (int b = 0; b < 2; ++b) {
    if (c->ev_map_done[b]) (void)hipEventDestroy(c->ev_map_done[b]);
    if (c->ev_eq_done[b]) (void)hipEventDestroy(c->ev_eq_done[b]);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int sq_map_batch(sq_ctx* c, const sq_read_batch* in, sq_aln_batch* out, sq_map_stats* stats) {
  if (!c || c->owner) { sq_set_error("sq_map_batch: bad context"); return SQ_ERR_ARG; }
  if (!c->tickets.empty()) {
    sq_set_error("sq_map_batch: %zu submitted batches are still outstanding (sq_map_wait them first)", c->tickets.size());
    return SQ_ERR_STATE;
  }
  int rc = sq_map_batch_impl(c, in, out, stats);
  c->last_src = c; c->api_have = (rc == SQ_OK);
  c->acc_n = c->last_n; c->acc_buf = c->last_buf; c->acc_total_aln = c->last_total_aln; c->acc_joint = c->last_joint;
  return rc;
}

// [r4] alignment-based mode (`salmon quant -a`, src/alignment/SalmonQuantifyAlignments.cpp:125-937): the alignments come from a SAM file instead of the
// mapping kernels.  A batch of them is put where sq_map_batch would have left its own — the lane's alignment buffer, offsets by fragment —
// and sq_eq_accumulate then runs the same online model / equivalence-class stage on it.
static int aln_inject_impl(sq_ctx* c, const sq_aln_batch* in, const sq_aln_reads* reads, uint64_t num_with_joint_hits);
extern "C" int sq_aln_inject(sq_ctx* c, const sq_aln_batch* in, uint64_t num_with_joint_hits) { return aln_inject_impl(c, in, nullptr, num_with_joint_hits); }
// [r5] the same with the reads behind the alignments (CIGARs, bases, positions): what the CIGAR-based error model scores and learns from (hip/online.hip)
extern "C" int sq_aln_inject_reads(sq_ctx* c, const sq_aln_batch* in, const sq_aln_reads* reads, uint64_t num_with_joint_hits) {
  if (!reads) { sq_set_error("sq_aln_inject_reads: bad arguments"); return SQ_ERR_ARG; }
  return aln_inject_impl(c, in, reads, num_with_joint_hits);
}
static int aln_inject_impl(sq_ctx* c, const sq_aln_batch* in, const sq_aln_reads* reads, uint64_t num_with_joint_hits) {
  if (!c || c->owner || !in || !in->read_off || (!in->aln && in->n && in->read_off[in->n])) { sq_set_error("sq_aln_inject: bad arguments"); return SQ_ERR_ARG; }
  if (!c->tickets.empty()) { sq_set_error("sq_aln_inject: submitted batches are still outstanding"); return SQ_ERR_STATE; }
  const uint32_t n = in->n; if (n > c->max_reads) { sq_set_error("batch of %u fragments exceeds ctx capacity %u", n, c->max_reads); return SQ_ERR_ARG; }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  hipStream_t st = c->stream; const int buf = c->cur_buf;
  const uint64_t total = n ? in->read_off[n] : 0; const uint32_t M = (uint32_t)c->idx->names.size();
  for (uint32_t f = 0; f < n; ++f) if (in->read_off[f] > in->read_off[f + 1] || in->read_off[f + 1] > total) { sq_set_error("sq_aln_inject: read_off is not a prefix sum"); return SQ_ERR_ARG; }
  for (uint64_t a = 0; a < total; ++a) if (in->aln[a].tid >= M) { sq_set_error("sq_aln_inject: alignment %llu names transcript %u of %u", (unsigned long long)a, in->aln[a].tid, M); return SQ_ERR_ARG; }
  const size_t CP = (size_t)total + 8;
  if (c->eq_pending[buf]) {   // the eq stage that read this buffer two batches ago (as in sq_map_batch)
    sq_eq_wait_enqueued(c, c->eq_job_of_buf[buf]);
    if ((buf ? c->aln_b1.n : c->aln.n) < CP) SQ_HIP_CHECK(hipEventSynchronize(c->ev_eq_done[buf]));
    SQ_HIP_CHECK(hipStreamWaitEvent(st, c->ev_eq_done[buf], 0)); c->eq_pending[buf] = false;
  }
  if ((buf ? c->aln_b1.ensure(CP) : c->aln.ensure(CP))) { sq_set_error("device allocation failed (injected alignments)"); return SQ_ERR_NOMEM; }
  SQ_HIP_CHECK(hipMemcpyAsync(c->aln_off_ptr(buf), in->read_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
  if (total) SQ_HIP_CHECK(hipMemcpyAsync(c->aln_ptr(buf), in->aln, (size_t)total * sizeof(sq_aln), hipMemcpyHostToDevice, st));
  c->rd_have[buf] = false;
  if (reads) {
    if (reads->num_alignments != total || (total && (!reads->cig_off || !reads->seq_off || !reads->pos || !reads->aligner_score))) { sq_set_error("sq_aln_inject_reads: the reads do not belong to this batch (%llu alignments, reads of %llu)", (unsigned long long)total, (unsigned long long)reads->num_alignments); return SQ_ERR_ARG; }
    const uint64_t nc = total ? reads->cig_off[2 * total] : 0, ns = total ? reads->seq_off[2 * total] : 0;
    for (uint64_t j = 0; j < 2 * total; ++j) if (reads->cig_off[j] > reads->cig_off[j + 1] || reads->seq_off[j] > reads->seq_off[j + 1]) { sq_set_error("sq_aln_inject_reads: offsets are not prefix sums"); return SQ_ERR_ARG; }
    if (c->rd_cig_off[buf].ensure(2 * total + 2) || c->rd_seq_off[buf].ensure(2 * total + 2) || c->rd_cig[buf].ensure(nc + 8) || c->rd_seq[buf].ensure(ns + 8) || c->rd_pos[buf].ensure(2 * total + 2) || c->rd_score[buf].ensure(total + 2)) {
      sq_set_error("device allocation failed (injected reads)"); return SQ_ERR_NOMEM; }
    if (total) {
      SQ_HIP_CHECK(hipMemcpyAsync(c->rd_cig_off[buf].p, reads->cig_off, (2 * total + 1) * 8, hipMemcpyHostToDevice, st)); SQ_HIP_CHECK(hipMemcpyAsync(c->rd_seq_off[buf].p, reads->seq_off, (2 * total + 1) * 8, hipMemcpyHostToDevice, st));
      if (nc) SQ_HIP_CHECK(hipMemcpyAsync(c->rd_cig[buf].p, reads->cigar, nc * 4, hipMemcpyHostToDevice, st));
      if (ns) SQ_HIP_CHECK(hipMemcpyAsync(c->rd_seq[buf].p, reads->seq, ns, hipMemcpyHostToDevice, st));
      SQ_HIP_CHECK(hipMemcpyAsync(c->rd_pos[buf].p, reads->pos, 2 * total * 4, hipMemcpyHostToDevice, st)); SQ_HIP_CHECK(hipMemcpyAsync(c->rd_score[buf].p, reads->aligner_score, total * 4, hipMemcpyHostToDevice, st));
    }
    c->rd_have[buf] = true;
  }
  SQ_HIP_CHECK(hipEventRecord(c->ev_map_done[buf], st));
  SQ_HIP_CHECK(hipStreamSynchronize(st));          // the caller's arrays may go
  c->last_n = n; c->last_paired = 1; c->last_total_aln = total; c->last_joint = num_with_joint_hits; c->have_batch = true; c->last_buf = buf; c->cur_buf = buf ^ 1;
  c->last_src = c; c->api_have = true;
  c->acc_n = n; c->acc_buf = buf; c->acc_total_aln = total; c->acc_joint = num_with_joint_hits;
  return SQ_OK;
}

// ---- lanes: asynchronous submit / in-order wait ---------------------------------------------------
static void lane_worker(sq_ctx* c) {
  (void)hipSetDevice(c->device);
  for (;;) {
    std::shared_ptr<sq_ctx::map_job> J;
    { std::unique_lock<std::mutex> lk(c->lane_mu); c->lane_cv.wait(lk, [&] { return c->lane_stop || !c->lane_q.empty(); });
      if (c->lane_q.empty()) return;
      J = c->lane_q.front(); c->lane_q.pop_front(); }
    int rc = sq_map_batch_impl(c, &J->in, J->has_out ? &J->out : nullptr, &J->st);
    {
      std::lock_guard<std::mutex> lk(c->lane_mu);
      J->rc = rc;
      if (rc) J->err = sq_last_error();
      J->n = c->last_n;
      J->buf = c->last_buf;
      J->total_aln = c->last_total_aln;
      J->joint = c->last_joint;
      J->done = true;
    }
    c->lane_cv_done.notify_all();
  }
}
extern "C" int sq_ctx_set_lanes(sq_ctx* c, int lanes) {
  if (!c || c->owner || lanes < 1 || lanes > 4) { sq_set_error("sq_ctx_set_lanes: 1..4 lanes"); return SQ_ERR_ARG; }
  if (!c->tickets.empty()) { sq_set_error("sq_ctx_set_lanes: batches are in flight"); return SQ_ERR_STATE; }
  c->n_lanes = lanes; c->submitted = 0;
  return SQ_OK;
}
extern "C" int sq_map_submit(sq_ctx* c, const sq_read_batch* in, sq_aln_batch* out) {
  if (!c || c->owner || !in) { sq_set_error("sq_map_submit: bad arguments"); return SQ_ERR_ARG; }
  if (c->n_lanes == 0) c->n_lanes = std::max(1, std::min(4, getenv("SQ_MAP_LANES") ? atoi(getenv("SQ_MAP_LANES")) : 2));
  const size_t want_lanes = (size_t)c->n_lanes;
  while (c->shadows.size() + 1 < want_lanes) {
    sq_ctx* sh = nullptr;
    int rc = ctx_create_lane(c->idx, &c->opts, c->device, c->max_reads, c, &sh);
    if (rc) return rc;
    sh->prof_on = c->prof_on;
    c->shadows.push_back(sh);
  }
  if (c->tickets.size() >= want_lanes) {
    sq_set_error("sq_map_submit: %zu batches already in flight (one per lane); call sq_map_wait first", c->tickets.size());
    return SQ_ERR_STATE;
  }
  sq_ctx* lane = (c->submitted % want_lanes) == 0 ? c : c->shadows[(c->submitted % want_lanes) - 1];
  auto J = std::make_shared<sq_ctx::map_job>(); J->in = *in; if (out) { J->out = *out; J->has_out = true; }
  {
    std::lock_guard<std::mutex> lk(lane->lane_mu);
    if (!lane->lane_thread.joinable()) lane->lane_thread = std::thread(lane_worker, lane);
    lane->lane_q.push_back(J);
  }
  lane->lane_cv.notify_one();
  c->tickets.emplace_back(lane, J); c->submitted++;
  return SQ_OK;
}
extern "C" int sq_map_wait(sq_ctx* c, sq_aln_batch* out, sq_map_stats* stats) {
  if (!c || c->owner) { sq_set_error("sq_map_wait: bad context"); return SQ_ERR_ARG; }
  if (c->tickets.empty()) { sq_set_error("sq_map_wait: nothing submitted"); return SQ_ERR_STATE; }
  auto tk = c->tickets.front(); c->tickets.pop_front();
  sq_ctx* lane = tk.first; auto J = tk.second;
  { std::unique_lock<std::mutex> lk(lane->lane_mu); lane->lane_cv_done.wait(lk, [&] { return J->done; }); }
  if (J->rc) { sq_set_error("%s", J->err.c_str()); return J->rc; }
  if (stats) *stats = J->st;
  if (out && J->has_out) *out = J->out;
  c->last_src = lane; c->api_have = true;
  c->acc_n = J->n; c->acc_buf = J->buf; c->acc_total_aln = J->total_aln; c->acc_joint = J->joint;
  return SQ_OK;
}

// alignments of the batch sq_map_wait / sq_map_batch returned last (they stay in that lane's buffers until the lane maps again)
extern "C" int sq_map_fetch(sq_ctx* c, sq_aln_batch* out) {
  if (!c || c->owner || !out) { sq_set_error("sq_map_fetch: bad arguments"); return SQ_ERR_ARG; }
  if (!c->api_have || !c->last_src) { sq_set_error("sq_map_fetch: no mapped batch"); return SQ_ERR_STATE; }
  sq_ctx* src = c->last_src; const uint32_t n = c->acc_n; const uint64_t total = c->acc_total_aln; const int buf = c->acc_buf;
  out->n = n;
  if (!out->aln && !out->read_off) {     // size query; with map_type given: the mapping types alone
    out->aln_cap = total;
    if (out->map_type) { SQ_HIP_CHECK(hipSetDevice(c->device)); SQ_HIP_CHECK(hipMemcpy(out->map_type, src->map_type.p, n, hipMemcpyDeviceToHost)); }
    return SQ_OK;
  }
  if (!out->read_off || (!out->aln && total)) { sq_set_error("sq_map_fetch: output arrays missing"); return SQ_ERR_ARG; }
  if (total > out->aln_cap) { sq_set_error("alignment buffer too small: need %llu, have %llu", (unsigned long long)total, (unsigned long long)out->aln_cap); return SQ_ERR_OVERFLOW; }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  SQ_HIP_CHECK(hipMemcpy(out->read_off, src->aln_off_ptr(buf), (size_t)(n + 1) * 8, hipMemcpyDeviceToHost));
  if (total) SQ_HIP_CHECK(hipMemcpy(out->aln, src->aln_ptr(buf), (size_t)total * sizeof(sq_aln), hipMemcpyDeviceToHost));
  if (out->map_type) SQ_HIP_CHECK(hipMemcpy(out->map_type, src->map_type.p, n, hipMemcpyDeviceToHost));
  return SQ_OK;
}

int sq_map_batch_impl(sq_ctx* c, const sq_read_batch* in, sq_aln_batch* out, sq_map_stats* stats) {
  if (!c || !in || !in->seq_off || !in->seq) { sq_set_error("sq_map_batch: bad arguments"); return SQ_ERR_ARG; }
  const uint32_t n = in->n, paired = in->paired ? 1 : 0, nrec = paired ? 2 * n : n;
  if (n > c->max_reads) { sq_set_error("batch of %u fragments exceeds ctx capacity %u", n, c->max_reads); return SQ_ERR_ARG; }
  SQ_HIP_CHECK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  c->have_batch = false;
  struct ActiveGuard { std::atomic<int>& a; explicit ActiveGuard(std::atomic<int>& x) : a(x) { a.fetch_add(1); } ~ActiveGuard() { a.fetch_sub(1); } } active_guard((c->owner ? c->owner : c)->map_active);
  const int buf = c->cur_buf;
  if (n == 0) {
    c->last_n = 0;
    c->last_buf = buf;
    c->last_paired = paired;
    c->last_total_aln = 0;
    c->have_batch = true;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (out && out->read_off) out->read_off[0] = 0;
    return SQ_OK;
  }
  // ---- stage reads in HBM ----
  const uint8_t* d_seq; const uint64_t* d_seq_off;
  if (in->on_device) { d_seq = in->seq; d_seq_off = in->seq_off; }
  else {
    uint64_t bytes = in->seq_off[nrec];
    if (c->seq.ensure(bytes + 16)) { sq_set_error("device allocation failed (reads)"); return SQ_ERR_NOMEM; }
    SQ_HIP_CHECK(hipMemcpyAsync(c->seq.p, in->seq, bytes, hipMemcpyHostToDevice, st));
    SQ_HIP_CHECK(hipMemcpyAsync(c->seq_off.p, in->seq_off, (size_t)(nrec + 1) * 8, hipMemcpyHostToDevice, st));
    d_seq = c->seq.p; d_seq_off = c->seq_off.p;
  }
  const sq_device_index* di = c->di; const sq_map_params& P = c->mp;
  int pack_attempt = 0, uni_attempt = 0, seed_attempt = 0;
pack_again:   // [r4] taken once more when the batch holds a read end longer than the packing stride allows (the stride is raised and stays raised)
  SQ_HIP_CHECK(hipMemsetAsync(c->stats.p, 0, ST_N * sizeof(unsigned long long), st));
  SQ_HIP_CHECK(hipMemsetAsync(c->counters.p, 0, 32 * sizeof(uint32_t), st));
  sq_prof_begin(c);
  if (c->read_words == 8) k_pack8_staged<<<nblk(nrec), 256, 0, st>>>(d_seq, d_seq_off, nrec, c->rpack.p, c->rnmask.p, c->rlen.p, c->rany.p, c->stats.p);   // [r6] the text through LDS (waves it does not fit run k_pack<8>'s code)
  else if (c->read_words == 16) k_pack<16><<<nblk(nrec), TB, 0, st>>>(d_seq, d_seq_off, nrec, c->rpack.p, c->rnmask.p, c->rlen.p, c->rany.p, c->stats.p);
  else k_pack<32><<<nblk(nrec), TB, 0, st>>>(d_seq, d_seq_off, nrec, c->rpack.p, c->rnmask.p, c->rlen.p, c->rany.p, c->stats.p);
  sq_prof_mark(c, SG_PACK);
  {  // persistent grid: 256 CUs x 24 waves; lanes pull read ends from counters[2].  The probe rate is bound by the memory system once six waves per SIMD are resident
     // (profiles/r06_seed_grid.txt: 16 / 20 / 24 waves per CU 5.30 / 4.76 / 4.38 ms, 26 and 28 no faster), so two blocks' worth of wave slots per CU stay free for the
     // eq stage's small kernels on stream2 — a full grid starves them for the whole launch.  [r6] k_seed2's blocks are ONE wave (nothing in it is shared between waves:
     // 4.38 -> 4.22 ms; the general kernel keeps 256 threads)
    const uint32_t grid = std::min<uint32_t>(nblk(nrec), 256u * 6u), grid2 = std::min<uint32_t>((nrec + SEED_TB - 1) / SEED_TB, 256u * 6u * (256u / SEED_TB));
    // [r5] k_seed2 (read words and filter block in LDS, the minimizer table's one-sector records) for the default k / m with reads of up to 256 bases;
    // everything else takes the general kernel (SQ_SEED_GENERAL=1 forces it: the tests run both)
    static const bool general = getenv("SQ_SEED_GENERAL") && atoi(getenv("SQ_SEED_GENERAL")) != 0;
    static const int force_lw = getenv("SQ_SEED_LW") ? atoi(getenv("SQ_SEED_LW")) : 0;
    const bool v2 = !general && P.k == 31 && di->dict.m == 20 && c->read_words == 8 && di->dict.kfilter && di->dict.uinfo && di->dict.mtab;
    if (v2) {
      if (force_lw == 5 || force_lw == 8) c->seed_lw = std::max<uint32_t>(c->seed_lw, (uint32_t)force_lw);
      const uint32_t lw = c->seed_lw;
      // LDS per wave: (LW + 8) x 512 B -> 6 KB (LW 4) or [r6] 6.5 KB (LW 5: reads of up to 160 bases): 24 waves per CU and more; 8 KB (LW 8): 20
#define SQ_SEED2_ARGS di->dict, P, nrec, c->rpack.p, c->rnmask.p, c->rlen.p, c->rany.p, c->unimems.p, c->n_uni.p, c->n_proj.p, c->stats.p, c->counters.p + 2, c->read_words, c->uni_slots
      if (lw == 4) k_seed2<31, 20, 2, 4><<<grid2, SEED_TB, 0, st>>>(SQ_SEED2_ARGS); else if (lw == 5) k_seed2<31, 20, 2, 5><<<grid2, SEED_TB, 0, st>>>(SQ_SEED2_ARGS);
      else k_seed2<31, 20, 2, 8><<<grid2, SEED_TB, 0, st>>>(SQ_SEED2_ARGS);
#undef SQ_SEED2_ARGS
    } else
#define SQ_SEED_ARGS di->dict, di->ctab_off, P, nrec, c->rpack.p, c->rnmask.p, c->rlen.p, c->unimems.p, c->n_uni.p, c->n_proj.p, c->stats.p, c->counters.p + 2, c->read_words, c->uni_slots
    if (P.k == 31 && di->dict.m == 20) {   // the default (k = 31, m = 20) gets the fully specialised kernel
      k_seed<31, 20, 2><<<grid, TB, 0, st>>>(SQ_SEED_ARGS);
    }
#undef SQ_SEED_ARGS
    else
      k_seed<0, 0><<<grid, TB, 0, st>>>(di->dict, di->ctab_off, P, nrec, c->rpack.p, c->rnmask.p, c->rlen.p, c->unimems.p, c->n_uni.p,
          c->n_proj.p, c->stats.p,
          c->counters.p + 2, c->read_words, c->uni_slots);
  }
  sq_prof_mark(c, SG_SEED);
  SQ_HIP_CHECK(hipMemsetAsync(c->n_proj.p + nrec, 0, sizeof(uint32_t), st));
  int rc = exclusive_scan_u32(c, c->n_proj.p, c->mem_off.p, nrec + 1); if (rc) return rc;
  // size classes of the ends (mem_kernels.h); their counts come back with the MEM total in the same read-back
  if (c->mlist.ensure((size_t)MK_NCLS * nrec + 8) || c->mlbase.ensure((size_t)nrec + 8) || c->mlinfo.ensure((size_t)(MK_NCLS - 1) * nrec + 8)) { sq_set_error("device allocation failed (MEM class lists)"); return SQ_ERR_NOMEM; }
  uint32_t* const lists = c->mlist.p; uint32_t* const list_l = lists + (size_t)(MK_NCLS - 1) * nrec;
  SQ_HIP_CHECK(hipMemsetAsync(c->counters.p + 16, 0, (MK_NCLS + 1) * sizeof(uint32_t), st));
  k_mem_classes<<<(nrec + 1023) / 1024, 1024, 0, st>>>(nrec, c->n_proj.p, c->mem_off.p, c->n_uni.p, c->rlen.p, lists, c->mlinfo.p, c->mlbase.p, c->n_chains.p,
      c->counters.p + 16);
  uint64_t total_mems = 0; uint32_t hcls[MK_NCLS + 1] = {0, 0, 0, 0, 0, 0, 0, 0};
  sq_prof_mark(c, SG_SCAN_MEMS);
  unsigned long long h_maxlen = 0, h_uniover = 0, h_seedlw = 0;
  SQ_HIP_CHECK(hipMemcpyAsync(&h_uniover, c->stats.p + ST_UNIOVER, 8, hipMemcpyDeviceToHost, st));
  SQ_HIP_CHECK(hipMemcpyAsync(&h_seedlw, c->stats.p + ST_SEEDLW, 8, hipMemcpyDeviceToHost, st));
  SQ_HIP_CHECK(hipMemcpyAsync(&total_mems, c->mem_off.p + nrec, 8, hipMemcpyDeviceToHost, st));
  SQ_HIP_CHECK(hipMemcpyAsync(hcls, c->counters.p + 16, sizeof(hcls), hipMemcpyDeviceToHost, st));
  SQ_HIP_CHECK(hipMemcpyAsync(&h_maxlen, c->stats.p + ST_MAXLEN, 8, hipMemcpyDeviceToHost, st));
  SQ_HIP_CHECK(hipStreamSynchronize(st));
  if (h_maxlen) {   // [r4] a read end did not fit: refuse it (the reference has no limit; this path maps reads of up to SQ_MAX_READ_LEN bases) or widen the stride and pack again
    if (h_maxlen > SQ_MAX_READ_LEN) { sq_set_error("a read of %llu bases: the GPU path maps reads of up to %u bases (nothing is cut silently)", h_maxlen, SQ_MAX_READ_LEN); return SQ_ERR_ARG; }
    if (P.recover_orphans && paired && h_maxlen > 256) { sq_set_error("--recoverOrphans is built for reads of up to 256 bases (a read of %llu bases is in the batch)", h_maxlen); return SQ_ERR_ARG; }
    const uint32_t need = h_maxlen <= 512 ? 16u : 32u;
    if (pack_attempt++ || need <= c->read_words) { sq_set_error("internal: read of %llu bases with a stride of %u words", h_maxlen, c->read_words); return SQ_ERR_STATE; }
    c->read_words = need;
    const size_t cap_ends = std::max<size_t>(2 * (size_t)c->max_reads, nrec);
    if (c->rpack.ensure(cap_ends * need + 8) || c->rnmask.ensure(cap_ends * (need / 2) + 8)) { sq_set_error("device allocation failed (packed reads of up to %u bases)", 32 * need); return SQ_ERR_NOMEM; }
    goto pack_again;
  }
  if (h_seedlw) {   // [r5] reads of more than 128 bases met the four-word instantiation of k_seed2: the eight-word one from now on, and this batch again
    if (c->seed_lw >= 8 || seed_attempt++) { sq_set_error("internal: a read end of %llu bases did not fit k_seed2's LDS column of %u words", h_seedlw, c->seed_lw); return SQ_ERR_STATE; }
    c->seed_lw = h_seedlw <= 160 ? 5 : 8;       // (h_seedlw: the longest end that did not fit)
    goto pack_again;
  }
  if (h_uniover) {   // [r4] read ends with more uni-MEMs than the slab has slots per end (the reference keeps them all): a wider slab, and the batch is seeded again
    const uint32_t need = c->uni_slots * 2;
    const size_t cap_ends = std::max<size_t>(2 * (size_t)c->max_reads, nrec);
    if (need > SQ_MAX_UNI_SLOTS || uni_attempt++ >= 6 || cap_ends * need * sizeof(sq_unimem_dev) > ((size_t)96 << 30)) {
      sq_set_error("%llu read ends have more than %u uni-MEMs and the slab cannot grow further; split the batch", h_uniover, c->uni_slots); return SQ_ERR_OVERFLOW; }
    c->uni_slots = need;
    if (c->unimems.ensure(cap_ends * need)) { sq_set_error("device allocation failed (uni-MEM slab of %u slots per end); split the batch", need); return SQ_ERR_NOMEM; }
    goto pack_again;
  }
  c->last_total_mems = total_mems;
  if (total_mems >= 0x7FFFFFF0ull) {   // 32-bit slab indices (candidates name chains by slab index; recovery doubles the slabs)
    sq_set_error("too many MEMs in one batch (%llu); split the batch", (unsigned long long)total_mems);
    return SQ_ERR_OVERFLOW;
  }
  const size_t MP = (size_t)total_mems + 8;
  const bool recover = P.recover_orphans && paired;   // recovered mates live in a second set of chain slabs (k_recover)
  if (c->mkey2.ensure(MP) || c->mval2.ensure(MP) || c->mnext.ensure(MP) || c->chains.ensure(recover ? 2 * MP : MP)) {
    sq_set_error("device allocation failed for %llu MEMs; split the batch", (unsigned long long)total_mems); return SQ_ERR_NOMEM; }
  uint64_t* skey = c->mkey2.p; uint64_t* sval = c->mval2.p;
  const uint32_t nL = hcls[MK_NCLS - 1], memsL = hcls[MK_NCLS];
#define SQ_MEMS_ARGS(cls) di->dict.uoff, di->ctab_off, di->ctab, di->ref_accum, P, c->gapcost.p, c->mlinfo.p + (size_t)(cls) * nrec, hcls[cls], c->unimems.p, c->uni_slots, \
      skey, sval, c->mnext.p, c->chains.p, c->n_chains.p
  if (hcls[0]) k_mems<8, 8, 256><<<(hcls[0] + 31) / 32, 256, 0, st>>>(SQ_MEMS_ARGS(0));   // [r3] the common end has <= 8 MEMs: eight ends per wave
  if (hcls[1]) k_mems<16, MK_X_CAP, 256><<<(hcls[1] + 15) / 16, 256, 0, st>>>(SQ_MEMS_ARGS(1));
  if (hcls[2]) k_mems<16, MK_T_CAP, 256><<<(hcls[2] + 15) / 16, 256, 0, st>>>(SQ_MEMS_ARGS(2));
  if (hcls[3]) k_mems<16, MK_S_CAP, 256><<<(hcls[3] + 15) / 16, 256, 0, st>>>(SQ_MEMS_ARGS(3));
  if (hcls[4]) k_mems<64, MK_L_CAP, 128><<<(hcls[4] + 1) / 2, 128, 0, st>>>(SQ_MEMS_ARGS(4));
#undef SQ_MEMS_ARGS
  sq_prof_mark(c, SG_PROJECT);
  if (nL) {   // ends with more than MK_L_CAP MEMs (repeats): compact projection, library radix sort, back into the slabs, HBM chaining
    const size_t LP = (size_t)memsL + 8;
    if (c->mkey.ensure(LP) || c->mval.ensure(LP) || c->lkey.ensure(LP) || c->lval.ensure(LP) || c->cf.ensure(MP) || c->cp.ensure(MP) || c->mused.ensure(MP)) {
      sq_set_error("device allocation failed for %u MEMs of large read ends; split the batch", memsL); return SQ_ERR_NOMEM; }
    k_project_list<<<(nL + 3) / 4, 256, 0, st>>>(di->dict, di->ctab_off, di->ctab, di->ref_accum, P, list_l, c->mlbase.p, nL, c->rlen.p, c->unimems.p, c->uni_slots,
        c->n_uni.p, c->mkey.p, c->mval.p);
    int endbits = 1; while ((1ull << endbits) < nrec) ++endbits;
    size_t tmp = 0;
    // [r6] every end's records lie together in the compact projection (k_mem_classes hands each large end a run of n_proj[e] records), so the sort by (end, position) is a
    // SEGMENTED sort by position alone: a block per end sorts its run in LDS (one trip through memory for an end of up to 4 096 MEMs, where the sort of the whole buffer by
    // 40 + endbits bits made eight).  Stable, as the whole-buffer sort was: MEMs at the same position keep their emission order.  The ends then stand in the order the cursor
    // handed out their runs, not by end id — nothing behind this point asks for that (boundaries are found by comparing neighbours, the chains are grouped by a sort of their own).
    int posbits = 1; while (posbits < 40 && (1ull << posbits) < c->idx->ref_accum.back()) ++posbits;
    rocprim::counting_iterator<uint32_t> seg_it(0u);
    typedef rocprim::transform_iterator<rocprim::counting_iterator<uint32_t>, LgSegBound, uint32_t> SegIt;
    SegIt seg_begin(seg_it, LgSegBound{c->mlbase.p, list_l, c->n_proj.p, 0u}), seg_end(seg_it, LgSegBound{c->mlbase.p, list_l, c->n_proj.p, 1u});
    (void)rocprim::segmented_radix_sort_pairs(nullptr, tmp, c->mkey.p, c->lkey.p, c->mval.p, c->lval.p, memsL, nL, seg_begin, seg_end, 0u, (unsigned)posbits, st);
    if (c->sort_tmp.ensure(tmp + 256)) { sq_set_error("sort temp allocation failed"); return SQ_ERR_NOMEM; }
    tmp = c->sort_tmp.n;
    SQ_HIP_CHECK(rocprim::segmented_radix_sort_pairs(c->sort_tmp.p, tmp, c->mkey.p, c->lkey.p, c->mval.p, c->lval.p, memsL, nL, seg_begin, seg_end, 0u, (unsigned)posbits, st));
    {   // [r4] flat passes over the sorted compact records (mem_kernels.h: k_lg_*); cf / cp / mused are indexed by compact record here
      if (c->lg_a.ensure(LP) || c->lg_b.ensure(LP) || c->lg_c.ensure(LP) || c->lg_d.ensure(LP) || c->lg_flags.ensure(LP) || c->lg_first.ensure((size_t)nrec + 8) || c->lg_cnt.ensure(8)) {
        sq_set_error("device allocation failed for %u MEMs of large read ends; split the batch", memsL); return SQ_ERR_NOMEM; }
      uint64_t* se_in = c->mkey.p; uint64_t* se = c->mval.p;   // the unsorted compact projection is dead once it is sorted
      uint64_t* gbest = c->lg_a.p; uint64_t* ebest = c->lg_b.p;
      k_lg_flags<<<nblk(memsL), TB, 0, st>>>(memsL, c->lkey.p, c->lval.p, c->lg_flags.p, se_in, gbest, ebest, c->n_chains.p);
      auto tmp_for = [&](size_t need) -> int { if (c->sort_tmp.ensure(need + 256)) { sq_set_error("sort temp allocation failed"); return SQ_ERR_NOMEM; } return SQ_OK; };
      size_t t1 = 0, t3 = 0, t4 = 0;
      rocprim::counting_iterator<uint32_t> cnt_it(0u);
      (void)rocprim::inclusive_scan(nullptr, t1, se_in, se, (size_t)memsL, LgMaxPair(), st);
      (void)rocprim::radix_sort_pairs(nullptr, t3, c->lg_a.p, c->lg_b.p, c->lg_d.p, c->lg_c.p, (size_t)memsL, 0u, 64u, st);
      (void)rocprim::select(nullptr, t4, cnt_it, (const uint8_t*)c->lg_flags.p, c->lg_d.p, c->lg_cnt.p + 1, (size_t)memsL, st);
      if (int rc = tmp_for(std::max(t1, std::max(t3, t4)))) return rc;
      size_t tb = c->sort_tmp.n;
      SQ_HIP_CHECK(rocprim::inclusive_scan(c->sort_tmp.p, tb, se_in, se, (size_t)memsL, LgMaxPair(), st));
      k_scatter_sorted<<<nblk(memsL), TB, 0, st>>>(memsL, c->lkey.p, c->lval.p, se, c->mem_off.p, skey, sval);   // the sorted records into the ends' slabs (what scoring reads)
      const uint32_t lg_cap = LG_T + (getenv("SQ_LG_OVERHANG") ? (uint32_t)std::min(std::max(atoi(getenv("SQ_LG_OVERHANG")), 0), (int)LG_O) : LG_O);   // (tests: 0 sends every cluster that crosses a tile's edge through the global-memory path)
      k_lg_dp2<<<(memsL + LG_T - 1) / LG_T, LG_TB, 0, st>>>(memsL, lg_cap, c->lkey.p, c->lval.p, c->lg_flags.p, se, P, c->gapcost.p, c->cf.p, c->cp.p, c->mused.p, gbest);
      k_lg_accept2<<<(memsL + LG_T - 1) / LG_T, LG_TB, 0, st>>>(memsL, lg_cap, c->lg_flags.p, se, P, c->cf.p, c->cp.p, c->mused.p, gbest, ebest);
      k_lg_keep<<<nblk(memsL), TB, 0, st>>>(memsL, se, P, c->cf.p, c->mused.p, ebest, c->lg_flags.p);
      tb = c->sort_tmp.n; SQ_HIP_CHECK(rocprim::select(c->sort_tmp.p, tb, cnt_it, (const uint8_t*)c->lg_flags.p, c->lg_d.p, c->lg_cnt.p + 1, (size_t)memsL, st));
      uint32_t nkept = 0;
      SQ_HIP_CHECK(hipMemcpyAsync(&nkept, c->lg_cnt.p + 1, 4, hipMemcpyDeviceToHost, st)); SQ_HIP_CHECK(hipStreamSynchronize(st));
      if (nkept) {
        int tbits = 1; while ((1ull << tbits) < (uint64_t)di->num_refs) ++tbits;
        // by score (descending; the sort is stable, so equal scores stay in index order), then by (end, transcript): k_chain's output order
        k_lg_key_score<<<nblk(nkept), TB, 0, st>>>(nkept, c->lg_d.p, c->cf.p, c->lg_a.p);
        tb = c->sort_tmp.n; SQ_HIP_CHECK(rocprim::radix_sort_pairs(c->sort_tmp.p, tb, c->lg_a.p, c->lg_b.p, c->lg_d.p, c->lg_c.p, (size_t)nkept, 0u, 64u, st));
        k_lg_key_group<<<nblk(nkept), TB, 0, st>>>(nkept, c->lg_c.p, c->lkey.p, c->lval.p, tbits, c->lg_a.p);
        tb = c->sort_tmp.n; SQ_HIP_CHECK(rocprim::radix_sort_pairs(c->sort_tmp.p, tb, c->lg_a.p, c->lg_b.p, c->lg_c.p, c->lg_d.p, (size_t)nkept, 0u, (unsigned)std::min(64, endbits + tbits), st));
        k_lg_first<<<nblk(nkept), TB, 0, st>>>(nkept, c->lg_b.p, tbits, c->lg_first.p);
        k_lg_write<<<nblk(nkept), TB, 0, st>>>(nkept, c->lg_b.p, tbits, c->lg_d.p, c->lg_first.p, c->lkey.p, c->lval.p, se, di->ref_accum, c->rlen.p, c->mem_off.p,
            c->cf.p, c->cp.p, c->mnext.p, c->chains.p, c->n_chains.p);
      }
    }
  }
  sq_prof_mark(c, SG_SORT);
  k_count_kmer_frags<<<std::min<uint32_t>(nblk(n), 1024u), TB, 0, st>>>(n, paired, c->n_chains.p, c->stats.p);
  // chains stay in their per-end slabs (slab of end e starts at mem_off[e]: #chains <= #MEMs); candidates refer to them by
  // absolute slab index.  (A dense copy used to be made here: 0.8 ms and 1.8 GB of traffic per 10^6 pairs for nothing.)
  sq_prof_mark(c, SG_CHAIN);
  // single-pass join; candidate blocks come from a global cursor (stats slot reused as the 64-bit cursor)
  uint64_t total_cands = 0;
  {
    size_t cap_guess = std::max<size_t>(c->cands.n, (size_t)n * 8 + 1024);
    for (int attempt = 0; attempt < 3; ++attempt) {
      if (c->cands.ensure(cap_guess) || c->cand_frag.ensure(cap_guess)) {
        sq_set_error("device allocation failed for candidates; split the batch");
        return SQ_ERR_NOMEM;
      }
      SQ_HIP_CHECK(hipMemsetAsync(c->stats.p + ST_CANDS, 0, sizeof(unsigned long long), st));
      SQ_HIP_CHECK(hipMemsetAsync(c->counters.p + 15, 0, sizeof(uint32_t), st));
      uint32_t* rest = c->mlist.p;   // the class lists of k_mems are done with: n entries hold the fragments left to k_join2_rest
      k_join2<<<nblk(n), TB, 0, st>>>(P, n, paired, c->mem_off.p, c->chains.p, c->n_chains.p, c->n_cand.p, c->cand_off.p, c->cands.p,
          c->cand_frag.p, c->cands.n,
          c->frag_flags.p, c->stats.p + ST_CANDS, rest, c->counters.p + 15);
      if (paired) k_join2_group<<<std::min<uint32_t>((n + 15) / 16, 192u * 12u), 256, 0, st>>>(P, c->mem_off.p, c->chains.p, c->n_chains.p, c->n_cand.p, c->cand_off.p, c->cands.p, c->cand_frag.p,
          c->cands.n, c->frag_flags.p, c->stats.p + ST_CANDS, rest, c->counters.p + 15);
      else k_join2_rest<<<nblk(n), TB, 0, st>>>(P, paired, c->mem_off.p, c->chains.p, c->n_chains.p, c->n_cand.p, c->cand_off.p, c->cands.p, c->cand_frag.p,
          c->cands.n, c->frag_flags.p, c->stats.p + ST_CANDS, rest, c->counters.p + 15);
      unsigned long long tc = 0;
      SQ_HIP_CHECK(hipMemcpyAsync(&tc, c->stats.p + ST_CANDS, 8, hipMemcpyDeviceToHost, st));
      SQ_HIP_CHECK(hipStreamSynchronize(st));
      total_cands = tc;
      if (total_cands <= c->cands.n) break;
      if (attempt == 2) { sq_set_error("candidate array overflow (%llu)", tc); return SQ_ERR_OVERFLOW; }
      cap_guess = (size_t)total_cands + 1024;
    }
  }
  sq_prof_mark(c, SG_JOIN_FILL);
  c->last_total_cands = total_cands;
  const size_t CP = (size_t)total_cands + 8;
  static_assert(sizeof(sq_aln) == 40, "sq_aln layout");
  bool wait_eq = false;
  if (c->eq_pending[buf]) {   // the eq stage that read this alignment buffer two batches ago: wait (on the device) for it
    sq_eq_wait_enqueued(c->owner ? c->owner : c, c->eq_job_of_buf[buf]);
    if ((buf ? c->aln_b1.n : c->aln.n) < CP) SQ_HIP_CHECK(hipEventSynchronize(c->ev_eq_done[buf]));   // the buffer is about to be reallocated: the eq stage must be done with it
    // [r6] the wait itself stands in front of k_select, the only kernel that writes what the eq stage reads (the alignment array and its offsets): in front of k_score it held
    // the mapping stream ~0.35 ms per 5 x 10^6 pairs while scoring, the DP and k_finalize (3.2 ms) had nothing to do with that buffer
    wait_eq = true;
  }
  if (c->aln_slots.ensure(std::max(CP, c->cands.n)) || (buf ? c->aln_b1.ensure(CP) : c->aln.ensure(CP)) ||
      c->dpq.ensure(std::max<size_t>(c->dpq.n, CP * 2 + 1024))) {
    sq_set_error("device allocation failed for %llu candidates; split the batch", (unsigned long long)total_cands);
    return SQ_ERR_NOMEM;
  }
  sq_dbuf<uint32_t>& cand_frag = c->cand_frag; sq_dbuf<int32_t>& hs_arr = c->hs_arr; sq_dbuf<uint32_t>& tid_arr = c->tid_arr;
  if (hs_arr.ensure(CP) || tid_arr.ensure(CP)) { sq_set_error("device allocation failed (candidate side arrays)"); return SQ_ERR_NOMEM; }
  ScoreCtx S;
  S.refseq = di->refseq;
  S.ref_accum = di->ref_accum;
  S.ref_len = di->ref_len;
  S.rpack = c->rpack.p;
  S.rnmask = c->rnmask.p;
  S.rlen = c->rlen.p; S.rw = c->read_words;
  S.mkey = skey;
  S.mval = sval;
  S.mnext = c->mnext.p;
  S.dpq = c->dpq.p;
  S.counters = c->counters.p;
  S.dpq_cap = (uint32_t)std::min<size_t>(c->dpq.n, 0xFFFFFFFFu);
  uint32_t hcount[2] = {0, 0};
  c->last_chain_slots = recover ? 2 * total_mems : total_mems;
  if (total_cands && recover) k_recover<<<nblk(n), TB, 0, st>>>(P, S, n, c->cand_off.p, c->n_cand.p, c->cands.p, c->chains.p,
      (uint32_t)total_mems, c->stats.p);
  if (total_cands) {
    for (int attempt = 0; attempt < 2; ++attempt) {
      SQ_HIP_CHECK(hipMemsetAsync(c->counters.p + 8, 0, 2 * sizeof(uint32_t), st));
      SQ_HIP_CHECK(hipMemsetAsync(c->stats.p + ST_DP, 0, sizeof(unsigned long long), st));
      k_score<<<nblk(total_cands), TB, 0, st>>>(P, S, total_cands, paired, c->mem_off.p, c->cand_off.p, n, c->chains.p, c->cands.p,
          cand_frag.p, c->stats.p);
      sq_prof_mark(c, SG_SCORE);
      SQ_HIP_CHECK(hipMemcpyAsync(hcount, c->counters.p + 8, sizeof(hcount), hipMemcpyDeviceToHost, st));
      SQ_HIP_CHECK(hipStreamSynchronize(st));
      if (hcount[0] <= S.dpq_cap) break;
      if (attempt == 1 || c->dpq.ensure((size_t)hcount[0] + 1024)) {
        sq_set_error("DP queue overflow (%u regions)", hcount[0]);
        return SQ_ERR_OVERFLOW;
      }
      S.dpq = c->dpq.p; S.dpq_cap = (uint32_t)c->dpq.n;
    }
    if (hcount[0]) {
      const uint32_t nq = hcount[0], nb = std::min<uint32_t>(1024u, (nq + 4095) / 4096), chunk = (nq + nb - 1) / nb;
      if (c->dp_bh.ensure((size_t)DP_CLASSES * nb + 8) || c->dp_off.ensure((size_t)DP_CLASSES * nb + 8) || c->dp_perm.ensure((size_t)nq + 8) ||
          c->sort_tmp.ensure((size_t)scan_tiles((uint64_t)DP_CLASSES * nb) * 8 + 256)) { sq_set_error("device allocation failed (DP queue order)"); return SQ_ERR_NOMEM; }
      k_dp_hist<<<nb, 256, 0, st>>>(S.dpq, nq, chunk, nb, c->dp_bh.p);
      exclusive_scan_u32_u64(c->dp_bh.p, c->dp_off.p, (uint64_t)DP_CLASSES * nb, (uint64_t*)c->sort_tmp.p, st);
      k_dp_scatter<<<nb, 256, 0, st>>>(S.dpq, nq, chunk, nb, c->dp_off.p, c->dp_perm.p);
      if (P.bw == SQ_MAX_BAND) k_dp<<<(nq + 63) / 64, 64, 0, st>>>(P, S, nq, c->cands.p, cand_frag.p, paired, c->dp_perm.p);   // full band: the condition-free form
      else k_dp_general<<<(nq + 63) / 64, 64, 0, st>>>(P, S, nq, c->cands.p, cand_frag.p, paired, c->dp_perm.p);
    }
    sq_prof_mark(c, SG_DP);
  }
  if (total_cands) k_finalize<<<nblk(total_cands), TB, 0, st>>>(P, total_cands, paired, c->cands.p, cand_frag.p, c->rlen.p, hs_arr.p,
      tid_arr.p);
  sq_prof_mark(c, SG_FINALIZE);
  // [r5] selection + compact alignment array in one launch (map_kernels.h): one descriptor per block for the look-back, a ticket counter in front of them
  const uint32_t sel_blocks = (n + SEL_TB - 1) / SEL_TB;
  if (c->sel_desc.ensure((size_t)sel_blocks + 8)) { sq_set_error("device allocation failed (selection descriptors)"); return SQ_ERR_NOMEM; }
  SQ_HIP_CHECK(hipMemsetAsync(c->sel_desc.p, 0, ((size_t)sel_blocks + 2) * 8, st));
  if (wait_eq) { SQ_HIP_CHECK(hipStreamWaitEvent(st, c->ev_eq_done[buf], 0)); c->eq_pending[buf] = false; }
  if (n == 0) SQ_HIP_CHECK(hipMemsetAsync(c->aln_off_ptr(buf), 0, 8, st));
  else k_select<<<sel_blocks, SEL_TB, 0, st>>>(P, n, paired, c->cand_off.p, c->n_cand.p, c->cands.p, hs_arr.p, tid_arr.p, c->rlen.p,
      c->frag_flags.p, c->aln_slots.p, c->n_aln.p, c->map_type.p, c->stats.p, c->aln_ptr(buf), c->aln_off_ptr(buf),
      (unsigned long long*)c->sel_desc.p + 1, (uint32_t*)c->sel_desc.p);
  sq_prof_mark(c, SG_SELECT);
  SQ_HIP_CHECK(hipEventRecord(c->ev_map_done[buf], st));
  uint64_t total_aln = 0; unsigned long long hst[ST_N];
  SQ_HIP_CHECK(hipMemcpyAsync(&total_aln, c->aln_off_ptr(buf) + n, 8, hipMemcpyDeviceToHost, st));
  SQ_HIP_CHECK(hipMemcpyAsync(hst, c->stats.p, sizeof(hst), hipMemcpyDeviceToHost, st));
  SQ_HIP_CHECK(hipStreamSynchronize(st));
  sq_prof_end(c);
  c->last_n = n;
  c->last_paired = paired;
  c->last_total_aln = total_aln;
  c->last_joint = hst[ST_JOINT]; c->rd_have[buf] = false;
  c->seed_fills += hst[ST_FILLS];
  c->have_batch = true;
  c->last_buf = buf;
  c->cur_buf = buf ^ 1;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->num_reads = n;
    stats->num_mapped_at_least_a_kmer = hst[ST_KMER];
    stats->num_with_joint_hits = hst[ST_JOINT];
    stats->num_mapped = hst[ST_MAPPED];
    stats->num_alignments = hst[ST_ALNS];
    stats->num_mappings_filtered = hst[ST_MAPFILT];
    stats->num_fragments_filtered = hst[ST_FRAGFILT];
    stats->num_dovetails = hst[ST_DOVETAIL];
    stats->num_decoy_fragments = hst[ST_DECOY];
    stats->num_seeds = hst[ST_SEEDS];
    stats->num_lookups = hst[ST_LOOKUPS];
    stats->num_mems = total_mems;
    stats->num_chains = hst[ST_CHAINS];
    stats->num_candidates = total_cands;
    stats->num_dp_alignments = hst[ST_DP];
    stats->num_orphans_rescued = hst[ST_RESCUED];
    stats->num_truncated_ends = hst[ST_TRUNC];
  }
  if (out) {
    if (!out->read_off || (!out->aln && total_aln)) { sq_set_error("sq_map_batch: output arrays missing"); return SQ_ERR_ARG; }
    if (total_aln > out->aln_cap) {
      sq_set_error("alignment buffer too small: need %llu, have %llu", (unsigned long long)total_aln, (unsigned long long)out->aln_cap);
      return SQ_ERR_OVERFLOW;
    }
    SQ_HIP_CHECK(hipMemcpy(out->read_off, c->aln_off_ptr(buf), (size_t)(n + 1) * 8, hipMemcpyDeviceToHost));
    if (total_aln) SQ_HIP_CHECK(hipMemcpy(out->aln, c->aln_ptr(buf), (size_t)total_aln * sizeof(sq_aln), hipMemcpyDeviceToHost));
    if (out->map_type) SQ_HIP_CHECK(hipMemcpy(out->map_type, c->map_type.p, n, hipMemcpyDeviceToHost));
    out->n = n;
  }
  return SQ_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int sq_debug_infix_align(int device, uint32_t ncases, const uint8_t* queries, const uint64_t* q_off, const uint8_t* windows,
    const uint64_t* w_off,
    const int32_t* k, int32_t* out) {
  if (!ncases) return SQ_OK;
  if (!queries || !q_off || !windows || !w_off || !k || !out) { sq_set_error("sq_debug_infix_align: null argument"); return SQ_ERR_ARG; }
  if (hipSetDevice(device) != hipSuccess) { sq_set_error("sq_debug_infix_align: no device %d", device); return SQ_ERR_DEVICE; }
  // host-side packing into the layouts the pipeline uses: read ends as SQ_READ_WORDS_MIN 2-bit words + N mask, text as one 2-bit pool
  std::vector<uint64_t> rp((size_t)ncases * SQ_READ_WORDS_MIN, 0), rn((size_t)ncases * (SQ_READ_WORDS_MIN / 2), 0), toff(ncases + 1, 0);
  std::vector<uint16_t> rl(ncases);
  auto code = [](uint8_t ch) -> int { switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } };
  for (uint32_t i = 0; i < ncases; ++i) {
    const uint64_t n = q_off[i + 1] - q_off[i];
    if (n > 256) {
      sq_set_error("sq_debug_infix_align: query %u longer than 256", i);
      return SQ_ERR_ARG;
    }
    rl[i] = (uint16_t)n;
    for (uint64_t j = 0; j < n; ++j) {
      const int cd = code(queries[q_off[i] + j]);
      if (cd > 3) rn[(size_t)i * (SQ_READ_WORDS_MIN / 2) + (j >> 6)] |= 1ull << (j & 63);
      else rp[(size_t)i * SQ_READ_WORDS_MIN + (j >> 5)] |= (uint64_t)cd << ((j & 31) * 2);
    }
    toff[i + 1] = toff[i] + (w_off[i + 1] - w_off[i]);
  }
  std::vector<uint64_t> text((size_t)(toff[ncases] >> 5) + 2, 0);
  for (uint32_t i = 0; i < ncases; ++i) for (uint64_t j = 0, m = w_off[i + 1] - w_off[i]; j < m; ++j) {
    const int cd = code(windows[w_off[i] + j]);
    if (cd > 3) {
      sq_set_error("sq_debug_infix_align: window %u holds a non-ACGT byte", i);
      return SQ_ERR_ARG;
    }
    const uint64_t p = toff[i] + j; text[p >> 5] |= (uint64_t)cd << ((p & 31) * 2);
  }
  sq_dbuf<uint64_t> d_rp, d_rn, d_text, d_toff; sq_dbuf<uint16_t> d_rl; sq_dbuf<int32_t> d_k, d_out;
  int rc = SQ_OK;
  if (d_rp.ensure(rp.size()) || d_rn.ensure(rn.size()) || d_text.ensure(text.size()) || d_toff.ensure(toff.size()) ||
      d_rl.ensure(ncases) || d_k.ensure(ncases) ||
      d_out.ensure((size_t)4 * ncases)) {
    sq_set_error("sq_debug_infix_align: device allocation failed");
    rc = SQ_ERR_NOMEM;
  }
  auto up = [&](void* d, const void* h, size_t b) {
    if (rc == SQ_OK && hipMemcpy(d, h, b, hipMemcpyHostToDevice) != hipSuccess) {
      sq_set_error("sq_debug_infix_align: copy failed");
      rc = SQ_ERR_DEVICE;
    }
  };
  up(d_rp.p, rp.data(), rp.size() * 8);
  up(d_rn.p, rn.data(), rn.size() * 8);
  up(d_text.p, text.data(), text.size() * 8);
  up(d_toff.p, toff.data(), toff.size() * 8);
  up(d_rl.p, rl.data(), (size_t)ncases * 2);
  up(d_k.p, k, (size_t)ncases * 4);
  if (rc == SQ_OK) {
    k_infix_cases<<<(ncases + 63) / 64, 64>>>(ncases, d_rp.p, d_rn.p, d_rl.p, d_text.p, d_toff.p, d_k.p, d_out.p);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, d_out.p, (size_t)16 * ncases, hipMemcpyDeviceToHost) != hipSuccess) {
      sq_set_error("sq_debug_infix_align: kernel failed: %s", hipGetErrorString(hipGetLastError()));
      rc = SQ_ERR_DEVICE;
    }
  }
  d_rp.free_(); d_rn.free_(); d_text.free_(); d_toff.free_(); d_rl.free_(); d_k.free_(); d_out.free_();
  return rc;
}

static int64_t tap_impl(sq_ctx* c, int what, void* buf, uint64_t cap);
extern "C" int64_t sq_debug_tap(sq_ctx* c, int what, void* buf, uint64_t cap) {
  if (!c || c->owner || !c->api_have) { sq_set_error("sq_debug_tap: no mapped batch"); return SQ_ERR_STATE; }
  return tap_impl(c->last_src ? c->last_src : c, what, buf, cap);   // the lane that mapped the batch last returned
}
static int64_t tap_impl(sq_ctx* c, int what, void* buf, uint64_t cap) {
  if (hipSetDevice(c->device) != hipSuccess) return SQ_ERR_DEVICE;
  const uint32_t n = c->last_n, nrec = c->last_paired ? 2 * n : n;
  std::vector<uint64_t> moff(nrec + 1);
  if (nrec && hipMemcpy(moff.data(), c->mem_off.p, (size_t)(nrec + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) return SQ_ERR_DEVICE;
  const uint64_t* racc = c->idx->ref_accum.data();
  if (what == SQ_TAP_UNIMEMS) {
    std::vector<uint32_t> nu(nrec); std::vector<sq_unimem_dev> um((size_t)nrec * c->uni_slots);
    if (hipMemcpy(nu.data(), c->n_uni.p, (size_t)nrec * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(um.data(), c->unimems.p,
        um.size() * sizeof(sq_unimem_dev),
        hipMemcpyDeviceToHost) != hipSuccess) return SQ_ERR_DEVICE;
    uint64_t cnt = 0; sq_unimem* o = (sq_unimem*)buf;
    for (uint32_t e = 0; e < nrec; ++e) for (uint32_t i = 0; i < nu[e]; ++i) {
      if (o && cnt < cap) {
        const sq_unimem_dev& m = um[(size_t)e * c->uni_slots + i];
        sq_unimem x;
        memset(&x, 0, sizeof(x));
        x.end = e;
        x.qpos = m.qpos;
        x.len = m.len;
        x.unitig = m.unitig;
        x.uoff = m.ustart;
        x.fw = m.fw;
        o[cnt] = x;
      }
      ++cnt;
    }
    return (int64_t)cnt;
  }
  const uint64_t tm = c->last_total_mems;
  std::vector<uint64_t> key(tm), val(tm);
  if (tm) {
    const uint64_t* sk = c->mkey2.p;
    const uint64_t* sv = c->mval2.p;
    if (hipMemcpy(key.data(), sk, tm * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(val.data(), sv, tm * 8,
        hipMemcpyDeviceToHost) != hipSuccess) return SQ_ERR_DEVICE;
  }
  if (what == SQ_TAP_MEMS) {
    sq_mem* o = (sq_mem*)buf;
    for (uint64_t i = 0; i < tm && o && i < cap; ++i) {
      sq_mem x;
      memset(&x, 0, sizeof(x));
      x.end = (uint32_t)(key[i] >> 40);
      x.tid = (uint32_t)(val[i] >> 32);
      x.rpos = (int32_t)((key[i] & ((1ULL << 40) - 1)) - racc[x.tid]);
      x.qpos = (uint16_t)((val[i] >> 10) & 1023);
      x.len = (uint16_t)(val[i] & 1023);
      x.fw = (val[i] >> 20) & 1;
      o[i] = x;
    }
    return (int64_t)tm;
  }
  std::vector<uint32_t> nch(nrec); std::vector<uint64_t> choff(nrec + 1);
  if (nrec && hipMemcpy(nch.data(), c->n_chains.p, (size_t)nrec * 4, hipMemcpyDeviceToHost) != hipSuccess) return SQ_ERR_DEVICE;
  choff = moff;   // chains live in per-end slabs that start at mem_off[e]
  // recovered mates (k_recover) sit past the MEM-count slabs
  const uint64_t tch = nrec ? std::max<uint64_t>(choff[nrec], c->last_chain_slots) : 0;
  std::vector<sq_chain_dev> ch(tch);
  if (tch && hipMemcpy(ch.data(), c->chains.p, tch * sizeof(sq_chain_dev), hipMemcpyDeviceToHost) != hipSuccess) return SQ_ERR_DEVICE;
  if (what == SQ_TAP_CHAINS) {
    uint64_t cnt = 0; sq_chain* o = (sq_chain*)buf;
    for (uint32_t e = 0; e < nrec; ++e) for (uint32_t i = 0; i < nch[e]; ++i) {
      if (o && cnt < cap) {
        const sq_chain_dev& d = ch[choff[e] + i];
        sq_chain x;
        memset(&x, 0, sizeof(x));
        x.end = e;
        x.tid = d.tid;
        x.pos = d.pos;
        x.last_end = d.last_end;
        x.fw = d.fw;
        x.n_mems = d.n_mems;
        x.score = d.score;
        o[cnt] = x;
      }
      ++cnt;
    }
    return (int64_t)cnt;
  }
  if (what == SQ_TAP_CANDIDATES) {
    const uint64_t tc = c->last_total_cands;
    std::vector<sq_cand_dev> cd(tc);
    std::vector<uint64_t> coff(n + 1);
    std::vector<uint32_t> ncd(n);
    if (tc && hipMemcpy(cd.data(), c->cands.p, tc * sizeof(sq_cand_dev), hipMemcpyDeviceToHost) != hipSuccess) return SQ_ERR_DEVICE;
    if (hipMemcpy(coff.data(), c->cand_off.p, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(ncd.data(), c->n_cand.p,
        (size_t)n * 4,
        hipMemcpyDeviceToHost) != hipSuccess) return SQ_ERR_DEVICE;
    std::vector<uint16_t> rl(nrec);
    if (nrec && hipMemcpy(rl.data(), c->rlen.p, (size_t)nrec * 2, hipMemcpyDeviceToHost) != hipSuccess) return SQ_ERR_DEVICE;
    sq_cand* o = (sq_cand*)buf; uint64_t cnt = 0;
    for (uint32_t f = 0; f < n; ++f) for (uint64_t i = coff[f]; i < coff[f] + ncd[f]; ++i) {
      if (o && cnt < cap) { const sq_cand_dev& d = cd[i]; sq_cand x; memset(&x, 0,
          sizeof(x)); x.frag = f; x.tid = d.tid; bool hl = d.lc != 0xFFFFFFFFu,
          hr = d.rc != 0xFFFFFFFFu;
        x.lpos = hl ? ch[d.lc].pos : 0; x.rpos = hr ? ch[d.rc].pos : 0; x.lfw = hl ? ch[d.lc].fw : 0; x.rfw = hr ? ch[d.rc].fw : 0; x.mate_status = d.mate_status;
        const uint32_t e0 = c->last_paired ? 2 * f : f;   // [r5] the scores as the selection saw them (the device no longer writes them back)
        const sqk::CandFinal cf = sqk::cand_final(c->mp, d, rl[e0], c->last_paired ? rl[e0 + 1] : 0);
        x.valid = cf.ok; x.lscore = d.lfail == 2 ? d.lscore : cf.ls; x.rscore = d.lfail == 2 ? d.rscore : cf.rs; x.frag_len = d.frag_len; o[cnt] = x; }
      ++cnt; }
    return (int64_t)cnt;
  }
  if (what == SQ_TAP_PACKED) {   // [r6] what k_pack left (tests hold it to a restatement of the packing rule)
    const uint32_t RW = c->read_words, rec = 1 + RW + RW / 2;
    const uint64_t cnt = (uint64_t)nrec * rec;
    if (!buf) return (int64_t)cnt;
    if (cap < cnt) { sq_set_error("sq_debug_tap: buffer of %llu words for %llu", (unsigned long long)cap, (unsigned long long)cnt); return SQ_ERR_ARG; }
    std::vector<uint64_t> w((size_t)nrec * RW), m((size_t)nrec * (RW / 2)); std::vector<uint16_t> rl(nrec); std::vector<uint8_t> ra(nrec);
    if (nrec && (hipMemcpy(w.data(), c->rpack.p, w.size() * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(m.data(), c->rnmask.p, m.size() * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                 hipMemcpy(rl.data(), c->rlen.p, (size_t)nrec * 2, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(ra.data(), c->rany.p, nrec, hipMemcpyDeviceToHost) != hipSuccess)) return SQ_ERR_DEVICE;
    uint64_t* o = (uint64_t*)buf;
    for (uint32_t e = 0; e < nrec; ++e) {
      uint64_t* r = o + (size_t)e * rec; r[0] = (uint64_t)rl[e] | ((uint64_t)ra[e] << 32);
      memcpy(r + 1, w.data() + (size_t)e * RW, RW * 8); memcpy(r + 1 + RW, m.data() + (size_t)e * (RW / 2), (RW / 2) * 8);
    }
    return (int64_t)cnt;
  }
  sq_set_error("sq_debug_tap: unknown tap %d", what);
  return SQ_ERR_ARG;
}
