// host/opts.cpp — defaults mirrored from include/salmon/internal/config/SalmonDefaults.hpp:8-127.
#include <cmath>
#include "../../../include/salmon_hip.h"
#include <string.h>
#include "../../../include/sq_math.h"
extern "C" void sq_quant_opts_default(sq_quant_opts* o) {
  memset(o, 0, sizeof(*o));
  o->lib_type = 1; o->lib_orientation = 2; o->lib_strand = 4;  // -l IU
  o->match_score = 2; o->mismatch_penalty = -4; o->gap_open = 6; o->gap_extend = 2; o->bandwidth = 15;  // :32-36
  o->mismatch_seed_skip = 3; o->max_occs_per_hit = 1000; o->max_read_occs = 200; o->frag_len_max = 1000;  // :37,66,64,58
  o->consensus_slack = 0.35f; o->min_score_fraction = 0.65; o->pre_merge_chain_sub_thresh = 0.75;        // :26-28
  o->post_merge_chain_sub_thresh = 0.9; o->orphan_chain_sub_thresh = 0.95; o->score_exp = 1.0;           // :29-31
  o->decoy_threshold = 1.0; o->min_aln_prob = 1e-5;                                                      // :91,93
  o->hard_filter = 0; o->allow_dovetail = 0; o->allow_orphans = 1; o->disable_chaining_heuristic = 0;
  o->ignore_incompat = 1;  // incompatPrior == 0 -> ignoreIncompat (QuantOptionsUtils.cpp:608-612)
  o->mini_batch_size = 5000; o->num_pre_burnin_frags = 5000; o->num_burnin_frags = 5000000;             // SalmonQuantify.cpp:150; :73-74
  o->fld_mean = 250.0; o->fld_sd = 25.0; o->forgetting_factor = 0.65; o->incompat_prior = 0.0;           // :59-61,16
  o->range_factorization_bins = 4; o->use_frag_len_dist = 1; o->model_single_frag_prob = 1;             // :78
  o->no_length_correction = 0; o->no_eff_length_correction = 0; o->seed = 0x5EED5A1A0ULL;
  o->mini_batches_in_flight = 8; o->seq_bias = 0; o->pos_bias = 0; o->error_model = 0; o->num_error_bins = 6;   /* SalmonDefaults.hpp:124 numErrorBins; the model itself is alignment-mode only and set by the driver */ o->num_bias_samples = 2000000;                                                          // numThreads, SalmonDefaults.hpp:15
}
// --mimicBT2 / --mimicStrictBT2 (QuantOptionsUtils.cpp:256-294; the two flags together are refused by the caller, :250-254)
extern "C" int sq_quant_opts_mimic_bt2(sq_quant_opts* o, int strict) {
  if (!o || (strict != 0 && strict != 1)) return SQ_ERR_ARG;
  o->max_read_occs = 1000; o->consensus_slack = 0.5;                                            // :257-261
  o->allow_orphans = 0;                                                                          // discardOrphansQuasi = true (:266, :287)
  if (!strict) { o->match_score = 2; o->mismatch_penalty = -4; o->gap_open = 5; o->gap_extend = 3; }   // :272-275
  else { o->min_score_fraction = 0.8; o->match_score = 1; o->mismatch_penalty = 0; o->gap_open = 25; o->gap_extend = 25; }   // :288-292
  return SQ_OK;
}
extern "C" void sq_em_opts_default(sq_em_opts* o) {
  memset(o, 0, sizeof(*o));
  o->use_vbem = 1; o->per_transcript_prior = 1; o->init_uniform = 0; o->eq_class_mode = 0; o->no_rich_eq_classes = 0;  // :76,87,63
  // :85; MappingPipelineStages.cpp:46-49
  o->vb_prior = 1e-2;
  o->rel_diff_tolerance = 0.01;
  o->max_iter = 10000;
  o->min_iter = 100;
  o->num_required_fragments = 50000000.0;                                                                               // :110
}

// LibraryTypeDetector::mostLikelyType (include/salmon/internal/model/LibraryTypeDetector.hpp:33-152): the most likely library format
// from the per-format sample counts (index = formatID: type | orientation << 1 | strandedness << 3; strandedness SA 0, AS 1, S 2, A 3, U 4;
// orientation SAME 0, AWAY 1, TOWARD 2, NONE 3).
void sq_detect_lib_format(uint8_t type, const uint64_t* counts64, uint8_t* out_type, uint8_t* out_orient, uint8_t* out_strand) {
  *out_type = type;
  if (type == 0) {                          // single end
    uint64_t nf = 0, nr = 0;
    for (int i = 0; i < 64; ++i) { const int st = i >> 3; nf += (st == 2) ? counts64[i] : 0; nr += (st == 3) ? counts64[i] : 0; }
    const double ratio = (nf + nr > 0) ? (double)nf / (double)(nf + nr) : -1.0;
    *out_orient = 3;
    *out_strand = ratio < 0.0 ? 4 : (ratio < 0.3 ? 3 : (ratio < 0.7 ? 4 : 2));
    return;
  }
  uint64_t nsf = 0, nsr = 0, nin = 0, nout = 0, nsame = 0;
  for (int i = 0; i < 64; ++i) {
    const int orient = (i >> 1) & 3, st = i >> 3; const uint64_t c = counts64[i];
    nsf += (st == 2 || st == 0) ? c : 0; nsr += (st == 3 || st == 1) ? c : 0;
    nin += (orient == 2) ? c : 0; nout += (orient == 1) ? c : 0; nsame += (orient == 0) ? c : 0;
  }
  if (nin + nout + nsame > 0 && nsf + nsr > 0) {
    const uint64_t no = nin + nout + nsame;
    const double rin = (double)nin / (double)no, rout = (double)nout / (double)no, rsame = (double)nsame / (double)no;
    bool same = false;
    if (rin >= rout && rin >= rsame) *out_orient = 2; else if (rout >= rin && rout >= rsame) *out_orient = 1; else { *out_orient = 0; same = true; }
    const double rfw = (double)nsf / (double)(nsf + nsr);
    if (rfw < 0.3) *out_strand = same ? 3 : 1; else if (rfw < 0.7) *out_strand = 4; else *out_strand = same ? 2 : 0;
  } else { *out_orient = 2; *out_strand = 4; }
}

// SPEC §MG: masses of R ranks (row r = rank r's log-masses, LOG_0 = +inf where a transcript has none) -> logAdd in rank order
extern "C" int sq_merge_log_masses(uint32_t M, uint32_t R, const double* all_log_mass, double* out) {
  if (!all_log_mass || !out || R == 0) return SQ_ERR_ARG;
  for (uint32_t t = 0; t < M; ++t) {
    double m = SQ_LOG_0;
    for (uint32_t r = 0; r < R; ++r) m = sq_log_add(m, all_log_mass[(size_t)r * M + t]);
    out[t] = m;
  }
  return SQ_OK;
}

// ForgettingMassCalculator (include/salmon/internal/quant/ForgettingMassCalculator.hpp:23-40,64-90): log forgetting mass of mini-batch b,
// fm_0 = 0, fm_b = fm_{b-1} + ff log(b) - log((b+1)^ff - 1).  The online stage (hip/online.hip) takes its schedule from here.
extern "C" int sq_forgetting_masses(double ff, uint64_t n, double* out) {
  if (!out && n) return SQ_ERR_ARG;
  double fm = 0.0;
  for (uint64_t b = 0; b < n; ++b) {
    if (b > 0) { const uint64_t i = b + 1; fm += ff * std::log((double)(i - 1)) - std::log(std::pow((double)i, ff) - 1.0); }   // prefill's `fm += a - b`: the increment first
    out[b] = fm;
  }
  return SQ_OK;
}
