"""ctypes view of include/salmon_hip.h (the C ABI of libsalmon_hip.so).

This is plumbing: struct layouts and prototypes only.  The product has NO CPU fallback — loading
fails loudly if the HIP library has not been built (run `python -m salmon_amd.build`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsalmon_hip.so")

u8, u16, u32, u64, i32, f64 = C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_int32, C.c_double
P = C.POINTER


class IndexOpts(C.Structure):
    _fields_ = [("k", u32), ("m", u32), ("keep_duplicates", u32), ("no_clip_polya", u32), ("threads", u32), ("gencode", u32)]


class IndexView(C.Structure):
    _fields_ = [("k", u32), ("m", u32), ("num_refs", u32), ("first_decoy", u32), ("num_unitigs", u64), ("total_unitig_nt", u64),
                ("num_kmers", u64), ("total_ref_nt", u64), ("num_occ", u64), ("ref_accum", P(u64)), ("ref_len", P(u32)),
                ("ref_clen", P(u32)), ("refseq", P(u64)), ("useq", P(u64)), ("uoff", P(u64)), ("ctab_off", P(u64)), ("ctab", P(u64))]


class QuantOpts(C.Structure):
    _fields_ = [("lib_type", u8), ("lib_orientation", u8), ("lib_strand", u8), ("_pad0", u8),
                ("match_score", i32), ("mismatch_penalty", i32), ("gap_open", i32), ("gap_extend", i32), ("bandwidth", i32),
                ("mismatch_seed_skip", u32), ("max_occs_per_hit", u32), ("max_read_occs", u32), ("frag_len_max", u32),
                ("consensus_slack", f64), ("min_score_fraction", f64), ("pre_merge_chain_sub_thresh", f64),
                ("post_merge_chain_sub_thresh", f64), ("orphan_chain_sub_thresh", f64), ("score_exp", f64),
                ("decoy_threshold", f64), ("min_aln_prob", f64),
                ("hard_filter", u8), ("allow_dovetail", u8), ("allow_orphans", u8), ("disable_chaining_heuristic", u8),
                ("ignore_incompat", u8), ("recover_orphans", u8), ("lib_autodetect", u8), ("gc_bias", u8),
                ("mini_batch_size", u32), ("num_pre_burnin_frags", u32), ("num_burnin_frags", u64),
                ("fld_mean", f64), ("fld_sd", f64), ("forgetting_factor", f64), ("incompat_prior", f64),
                ("range_factorization_bins", u32), ("use_frag_len_dist", u8), ("model_single_frag_prob", u8),
                ("no_length_correction", u8), ("no_eff_length_correction", u8), ("seed", u64),
                ("seq_bias", u8), ("pos_bias", u8), ("error_model", u8), ("num_error_bins", u8), ("num_bias_samples", u32), ("mini_batches_in_flight", u32)]


class ReadBatch(C.Structure):
    _fields_ = [("n", u32), ("paired", u32), ("seq", C.c_void_p), ("seq_off", C.c_void_p), ("on_device", C.c_int)]


class Aln(C.Structure):
    _fields_ = [("tid", u32), ("pos", i32), ("mate_pos", i32), ("score", i32), ("mate_score", i32), ("frag_len", u32),
                ("read_len", u16), ("mate_len", u16), ("fwd", u8), ("mate_fwd", u8), ("mate_status", u8), ("format_id", u8),
                ("est_aln_prob", f64)]


class AlnBatch(C.Structure):
    _fields_ = [("n", u32), ("read_off", P(u64)), ("aln", P(Aln)), ("aln_cap", u64), ("map_type", P(u8))]


class AlnReads(C.Structure):   # sq_aln_reads
    _fields_ = [("num_alignments", u64), ("cig_off", P(u64)), ("cigar", P(u32)), ("seq_off", P(u64)), ("seq", P(u8)), ("pos", P(C.c_int32)), ("aligner_score", P(C.c_int32))]


class MapStats(C.Structure):
    _names = ["num_reads", "num_mapped_at_least_a_kmer", "num_with_joint_hits", "num_mapped", "num_alignments",
              "num_mappings_filtered", "num_fragments_filtered", "num_dovetails", "num_decoy_fragments", "num_seeds",
              "num_lookups", "num_mems", "num_chains", "num_candidates", "num_dp_alignments", "num_orphans_rescued",
              "num_truncated_ends"]
    _fields_ = [(n, u64) for n in _names]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in self._names}


class EqTable(C.Structure):
    _fields_ = [("num_classes", u64), ("num_labels", u64), ("off", P(u64)), ("tid", P(u32)), ("w", P(f64)), ("wq", P(u64)),
                ("count", P(u64)), ("bins", P(u32)), ("h1", P(u64)), ("h2", P(u64))]


class ModelSummary(C.Structure):
    _fields_ = [("num_observed", u64), ("num_assigned", u64), ("num_mapped_ub", u64), ("burned_in", C.c_int), ("num_compatible", u64),
                ("lib_format_id", u32), ("lib_detected", u32)]


class EmOpts(C.Structure):
    _fields_ = [("use_vbem", u8), ("per_transcript_prior", u8), ("init_uniform", u8), ("eq_class_mode", u8),
                ("no_rich_eq_classes", u8), ("alt_init_mode", u8), ("_pad", u8 * 2), ("vb_prior", f64), ("rel_diff_tolerance", f64),
                ("max_iter", u32), ("min_iter", u32), ("num_required_fragments", f64)]


class TxpIn(C.Structure):
    _fields_ = [("num_txp", u32), ("projected_counts", P(f64)), ("unique_count", P(u64)), ("eff_len", P(f64))]


class EmReport(C.Structure):
    _fields_ = [("iters", u32), ("converged", C.c_int), ("max_rel_diff", f64), ("alpha_sum", f64), ("device_ms", f64),
                ("ms_per_iter", f64), ("num_degenerate", u32), ("_pad", u32)]


class BiasModels(C.Structure):   # sq_bias_models
    _fields_ = [("gc_observed", C.c_void_p), ("seq_fw", C.c_void_p), ("seq_rc", C.c_void_p), ("pos_observed", C.c_void_p), ("threads", C.c_uint32), ("_pad", C.c_uint32)]


class BiasReport(C.Structure):
    _fields_ = [("num_processed", u32), ("fld_low", i32), ("fld_high", i32), ("_pad", u32), ("gc_bias_row0", f64 * 25)]


EFFLEN_CB = C.CFUNCTYPE(C.c_int, P(f64), P(f64), P(f64), u32, C.c_void_p)


class GibbsOpts(C.Structure):
    _fields_ = [("thinning_factor", u32), ("no_gamma_draw", u8), ("use_vbem", u8), ("per_transcript_prior", u8), ("_pad", u8),
                ("vb_prior", f64)]


class GibbsReport(C.Structure):
    _fields_ = [("rounds", u64), ("device_ms", f64), ("ms_per_round", f64), ("draws_per_round", u64), ("items", u32 * 3), ("_pad", u32)]


class UniMem(C.Structure):
    _fields_ = [("end", u32), ("qpos", u16), ("len", u16), ("unitig", u64), ("uoff", u32), ("fw", u8), ("_p", u8 * 3)]


class Mem(C.Structure):
    _fields_ = [("end", u32), ("tid", u32), ("rpos", i32), ("qpos", u16), ("len", u16), ("fw", u8), ("_p", u8 * 3)]


class Chain(C.Structure):
    _fields_ = [("end", u32), ("tid", u32), ("pos", i32), ("last_end", i32), ("fw", u8), ("_p", u8 * 3), ("n_mems", u32), ("score", f64)]


class Cand(C.Structure):
    _fields_ = [("frag", u32), ("tid", u32), ("lpos", i32), ("rpos", i32), ("lfw", u8), ("rfw", u8), ("mate_status", u8),
                ("valid", u8), ("lscore", i32), ("rscore", i32), ("frag_len", u32)]


REPLICATE_CB = C.CFUNCTYPE(C.c_int, P(f64), u32, C.c_void_p)

_lib = None


class SamCounts(C.Structure):   # sq_sam_counts
    _fields_ = [(n, C.c_uint64) for n in ("num_records", "num_fragments", "num_alignments", "num_unaligned", "num_suspicious_pairs", "num_skipped_unknown_target", "num_frags_without_as")]


class MetaInfo(C.Structure):   # sq_meta_info
    _fields_ = [("salmon_version", C.c_char_p), ("samp_type", C.c_char_p), ("opt_type", C.c_char_p), ("quant_errors", C.c_char_p),
                ("num_libraries", C.c_uint32), ("frag_dist_length", C.c_uint32), ("library_types", C.POINTER(C.c_char_p)),
                ("frag_length_mean", C.c_double), ("frag_length_sd", C.c_double),
                ("seq_bias_correct", C.c_int32), ("gc_bias_correct", C.c_int32), ("pos_bias_correct", C.c_int32), ("num_bias_bins", C.c_uint32),
                ("mapping_type", C.c_char_p), ("keep_duplicates", C.c_int32), ("serialized_eq_classes", C.c_int32), ("range_factorized", C.c_int32), ("scalar_weights", C.c_int32),
                ("num_valid_targets", C.c_uint64), ("num_decoy_targets", C.c_uint64), ("num_eq_classes", C.c_uint64),
                ("num_length_classes", C.c_uint32), ("_pad0", C.c_uint32), ("length_classes", C.POINTER(C.c_uint32)),
                ("index_seq_hash", C.c_char_p), ("index_name_hash", C.c_char_p), ("index_seq_hash512", C.c_char_p), ("index_name_hash512", C.c_char_p),
                ("index_decoy_seq_hash", C.c_char_p), ("index_decoy_name_hash", C.c_char_p),
                ("num_bootstraps", C.c_uint64), ("num_processed", C.c_uint64), ("num_mapped", C.c_uint64), ("num_decoy_fragments", C.c_uint64),
                ("num_dovetail_fragments", C.c_uint64), ("num_fragments_filtered_vm", C.c_uint64), ("num_alignments_below_threshold_vm", C.c_uint64),
                ("percent_mapped", C.c_double), ("start_time", C.c_char_p), ("end_time", C.c_char_p),
                ("backend", C.c_char_p), ("num_em_iterations", C.c_uint32), ("num_degenerate_eq_classes", C.c_uint32), ("runtime_s", C.c_double)]


def lib():
    """Load libsalmon_hip.so (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("salmon_amd: %s is missing — build it with `python -m salmon_amd.build` "
                           "(the HIP extension is the product; there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    sig = {
        "sq_last_error": (C.c_char_p, []), "sq_version": (C.c_char_p, []),
        "sq_index_build": (C.c_int, [P(IndexOpts), C.c_char_p, C.c_char_p, C.c_char_p]),
        "sq_ctx_seed_filter_fills": (u64, [vp, C.c_int]), "sq_debug_bgzf_inflate": (C.c_int, [C.c_int, vp, u64, vp, u32, vp, u64, vp]), "sq_debug_inflate_core_host": (C.c_int, [vp, u64, vp, u32, vp]),
        "sq_debug_gzip_inflate": (C.c_int, [C.c_int, vp, u64, u64, vp, u64, vp, vp]), "sq_debug_inflate_span_host": (C.c_int, [vp, u64, u64, u64, vp, u32, vp, vp, vp]),
        "sq_debug_find_block_start_host": (u64, [vp, u64, u64, u64]),
        "sq_index_build_fasta_mem": (C.c_int, [P(IndexOpts), C.c_char_p, C.c_char_p, P(vp)]),
        "sq_index_build_mem": (C.c_int, [P(IndexOpts), u32, P(C.c_char_p), P(C.c_char_p), P(u32), u32, C.c_char_p, P(vp)]),
        "sq_index_load": (C.c_int, [C.c_char_p, C.c_int, P(vp)]), "sq_index_build_set_device": (C.c_int, [C.c_int]),
        "sq_index_to_device": (C.c_int, [vp, C.c_int]), "sq_index_free": (None, [vp]),
        "sq_index_k": (u32, [vp]), "sq_index_m": (u32, [vp]), "sq_index_num_refs": (u32, [vp]), "sq_index_first_decoy": (u32, [vp]),
        "sq_index_ref_name": (C.c_char_p, [vp, u32]), "sq_index_ref_len": (u32, [vp, u32]), "sq_index_ref_complete_len": (u32, [vp, u32]),
        "sq_index_is_decoy": (C.c_int, [vp, u32]), "sq_index_num_unitigs": (u64, [vp]), "sq_index_num_kmers": (u64, [vp]),
        "sq_index_device_bytes": (u64, [vp]), "sq_index_get_view": (C.c_int, [vp, P(IndexView)]),
        "sq_index_lookup_host": (C.c_int, [vp, u64, P(u64), P(u32), P(C.c_int)]),
        "sq_quant_opts_default": (None, [P(QuantOpts)]), "sq_quant_opts_mimic_bt2": (C.c_int, [P(QuantOpts), C.c_int]), "sq_em_opts_default": (None, [P(EmOpts)]),
        "sq_ctx_create": (C.c_int, [vp, P(QuantOpts), C.c_int, u32, P(vp)]), "sq_ctx_free": (None, [vp]), "sq_ctx_reset": (C.c_int, [vp]),
        "sq_map_batch": (C.c_int, [vp, P(ReadBatch), P(AlnBatch), P(MapStats)]),
        "sq_eq_export_device": (C.c_int, [vp, P(EqTable)]), "sq_eq_merge_device": (C.c_int, [vp, P(EqTable)]),
        "sq_map_submit": (C.c_int, [vp, P(ReadBatch), P(AlnBatch)]), "sq_ctx_set_lanes": (C.c_int, [vp, C.c_int]),
        "sq_reader_open": (C.c_int, [P(C.c_char_p), u32, P(C.c_char_p), u32, u32, u32, P(vp)]), "sq_reader_next": (C.c_int, [vp, P(ReadBatch),
            P(C.c_int)]),
        "sq_reader_open_ex": (C.c_int, [P(C.c_char_p), u32, P(C.c_char_p), u32, u32, u32, u32, P(vp)]),
        "sq_reader_names": (C.c_int, [vp, C.c_int, P(C.c_void_p), P(P(u64))]), "sq_map_fetch": (C.c_int, [vp, P(AlnBatch)]),
        "sq_reader_release": (None, [vp, C.c_int]), "sq_reader_total": (u64, [vp]), "sq_reader_close": (None, [vp]),
        "sq_map_wait": (C.c_int, [vp, P(AlnBatch), P(MapStats)]),
        "sq_eq_accumulate": (C.c_int, [vp]), "sq_eq_finish": (C.c_int, [vp, P(EqTable)]), "sq_eq_merge": (C.c_int, [vp, P(EqTable)]),
        "sq_model_summary_get": (C.c_int, [vp, P(ModelSummary)]),
        "sq_model_fetch": (C.c_int, [vp, P(f64), P(u64), P(u64), P(f64)]), "sq_model_fetch_fld": (C.c_int, [vp, P(f64)]),
        "sq_em_optimize": (C.c_int, [vp, P(EqTable), P(TxpIn), P(EmOpts), P(f64), P(EmReport)]),
        "sq_em_optimize_dev": (C.c_int, [C.c_int, P(EqTable), P(TxpIn), P(EmOpts), P(f64), P(EmReport)]),
        "sq_em_steps_dev": (C.c_int, [C.c_int, P(EqTable), P(TxpIn), P(EmOpts), P(f64), u32, P(f64), P(EmReport)]),
        "sq_bootstrap_dev": (C.c_int, [C.c_int, P(EqTable), P(TxpIn), P(EmOpts), u32, u64, u64, REPLICATE_CB, vp]),
        "sq_gibbs_dev": (C.c_int, [C.c_int, P(EqTable), P(TxpIn), P(GibbsOpts), P(f64), u32, u64, u64, REPLICATE_CB, vp]),
        "sq_bootstrap_range_dev": (C.c_int, [C.c_int, P(EqTable), P(TxpIn), P(EmOpts), u32, u32, u32, u64, u64, REPLICATE_CB, vp]),
        "sq_gibbs_range_dev": (C.c_int, [C.c_int, P(EqTable), P(TxpIn), P(GibbsOpts), P(f64), u32, u32, u32, u64, u64, REPLICATE_CB, vp]),
        "sq_gibbs_range_report_dev": (C.c_int, [C.c_int, P(EqTable), P(TxpIn), P(GibbsOpts), P(f64), u32, u32, u32, u64, u64, REPLICATE_CB, vp, P(GibbsReport)]),
        "sq_gibbs_chain_step": (u32, [u32]),
        "sq_merge_log_masses": (C.c_int, [u32, u32, vp, vp]),
        "sq_forgetting_masses": (C.c_int, [f64, u64, vp]),
        "sq_model_fetch_gc_observed": (C.c_int, [vp, vp]),
        "sq_bias_gc_eff_lengths": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, P(BiasReport)]),
        "sq_model_fetch_seq_observed": (C.c_int, [vp, vp, vp, P(u64)]),
        "sq_model_fetch_pos_observed": (C.c_int, [vp, vp]), "sq_model_fetch_error_model": (C.c_int, [vp, vp, vp, P(u32)]),
        "sq_bias_eff_lengths": (C.c_int, [vp, P(BiasModels), vp, u32, vp, vp, vp, vp, vp, P(BiasReport)]),
        "sq_index_length_classes": (C.c_int, [vp, vp, vp]),
        "sq_bias_seq_eff_lengths": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, u32, vp, vp, vp, vp, P(BiasReport)]),
        "sq_em_optimize_bias": (C.c_int, [vp, P(EqTable), P(TxpIn), P(EmOpts), EFFLEN_CB, vp, P(f64), P(f64), P(EmReport)]),
        "sq_dist_make_id": (C.c_int, [vp]), "sq_dist_init": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, P(vp)]), "sq_dist_free": (None, [vp]),
        "sq_dist_rank": (C.c_int, [vp]), "sq_dist_world": (C.c_int, [vp]), "sq_dist_merge_eq": (C.c_int, [vp, vp]),
        "sq_dist_merge_eq_loopback": (C.c_int, [vp, P(vp), u32]),
        "sq_dist_reduce_model": (C.c_int, [vp, u32, P(f64), P(u64), P(u64), P(f64)]), "sq_dist_allreduce_u64": (C.c_int, [vp, vp, C.c_size_t]),
        "sq_dist_bcast": (C.c_int, [vp, vp, C.c_size_t, C.c_int]), "sq_dist_allgather": (C.c_int, [vp, vp, C.c_size_t, vp]),
        "sq_dist_barrier": (C.c_int, [vp]), "sq_dist_share": (None, [vp, u32, u32, P(u32), P(u32)]),
        "sq_debug_tap": (C.c_int64, [vp, C.c_int, vp, u64]),
        "sq_ctx_reserve": (C.c_int, [vp, u64, u64]),
        "sq_debug_infix_align": (C.c_int, [C.c_int, u32, vp, vp, vp, vp, vp, vp]),
        "sq_normalize_alphas": (C.c_int, [u32, P(EqTable), P(f64), P(u64), P(u64), P(f64)]),
        "sq_write_quant_sf": (C.c_int, [C.c_char_p, vp, P(f64), P(f64), f64]), "sq_write_eq_classes": (C.c_int, [C.c_char_p, vp, P(EqTable),
            C.c_int]),
        "sq_model_fetch_lib_counts": (C.c_int, [vp, P(u64)]), "sq_write_lib_format_counts": (C.c_int, [C.c_char_p, C.c_char_p, u8, u8, u8, P(u64),
            u64, u64]),
        "sq_write_ambig_info": (C.c_int, [C.c_char_p, u32, P(EqTable)]),
        "sq_write_quant_sf_digits": (C.c_int, [C.c_char_p, vp, P(f64), P(f64), f64, C.c_int]),
        "sq_write_quant_sf_names": (C.c_int, [C.c_char_p, u32, P(C.c_char_p), P(u32), P(f64), P(f64), f64]),
        "sq_eq_file_read": (C.c_int, [C.c_char_p, P(vp)]), "sq_eq_file_free": (None, [vp]), "sq_eq_file_num_txp": (u32,
            [vp]), "sq_eq_file_name": (C.c_char_p, [vp, u32]),
        "sq_eq_file_eff_lens": (P(f64), [vp]), "sq_eq_file_table": (C.c_int, [vp, P(EqTable)]),
        "sq_boot_writer_open": (C.c_int, [C.c_char_p, u32, P(C.c_char_p), P(vp)]), "sq_boot_writer_append": (C.c_int, [vp, P(f64),
            u32]), "sq_boot_writer_close": (u64, [vp]),
        "sq_bias_last_gc_expected": (C.c_int, [vp]), "sq_aln_inject": (C.c_int, [vp, P(AlnBatch), u64]), "sq_model_drop_counts": (C.c_int, [vp]),
        "sq_sam_open": (C.c_int, [C.c_char_p, C.c_int, P(vp)]), "sq_sam_first_flag": (C.c_int, [C.c_char_p, P(C.c_int)]), "sq_sam_num_refs": (u32, [vp]), "sq_sam_ref_name": (C.c_char_p, [vp, u32]), "sq_sam_ref_len": (u32, [vp, u32]),
        "sq_sam_set_tid_map": (C.c_int, [vp, vp, u32]), "sq_sam_keep_reads": (C.c_int, [vp, C.c_int]), "sq_sam_reads": (C.c_int, [vp, P(AlnReads)]),
        "sq_aln_inject_reads": (C.c_int, [vp, P(AlnBatch), P(AlnReads), u64]), "sq_sam_next": (C.c_int, [vp, u32, C.c_int, f64, P(AlnBatch), P(SamCounts)]), "sq_sam_close": (None, [vp]),
        "sq_index_hash": (C.c_char_p, [vp, C.c_int]), "sq_index_keeps_duplicates": (C.c_int, [vp]), "sq_model_fld_min": (C.c_int, [vp, P(u32)]),
        "sq_write_fld_samples": (C.c_int, [C.c_char_p, vp, u32, u32, u32, u64, P(f64), P(f64), P(u32)]), "sq_write_legacy_bias": (C.c_int, [C.c_char_p, P(u32)]),
        "sq_write_gc_model": (C.c_int, [C.c_char_p, C.c_int32, u32, u32, vp, vp]), "sq_write_seq_model": (C.c_int, [C.c_char_p, vp]),
        "sq_write_pos_models": (C.c_int, [C.c_char_p, u32, vp, u32, vp]), "sq_write_meta_info": (C.c_int, [C.c_char_p, P(MetaInfo)]),
        "sq_ctx_set_profiling": (C.c_int, [vp, C.c_int]), "sq_ctx_num_stages": (C.c_int, []), "sq_ctx_stage_name": (C.c_char_p, [C.c_int]),
        "sq_ctx_stage_times": (C.c_int, [vp, P(f64), P(u64), C.c_int]),
    }
    missing = []
    for name, (res, args) in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    L._missing = missing
    L._declared = sorted(sig)
    _lib = L
    return L


class SalmonHipError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        raise SalmonHipError("%s failed (%d): %s" % (what, rc, lib().sq_last_error().decode()))
