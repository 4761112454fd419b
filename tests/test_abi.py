"""The C-ABI library loads and exports every symbol include/salmon_hip.h declares (no GPU needed)."""
import ctypes, os, re
from salmon_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "salmon_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(sq_[a-z0-9_]+)\s*\(", src))
    names -= {"sq_replicate_cb"}
    return sorted(names)


def test_library_loads_and_exports_all_declared_symbols(built):
    L = capi.lib()
    missing = [n for n in _declared() if not hasattr(L, n)]
    assert not missing, "declared in salmon_hip.h but not exported: %s" % missing
    assert not L._missing


def test_struct_sizes_match_header(built):
    # compile a tiny C program that prints sizeof() of every ABI struct and compare with ctypes
    import subprocess, tempfile
    structs = {"sq_index_opts": capi.IndexOpts, "sq_index_view": capi.IndexView, "sq_quant_opts": capi.QuantOpts,
               "sq_read_batch": capi.ReadBatch, "sq_aln": capi.Aln, "sq_aln_batch": capi.AlnBatch, "sq_map_stats": capi.MapStats,
               "sq_eq_table": capi.EqTable, "sq_model_summary": capi.ModelSummary, "sq_em_opts": capi.EmOpts, "sq_txp_in": capi.TxpIn,
               "sq_em_report": capi.EmReport, "sq_gibbs_opts": capi.GibbsOpts, "sq_unimem": capi.UniMem, "sq_mem": capi.Mem,
               "sq_chain": capi.Chain, "sq_cand": capi.Cand}
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        with open(c, "w") as f:
            f.write('#include <stdio.h>\n#include "salmon_hip.h"\nint main(){\n')
            for n in structs:
                f.write('printf("%s %%zu\\n", sizeof(%s));\n' % (n, n))
            f.write("return 0;}\n")
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split("\n")
    got = dict(l.split() for l in out if l.strip())
    for n, cls in structs.items():
        assert int(got[n]) == ctypes.sizeof(cls), n


def test_defaults_match_reference_defaults(built):
    # include/salmon/internal/config/SalmonDefaults.hpp:24-93
    from salmon_amd import api
    o = api.quant_opts()
    assert (o.match_score, o.mismatch_penalty, o.gap_open, o.gap_extend, o.bandwidth) == (2, -4, 6, 2, 15)
    assert abs(o.consensus_slack - 0.35) < 1e-6 and o.min_score_fraction == 0.65
    assert (o.pre_merge_chain_sub_thresh, o.post_merge_chain_sub_thresh, o.orphan_chain_sub_thresh) == (0.75, 0.9, 0.95)
    assert (o.mismatch_seed_skip, o.max_occs_per_hit, o.max_read_occs, o.frag_len_max) == (3, 1000, 200, 1000)
    assert (o.num_pre_burnin_frags, o.num_burnin_frags, o.mini_batch_size) == (5000, 5000000, 5000)
    assert (o.fld_mean, o.fld_sd, o.forgetting_factor, o.range_factorization_bins) == (250.0, 25.0, 0.65, 4)
    e = api.em_opts()
    assert (e.use_vbem, e.per_transcript_prior, e.vb_prior, e.rel_diff_tolerance, e.max_iter, e.min_iter) == (1, 1, 1e-2, 0.01, 10000, 100)


def test_mimic_bt2_presets_match_reference(built):
    # src/util/QuantOptionsUtils.cpp:256-294: both presets raise maxReadOccs to 1000 and consensusSlack to 0.5 and discard orphans; --mimicBT2 then sets
    # Bowtie2-like scores (2 / -4 / 5 / 3), --mimicStrictBT2 RSEM+Bowtie2-like ones (1 / 0 / 25 / 25) with minScoreFraction 0.8; everything else stays
    from salmon_amd import api
    d = api.quant_opts()
    o = api.mimic_bt2(api.quant_opts(hard_filter=1, match_score=7))
    assert (o.max_read_occs, o.allow_orphans, o.match_score, o.mismatch_penalty, o.gap_open, o.gap_extend) == (1000, 0, 2, -4, 5, 3)
    assert abs(o.consensus_slack - 0.5) < 1e-9 and o.min_score_fraction == d.min_score_fraction and o.hard_filter == 1
    s = api.mimic_bt2(api.quant_opts(), strict=True)
    assert (s.max_read_occs, s.allow_orphans, s.match_score, s.mismatch_penalty, s.gap_open, s.gap_extend) == (1000, 0, 1, 0, 25, 25)
    assert abs(s.consensus_slack - 0.5) < 1e-9 and s.min_score_fraction == 0.8
    assert (s.bandwidth, s.max_occs_per_hit, s.range_factorization_bins) == (d.bandwidth, d.max_occs_per_hit, d.range_factorization_bins)
    from salmon_amd import capi
    import ctypes
    assert capi.lib().sq_quant_opts_mimic_bt2(ctypes.byref(s), 2) != 0


def test_no_device_fails_loudly(built):
    # on a box without a GPU the device entry points must refuse, not fall back
    import numpy as np, pytest
    from salmon_amd import api
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    off = np.array([0, 2], np.uint64); tid = np.array([0, 1], np.uint32); w = np.array([0.5, 0.5]); cnt = np.array([3], np.uint64)
    with pytest.raises(capi.SalmonHipError):
        api.em_optimize(api.EqClasses(off, tid, w, cnt), np.array([100.0, 100.0]))
