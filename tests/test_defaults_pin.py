"""Option defaults and log-space helpers against the reference's own header-only files, compiled from where they lie under
/root/reference into oracle/_ref/libdefaults_ref.so (oracle/ref_defaults_shim.cpp; `make -C oracle ref`).  Skipped where the
reference is not present (the GPU box only carries the built library: it is used there if it travelled)."""
import ctypes as C, math, os, re
import numpy as np
import pytest
from salmon_amd import api
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libdefaults_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libdefaults_ref.so not built (no /root/reference on this machine)")
    L = C.CDLL(path)
    L.ref_default.argtypes = [C.c_char_p, C.POINTER(C.c_double)]; L.ref_default.restype = C.c_int
    L.ref_default_aux_dir.restype = C.c_char_p
    for f in (L.ref_log_add, L.ref_log_sub): f.argtypes = [C.c_double, C.c_double]; f.restype = C.c_double
    L.ref_math_const.argtypes = [C.c_int]; L.ref_math_const.restype = C.c_double
    return L


def _d(L, name):
    v = C.c_double(); assert L.ref_default(name.encode(), C.byref(v)) == 1, name
    return v.value


def test_quant_and_em_option_defaults_are_the_references(built):
    L = _ref(); q = api.quant_opts(); e = api.em_opts()
    pairs = [(q.match_score, "matchScore"), (q.mismatch_penalty, "mismatchPenalty"), (q.gap_open, "gapOpenPenalty"), (q.gap_extend, "gapExtendPenalty"),
             (q.bandwidth, "dpBandwidth"), (q.mismatch_seed_skip, "mismatchSeedSkip"), (q.max_occs_per_hit, "maxOccsPerHit"), (q.max_read_occs, "maxReadOccs"),
             (q.frag_len_max, "maxFragLength"), (q.consensus_slack, "consensusSlack"), (q.min_score_fraction, "minScoreFraction"),
             (q.pre_merge_chain_sub_thresh, "pre_merge_chain_sub_thresh"), (q.post_merge_chain_sub_thresh, "post_merge_chain_sub_thresh"),
             (q.orphan_chain_sub_thresh, "orphan_chain_sub_thresh"), (q.score_exp, "scoreExp"), (q.decoy_threshold, "decoyThreshold"),
             (q.min_aln_prob, "minAlnProb"), (q.hard_filter, "hardFilter"), (q.allow_dovetail, "allowDovetail"), (q.recover_orphans, "recoverOrphans"),
             (q.disable_chaining_heuristic, "disableChainingHeuristic"), (q.num_pre_burnin_frags, "numPreBurninFrags"), (q.num_burnin_frags, "numBurninFrags"),
             (q.fld_mean, "fragLenPriorMean"), (q.fld_sd, "fragLenPriorSD"), (q.forgetting_factor, "ffactor"), (q.incompat_prior, "incompatPrior"),
             (q.range_factorization_bins, "rangeFactorizationBins"), (q.no_length_correction, "noLengthCorrection"),
             (q.no_eff_length_correction, "noEffectiveLengthCorrection"), (q.mini_batches_in_flight, "numThreads"),
             (e.use_vbem, "useVBOpt"), (e.per_transcript_prior, "perTranscriptPrior"), (e.vb_prior, "vbPrior"), (e.init_uniform, "initUniform"),
             (e.alt_init_mode, "alternativeInitMode")]
    for mine, name in pairs:
        assert float(mine) == _d(L, name), name                       # consensusSlack is a float in the reference: 0.35f, not 0.35
    assert q.allow_orphans == 1 - int(_d(L, "discardOrphansQuasi")) and q.use_frag_len_dist == 1 - int(_d(L, "noFragLengthDist"))
    assert q.model_single_frag_prob == 1 - int(_d(L, "noSingleFragProb")) and _d(L, "useEM") == 0.0 and _d(L, "validateMappings") == 1.0


def test_cli_and_bias_constants_are_the_references(built):
    # constants the stand-alone driver and the bias code carry as literals: looked up in the sources next to their citations
    L = _ref()
    cli = open(os.path.join(ROOT, "salmon_amd", "csrc", "cli", "salmon_main.cpp")).read()
    assert int(re.search(r'"--minAssignedFrags"\)\) \? strtoull\(v, nullptr, 10\) : (\d+)', cli).group(1)) == int(_d(L, "minAssignedFrags"))
    assert int(re.search(r'"--sigDigits"\)\) \? .*? : (\d+);', cli).group(1)) == int(_d(L, "sigDigits"))
    assert int(re.search(r'"--thinningFactor"\)\) \? \(uint32_t\)atoi\(v\) : (\d+)', cli).group(1)) == int(_d(L, "thinningFactor"))
    assert L.ref_default_aux_dir().decode() == re.search(r'"--auxDir"\)\) \? v : "([a-z_]+)"', cli).group(1)
    hdr = open(os.path.join(ROOT, "salmon_amd", "csrc", "sq_internal.h")).read()
    assert int(re.search(r"#define SQ_GC_FRAG_BINS (\d+)", hdr).group(1)) == int(_d(L, "numFragGCBins"))
    assert int(re.search(r"#define SQ_GC_COND_BINS (\d+)", hdr).group(1)) == int(_d(L, "numConditionalGCBins"))


def test_log_space_helpers_follow_the_references(built):
    # include/sq_math.h evaluates log / exp with its own fixed operation sequences (bit-identical on CPU and GPU), so against the
    # reference's std::log / std::exp form the agreement is to a few ulp; the special cases (LOG_0 operands, the swap) are exact
    L = _ref(); O = orc.lib()
    assert L.ref_math_const(0) == math.inf and L.ref_math_const(1) == 0.0
    assert abs(L.ref_math_const(5) - O.orc_log(L.ref_math_const(4))) <= 4e-15 * abs(L.ref_math_const(5))     # LOG_EPSILON = log(0.375e-10)
    rng = np.random.default_rng(3)
    for x, y in zip(rng.uniform(-700, 50, 4000), rng.uniform(-700, 50, 4000)):
        r = L.ref_log_add(x, y); m = O.orc_log_add(x, y)
        assert abs(m - r) <= 8e-16 * max(1.0, abs(r))
    for x in (-3.5, 0.0, 12.0):
        assert O.orc_log_add(math.inf, x) == L.ref_log_add(math.inf, x) == x and O.orc_log_add(x, math.inf) == L.ref_log_add(x, math.inf) == x
        assert O.orc_log_add(x, -800.0) == x == L.ref_log_add(x, -800.0)


def test_forgetting_mass_schedule_is_the_reference_calculators(built):
    """ForgettingMassCalculator.hpp compiled from the reference tree (prefill + getLogMassAndTimestep, as SalmonQuantify.cpp uses it) against the
    schedule the online stage takes (sq_forgetting_masses) and the checker's (orc_forgetting_mass): the same libm calls in the same order."""
    L = _ref(); L.ref_forgetting_masses.argtypes = [C.c_double, C.c_uint32, C.c_void_p]
    from salmon_amd import capi
    n = 50000
    for ff in (0.65, 0.9):
        ref = np.zeros(n); L.ref_forgetting_masses(ff, n, ref.ctypes.data)
        mine = np.zeros(n); capi.check(capi.lib().sq_forgetting_masses(ff, n, mine.ctypes.data), "sq_forgetting_masses")
        assert ref[0] == 0.0 and np.array_equal(ref, mine)
        for b in (0, 1, 2, 17, 4999, n - 1):
            assert orc.lib().orc_forgetting_mass(ff, b) == ref[b]
