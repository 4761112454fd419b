"""[r5] Rows a7 / a8 — which of a fragment's scored candidates become its alignments, and with what probability — pinned to the reference's own code:
`MappingScoreInfo`, `updateRefMappings`, `haveOnlyDecoyMappings`, `filterAndCollectAlignments` of include/salmon/internal/quant/SalmonMappingUtils.hpp, compiled
where they lie with stand-in pufferfish types (oracle/ref_mapping_utils_shim.cpp -> oracle/_ref/libmappingutils_ref.so) and driven by the call site's loop
(src/quant/SalmonQuantify.cpp:1457-1631).  The checker's select_hits (what the kernels k_finalize / k_select are held to on the GPU) must pick the same
candidates, in the same order, with estAlnProb within an ulp (libm's exp against the shared one), the same best / best-decoy scores and the same "only decoys" verdict — over ties, duplicates of a
transcript, decoys before and behind the hits they cut off, failed alignments, skipped (incompatible) candidates, hard filtering and every threshold.
Also shown: what the call site's un-advanced slot index does when an incompatible candidate is skipped (SPEC section a7: a reference quirk that is not followed).
No GPU."""
import ctypes as C, os
import numpy as np
import pytest
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libmappingutils_ref.so")
INVALID = -2 ** 31


def _ref():
    if not os.path.exists(REF): pytest.skip("oracle/_ref/libmappingutils_ref.so is built where /root/reference exists (make -C oracle ref)")
    L = C.CDLL(REF); L.ref_select_hits.restype = C.c_uint32
    L.ref_select_hits.argtypes = [C.c_uint32] + [C.c_void_p] * 5 + [C.c_uint32, C.c_uint32, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int] + [C.c_void_p] * 4
    return L


def _checker():
    L = orc.lib(); L.orc_select_hits.restype = C.c_uint32
    L.orc_select_hits.argtypes = [C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint32, C.c_double, C.c_int, C.c_double, C.c_double] + [C.c_void_p] * 3
    return L


def _case(rng, n, n_txp, first_decoy):
    tid = rng.integers(0, n_txp, n).astype(np.uint32)
    score = rng.choice([rng.integers(60, 200), rng.integers(60, 200), rng.integers(60, 200), rng.integers(-40, 260)], n).astype(np.int32)   # many ties
    compat = rng.integers(0, 2, n).astype(np.uint8)
    state = rng.choice([0, 0, 0, 0, 1, 2], n)          # 0 scored, 1 skipped as incompatible, 2 alignment failed
    return tid, score, compat, (state == 1).astype(np.uint8), (state == 2).astype(np.uint8)


def _run_ref(L, tid, score, compat, skipped, failed, first_decoy, n_txp, thr, hard, sexp, minp, lag):
    n = len(tid); slot = np.zeros(n + 1, np.uint32); kt = np.zeros(n + 1, np.uint32); pr = np.zeros(n + 1); info = np.zeros(3, np.int32)
    k = L.ref_select_hits(n, tid.ctypes.data, score.ctypes.data, compat.ctypes.data, skipped.ctypes.data, failed.ctypes.data, first_decoy, n_txp, thr, hard, sexp, minp, lag,
                          slot.ctypes.data, kt.ctypes.data, pr.ctypes.data, info.ctypes.data)
    return slot[:k].copy(), kt[:k].copy(), pr[:k].copy(), info


def _run_chk(L, tid, score, compat, scored, first_decoy, thr, hard, sexp, minp):
    n = len(tid); kept = np.zeros(n + 1, np.uint32); pr = np.zeros(n + 1); info = np.zeros(3, np.int32)
    k = L.orc_select_hits(n, tid.ctypes.data, score.ctypes.data, compat.ctypes.data, scored.ctypes.data, first_decoy, thr, hard, sexp, minp, kept.ctypes.data, pr.ctypes.data, info.ctypes.data)
    return kept[:k].copy(), pr[:k].copy(), info


def test_selection_equals_the_reference_functions(built):
    R, K = _ref(), _checker(); rng = np.random.default_rng(61); seen_decoy_only = seen_dups = seen_cut = 0; worst = 0.0
    for it in range(6000):
        n = int(rng.integers(1, 14)); n_txp = int(rng.integers(2, 9)); first_decoy = int(rng.choice([n_txp, n_txp, n_txp - 1, max(1, n_txp // 2)]))
        tid, score, compat, skipped, failed = _case(rng, n, n_txp, first_decoy)
        thr = float(rng.choice([1.0, 1.0, 0.9, 0.5])); hard = int(rng.random() < 0.2); sexp = float(rng.choice([1.0, 0.5, 2.0])); minp = float(rng.choice([1e-5, 1e-5, 1e-2, 0.3]))
        scored = ((skipped == 0) & (failed == 0)).astype(np.uint8)
        slot, kt, pr, info = _run_ref(R, tid, score, compat, skipped, failed, first_decoy, n_txp, thr, hard, sexp, minp, lag=0)
        kept, pc, ic = _run_chk(K, tid, score, compat, scored, first_decoy, thr, hard, sexp, minp)
        only_decoy = int(info[2] & 3 != 0)
        assert (int(info[0]), int(info[1]), only_decoy) == (int(ic[0]), int(ic[1]), int(ic[2])), (it, info, ic)
        assert np.array_equal(slot, kept) and np.array_equal(kt, tid[kept]), (it, slot, kept)      # the same candidates, the same order
        # estAlnProb = exp(-scoreExp (best - score)): the reference calls libm's exp, checker and kernels share include/sq_math.h's (so that host and device agree bit for bit:
        # tests/test_math.py holds it to libm within an ulp) — an ulp, then, is the most the two may differ by here
        assert np.all(np.abs(pr - pc) <= np.spacing(np.maximum(np.abs(pr), np.abs(pc)))), (it, pr, pc); worst = max(worst, float(np.max(np.abs(pr - pc) / np.maximum(np.abs(pc), 1e-300))) if len(pr) else 0.0)
        seen_decoy_only += only_decoy; seen_dups += int(len(np.unique(tid[scored == 1])) < int(scored.sum())); seen_cut += int(len(kept) < len(np.unique(tid[(scored == 1) & (tid < first_decoy)])))
    assert seen_decoy_only > 50 and seen_dups > 1000 and seen_cut > 300          # the cases the comparison is about all occurred
    assert worst < 2.3e-16


def test_what_the_call_sites_unadvanced_slot_does(built):
    """SalmonQuantify.cpp:1521-1523 skips an incompatible candidate without advancing `idx`: the hit behind it is recorded in the slot of the candidate BEFORE it,
    and filterAndCollectAlignments builds its record from jointHits[that slot].  The transcript and the probability are the hit's own; positions, orientation,
    fragment length and mate status are another candidate's.  Not followed (it depends on pufferfish's candidate order, which is not in the tree): SPEC section a7."""
    R = _ref()
    tid = np.array([3, 5], np.uint32); score = np.array([0, 150], np.int32); compat = np.array([0, 1], np.uint8); skipped = np.array([1, 0], np.uint8); failed = np.zeros(2, np.uint8)
    slot, kt, pr, _ = _run_ref(R, tid, score, compat, skipped, failed, 8, 8, 1.0, 0, 1.0, 1e-5, lag=1)
    assert list(kt) == [5] and list(slot) == [0] and pr[0] == 1.0                # transcript 5's hit, built from candidate 0's record
    slot, kt, pr, _ = _run_ref(R, tid, score, compat, skipped, failed, 8, 8, 1.0, 0, 1.0, 1e-5, lag=0)
    assert list(kt) == [5] and list(slot) == [1]
    # without a skipped candidate in front the two agree
    skipped[:] = 0; score[0] = 120
    a = _run_ref(R, tid, score, compat, skipped, failed, 8, 8, 1.0, 0, 1.0, 1e-5, lag=1); b = _run_ref(R, tid, score, compat, skipped, failed, 8, 8, 1.0, 0, 1.0, 1e-5, lag=0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
