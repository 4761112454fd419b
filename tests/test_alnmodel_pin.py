"""[r5] The CIGAR-based alignment error model of alignment-based input (row f4) against the reference's own code: src/alignment/AlignmentModel.cpp and
AlignmentCommon.cpp compiled from where they lie under /root/reference into oracle/_ref/libalnmodel_ref.so (oracle/ref_alnmodel_shim.cpp; htslib's bam1_t,
spdlog, TBB, SalmonUtils / Transcript / ReadPair / UnpairedRead stood in for under oracle/_stub/aln).  A stream of random alignments — matches, mismatches,
insertions, deletions, reference skips, soft and hard clips, pads, pairs whose ends tie or cross, orphans, single-end reads, CIGARs that run past the read
or the transcript — is scored and learned from by both: every log-likelihood equal to 1e-9 while the matrices evolve under the updates.  The checker
applies an update as a mini-batch of its own here (SPEC D1 batches them in fixed point: the same sums).  Skipped where the library was not built."""
import ctypes as C, os
import numpy as np
import pytest
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EVAL = [C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int32, C.c_int, C.c_double, C.c_double]


def _ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libalnmodel_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libalnmodel_ref.so not built (no /root/reference on this machine)")
    L = C.CDLL(path); L.ref_aln_model_new.restype = C.c_void_p; L.ref_aln_model_new.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32]
    L.ref_aln_model_free.argtypes = [C.c_void_p]; L.ref_aln_model_eval.restype = C.c_double; L.ref_aln_model_eval.argtypes = EVAL
    return L


def _record(rng, T, tlen, messy):
    """a read taken from the transcript with substitutions, its CIGAR (BAM encoding) and bases; `messy` adds the rare operations and inconsistencies"""
    L = int(rng.integers(30, 150)); pos = int(rng.integers(0, tlen - 20)); ops = []; seq = []; t = pos; left = L
    if messy and rng.random() < 0.3: n = int(rng.integers(1, 6)); ops.append((5, n))                     # H
    if rng.random() < 0.3: n = int(rng.integers(1, 8)); ops.append((4, n)); seq += list(rng.integers(0, 4, n)); left -= n   # S
    while left > 0:
        u = rng.random(); n = int(min(left, rng.integers(1, 40)))
        if u < 0.7 or not ops or ops[-1][0] not in (0, 7, 8):
            op = 0 if u < 0.6 or not messy else (7 if rng.random() < 0.5 else 8)
            for j in range(n): b = int(T[min(t + j, tlen - 1)]); seq.append(b if rng.random() > 0.05 else int(rng.integers(0, 4)))
            ops.append((op, n)); t += n; left -= n
        elif u < 0.8: n = int(min(left, rng.integers(1, 4))); ops.append((1, n)); seq += list(rng.integers(0, 4, n)); left -= n      # I
        elif u < 0.9: n = int(rng.integers(1, 4)); ops.append((2, n)); t += n                                                          # D
        elif messy and u < 0.95: n = int(rng.integers(1, 30)); ops.append((3, n)); t += n                                              # N
        elif messy: ops.append((6, int(rng.integers(1, 3))))                                                                          # P
    if messy and rng.random() < 0.2: ops.append((5, int(rng.integers(1, 5))))                            # trailing H: the two walks treat it differently
    if messy and rng.random() < 0.05: ops.append((0, int(rng.integers(1, 9))))                           # a CIGAR that claims more bases than the read has
    cig = np.array([(n << 4) | op for op, n in ops], np.uint32); s = np.array(seq, np.uint8)
    return pos, cig, s


@pytest.mark.parametrize("bins,messy", [(6, False), (4, True), (1, True)])
def test_error_model_follows_alignmentmodel(built, bins, messy):
    R = _ref(); O = orc.lib(); O.orc_errmodel_new.restype = C.c_void_p; O.orc_errmodel_new.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32]
    O.orc_errmodel_free.argtypes = [C.c_void_p]; O.orc_errmodel_eval.restype = C.c_double; O.orc_errmodel_eval.argtypes = EVAL
    rng = np.random.default_rng(bins * 10 + messy); tlen = 700; T = rng.integers(0, 4, tlen).astype(np.uint8)
    hr = R.ref_aln_model_new(bins, T.ctypes.data, tlen); ho = O.orc_errmodel_new(bins, T.ctypes.data, tlen); worst = 0.0
    try:
        for it in range(1500):
            kind = int(rng.integers(0, 4)); p1, c1, s1 = _record(rng, T, tlen, messy); p2, c2, s2 = _record(rng, T, tlen, messy)
            if kind == 0 and rng.random() < 0.2: p2 = p1                                                 # a tie: the second record is the left one
            upd = int(rng.random() < 0.6); p = 0.0 if rng.random() < 0.7 else float(-rng.integers(0, 25)); mass = float(-rng.random() * 3)
            args = (kind, p1, c1.ctypes.data, len(c1), s1.ctypes.data, len(s1), p2, c2.ctypes.data, len(c2), s2.ctypes.data, len(s2), upd, p, mass)
            a = R.ref_aln_model_eval(hr, *args); b = O.orc_errmodel_eval(ho, *args)
            assert np.isfinite(a) and abs(a - b) <= 1e-9 * max(1.0, abs(a)), (it, kind, a, b)
            worst = max(worst, abs(a - b))
        # an alignment that starts behind the transcript's end has no likelihood (:111-116); one without a CIGAR has LOG_EPSILON (:128-130)
        z = np.zeros(0, np.uint32); a = R.ref_aln_model_eval(hr, 3, tlen + 5, c1.ctypes.data, len(c1), s1.ctypes.data, len(s1), 0, z.ctypes.data, 0, s1.ctypes.data, 0, 0, 0.0, 0.0)
        b = O.orc_errmodel_eval(ho, 3, tlen + 5, c1.ctypes.data, len(c1), s1.ctypes.data, len(s1), 0, z.ctypes.data, 0, s1.ctypes.data, 0, 0, 0.0, 0.0)
        assert a == b == np.inf                                                            # LOG_0 is +HUGE_VAL in the reference (SalmonMath.hpp:40) and here: every test on it is on the absolute value
        a = R.ref_aln_model_eval(hr, 3, 5, z.ctypes.data, 0, s1.ctypes.data, len(s1), 0, z.ctypes.data, 0, s1.ctypes.data, 0, 0, 0.0, 0.0)
        b = O.orc_errmodel_eval(ho, 3, 5, z.ctypes.data, 0, s1.ctypes.data, len(s1), 0, z.ctypes.data, 0, s1.ctypes.data, 0, 0, 0.0, 0.0)
        assert a == b and abs(a + 24.0066801829) < 1e-9                                             # log(0.375e-10)
    finally:
        R.ref_aln_model_free(hr); O.orc_errmodel_free(ho)
