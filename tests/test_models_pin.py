"""[r4] --seqBias / --gcBias: the checker's restatement of the read-start context model (sb_cell, sb_rc, sb_normalize, sb_eval) and of the
fragment-GC model's bins, normalisation and clamped ratio against the reference's own SBModel.cpp and GCFragModel.hpp, compiled from where
they lie under /root/reference into oracle/_ref/libmodels_ref.so (oracle/ref_models_shim.cpp; `make -C oracle ref`).  Skipped where that
library was not built."""
import ctypes as C, os
import numpy as np
import pytest
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = C.POINTER(C.c_double)


def _ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libmodels_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libmodels_ref.so not built (no /root/reference on this machine)")
    L = C.CDLL(path)
    L.ref_sb_train_eval.argtypes = [C.c_char_p, dp, C.c_void_p, C.c_int, dp, dp, C.c_char_p, C.c_int, dp]
    L.ref_gc_ratio.argtypes = [dp, dp, dp]
    L.ref_gc_bins.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]; L.ref_gc_bins.restype = C.c_int
    return L


def _orc():
    O = orc.lib()
    O.orc_sb_normalize.argtypes = [dp, dp]; O.orc_sb_cell.argtypes = [C.c_uint32, C.c_int]; O.orc_sb_cell.restype = C.c_uint32
    O.orc_sb_rc.argtypes = [C.c_uint32]; O.orc_sb_rc.restype = C.c_uint32
    O.orc_sb_eval.argtypes = [dp, C.c_uint32]; O.orc_sb_eval.restype = C.c_double
    O.orc_gc_ratio.argtypes = [dp, dp, dp]; O.orc_gc_frag_bin.argtypes = [C.c_int]; O.orc_gc_ctx_bin.argtypes = [C.c_int]
    return O


def _p(a): return a.ctypes.data_as(dp)
def _code(s): return int(sum("ACGT".index(c) << (2 * (8 - i)) for i, c in enumerate(s)))     # first base in the high bits (oracle.cpp sb_ctx)


def test_context_model_counts_probabilities_and_scores_are_sbmodels(built):
    L = _ref(); O = _orc(); rng = np.random.default_rng(11)
    for trial, n in enumerate((1, 7, 300, 20000)):
        ctx = ["".join(rng.choice(list("ACGT"), 9, p=[0.1, 0.4, 0.3, 0.2])) for _ in range(n)]
        rc = rng.integers(0, 2, n).astype(np.uint8)
        w = np.ones(n) if trial % 2 == 0 else rng.integers(1, 50, n).astype(np.float64)          # the observed model adds 1 per fragment, the expected one fractional masses
        q = ["".join(rng.choice(list("ACGT"), 9)) for _ in range(500)]
        lp_r = np.zeros(576); mg = np.zeros(36); qr = np.zeros(len(q))
        L.ref_sb_train_eval("".join(ctx).encode(), _p(w), rc.ctypes.data, n, _p(lp_r), _p(mg), "".join(q).encode(), len(q), _p(qr))
        # the checker's side: the same contexts through sb_rc / sb_cell into plain counts, then sb_normalize and sb_eval
        cnt = np.zeros(576)
        for s, r, ww in zip(ctx, rc, w):
            v = _code(s); v = O.orc_sb_rc(v) if r else v
            for i in range(9): cnt[O.orc_sb_cell(v, i)] += ww
        lp_o = np.zeros(576); O.orc_sb_normalize(_p(cnt), _p(lp_o))
        order = [0, 1, 2, 2, 2, 2, 2, 2, 2]
        used = np.concatenate([np.arange(i * 64, i * 64 + 4 ** (order[i] + 1)) for i in range(9)])   # the cells a position of that order can address
        assert np.allclose(lp_r[used], lp_o[used], rtol=1e-12, atol=1e-12), trial                    # (the reference also "normalises" the rows above them: never read)
        qo = np.array([O.orc_sb_eval(_p(lp_o), _code(s)) for s in q])
        assert np.allclose(qr, qo, rtol=1e-12, atol=1e-11), trial
        if n >= 300: assert np.abs(np.exp(lp_o[:4]).sum() - 1.0) < 1e-9                              # a proper distribution at position 0


def test_reverse_complement_is_the_kmer_words(built):
    L = _ref(); O = _orc()
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for s in ("ACGTACGTA", "AAAAAAAAC", "TGCATGCAT", "GATTACAGA"):
        assert O.orc_sb_rc(_code(s)) == _code("".join(comp[c] for c in reversed(s)))


def test_gc_bins_normalisation_and_ratio_are_gcfragmodels(built):
    L = _ref(); O = _orc(); rng = np.random.default_rng(12)
    for f in range(0, 101):
        cb = C.c_int(); fb = L.ref_gc_bins(f, f, C.byref(cb))
        assert fb == O.orc_gc_frag_bin(f) and cb.value == O.orc_gc_ctx_bin(f), f
    for trial in range(30):
        obs = rng.random(75) * (10.0 ** rng.integers(-3, 6)); ex = rng.random(75) * (10.0 ** rng.integers(-3, 6))
        if trial % 3 == 0: obs[rng.integers(0, 75, 20)] = 0.0                   # bins nothing fell into
        if trial % 5 == 0: ex[25:50] = 0.0                                      # a whole context class without expectation
        a = np.zeros(75); b = np.zeros(75)
        L.ref_gc_ratio(_p(obs), _p(ex), _p(a)); O.orc_gc_ratio(_p(obs), _p(ex), _p(b))
        # the reference keeps the observed masses in log space and exponentiates in normalize(); the checker keeps them linear: equal to the rounding of that round trip
        assert np.allclose(a, b, rtol=1e-10, atol=0), (trial, np.abs(a / b - 1).max())
        assert b.min() >= 1e-3 and b.max() <= 1e3
