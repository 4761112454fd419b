"""--posBias: the checker's restatement of SimplePosBias (bins, finalize, projection) and of tk::spline against the reference's own
sources, compiled from where they lie under /root/reference into oracle/_ref/libposbias_ref.so (oracle/ref_posbias_shim.cpp;
`make -C oracle ref`).  Skipped where that library was not built."""
import ctypes as C, os
import numpy as np
import pytest
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = C.POINTER(C.c_double)


def _ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libposbias_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libposbias_ref.so not built (no /root/reference on this machine)")
    L = C.CDLL(path)
    L.ref_spline_eval.argtypes = [dp, dp, C.c_int, dp, C.c_int, dp]
    L.ref_pos_project.argtypes = [dp, C.c_int32, dp, dp]
    L.ref_pos_bin.argtypes = [C.c_int32, C.c_int32]; L.ref_pos_bin.restype = C.c_int
    return L


def _p(a): return a.ctypes.data_as(dp)


def test_spline_is_tk_spline_bit_for_bit(built):
    L = _ref(); O = orc.lib(); rng = np.random.default_rng(5)
    for trial in range(40):
        n = int(rng.integers(3, 23))          # the checker holds 22 knots (20 bins + 2 end knots)
        xs = np.sort(rng.random(n)) + np.arange(n) * 1e-3; ys = rng.random(n) * (10.0 ** rng.integers(-6, 3))
        q = np.concatenate([xs, rng.uniform(xs[0], xs[-1], 500)])          # the knots themselves (lower_bound picks the interval before) and points between
        a = np.empty(len(q)); b = np.empty(len(q))
        L.ref_spline_eval(_p(xs), _p(ys), n, _p(q), len(q), _p(a)); O.orc_spline_eval(_p(xs), _p(ys), n, _p(q), len(q), _p(b))
        assert a.tobytes() == b.tobytes(), trial


def test_bins_are_the_references(built):
    L = _ref(); O = orc.lib(); rng = np.random.default_rng(6)
    for length in [1, 2, 3, 7, 19, 20, 21, 39, 40, 41, 100, 999, 1000, 1001, 12345, 99999] + [int(x) for x in rng.integers(30, 50000, 40)]:
        for pos in sorted(set([0, length - 1, length // 2] + [int(x) for x in rng.integers(0, length, 25)])):
            assert O.orc_pos_bin(pos, length) == L.ref_pos_bin(pos, length), (pos, length)


def test_finalize_and_projection_match_the_reference_model(built):
    # the reference keeps log masses and exponentiates in finalize(); the checker keeps linear masses: equal to rounding of that round trip
    L = _ref(); O = orc.lib(); rng = np.random.default_rng(7)
    for trial in range(20):
        extra = rng.random(20) * (10.0 ** rng.integers(0, 7)) + 1e-3         # mass on top of the initial 1.0 per bin
        mass = 1.0 + extra; logm = np.log(extra)
        for length in (50, 777, 4096):
            a = np.empty(length); na = np.empty(20); b = np.empty(length); nb = np.empty(20)
            L.ref_pos_project(_p(logm), length, _p(a), _p(na)); O.orc_pos_project(_p(mass), length, _p(b), _p(nb))
            assert np.allclose(na, nb, rtol=1e-12, atol=0) and np.allclose(a, b, rtol=1e-9, atol=1e-15), (trial, length)
            assert (b >= 0.001).all()
