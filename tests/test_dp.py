"""Selective-alignment scoring arithmetic (row a4): the checker's banded affine-gap DP against an independent
plain-Python Gotoh (no band), plus known answers.  The reference's aligner (ksw2, in the absent pufferfish tree)
cannot be run here; this pins the restatement to the textbook recurrence it claims to be (SPEC §a4)."""
import ctypes as C
import numpy as np
import pytest
from salmon_amd import api
import orc

NEG = -(1 << 29)


def gotoh(q, t, ma, mp, go, ge, mode):
    n, m = len(q), len(t)
    H = [[NEG] * (m + 1) for _ in range(n + 1)]; E = [[NEG] * (m + 1) for _ in range(n + 1)]; F = [[NEG] * (m + 1) for _ in range(n + 1)]
    H[0][0] = 0
    for j in range(1, m + 1): H[0][j] = E[0][j] = -(go + ge * j)
    for i in range(1, n + 1):
        H[i][0] = F[i][0] = -(go + ge * i)
        for j in range(1, m + 1):
            E[i][j] = max(E[i][j - 1], H[i][j - 1] - go) - ge
            F[i][j] = max(F[i - 1][j], H[i - 1][j] - go) - ge
            s = ma if (q[i - 1] == t[j - 1] and q[i - 1] < 4) else mp
            H[i][j] = max(H[i - 1][j - 1] + s, E[i][j], F[i][j])
    return H[n][m] if mode == 0 else max(H[n])


def checker(opts, q, t, mode):
    qa = np.ascontiguousarray(q, np.uint8); ta = np.ascontiguousarray(t, np.uint8)
    return orc.lib().orc_dp_align(C.byref(opts), qa.ctypes.data, len(qa), ta.ctypes.data, len(ta), mode)


@pytest.mark.parametrize("scoring", [dict(), dict(match_score=1, mismatch_penalty=-3, gap_open=4, gap_extend=1),
    dict(match_score=3, mismatch_penalty=-2, gap_open=2, gap_extend=3)])
def test_banded_dp_equals_unbanded_gotoh_when_the_band_covers_the_matrix(built, scoring):
    rng = np.random.default_rng(17)
    opts = api.quant_opts(bandwidth=64, **scoring)     # band >= every case below
    for it in range(400):
        n = int(rng.integers(1, 40)); m = int(rng.integers(1, 40))
        t = rng.integers(0, 4, m)
        if it % 3:                                      # related sequences: a mutated copy
            q = list(t[:n]) + list(rng.integers(0, 4, max(0, n - m)))
            for _ in range(int(rng.integers(0, 4))):
                p = int(rng.integers(0, len(q))); op = int(rng.integers(0, 3))
                if op == 0: q[p] = int(rng.integers(0, 5))
                elif op == 1 and len(q) > 1: del q[p]
                else: q.insert(p, int(rng.integers(0, 4)))
            q = np.array(q)
        else:
            q = rng.integers(0, 5, n)                   # 4 = N: matches nothing
        for mode in (0, 1):
            want = gotoh(list(q), list(t), opts.match_score, opts.mismatch_penalty, opts.gap_open, opts.gap_extend, mode)
            assert checker(opts, q, t, mode) == want, (it, mode, list(q), list(t))


def test_dp_known_answers_and_band_limits(built):
    o = api.quant_opts()                                # ma 2, mp -4, go 6, ge 2, band 15
    a = np.array([0, 1, 2, 3] * 5)
    assert checker(o, a, a, 0) == 40 and checker(o, a, a, 1) == 40
    b = a.copy(); b[7] = (b[7] + 1) % 4
    assert checker(o, b, a, 0) == 38 - 4                # one mismatch: 19 matches, one -4
    assert checker(o, np.delete(a, 9), a, 0) == 38 - 8  # one base missing from the read: gap of 1 = -(6+2)
    assert checker(o, a, np.concatenate([a, [0, 0, 0]]), 1) == 40           # extension: target tail is free
    assert checker(o, a, np.concatenate([a, [0, 0, 0]]), 0) == 40 - 12      # global: it is not (gap of 3)
    n4 = np.full(6, 4)
    assert checker(o, n4, np.zeros(6, np.int64), 0) == -24                  # N mismatches every base
    # outside the band the corner is unreachable: the end is invalid (a large negative sentinel)
    assert checker(o, a, np.concatenate([a, np.zeros(16, np.int64)]), 0) < -(1 << 28)
    assert checker(o, a, np.concatenate([a, np.zeros(15, np.int64)]), 0) == 40 - (6 + 2 * 15)
    # empty sides
    e = np.zeros(0, np.int64)
    assert checker(o, e, e, 0) == 0 and checker(o, e, a, 1) == 0 and checker(o, e, a[:3], 0) == -12 and checker(o, a[:2], e, 0) == -10
