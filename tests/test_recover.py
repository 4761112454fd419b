"""Orphan recovery (--recoverOrphans).

The infix aligner behind it is the one part of the mapping path whose reference source is in-tree and compiles on
its own (reference src/edlib.cpp).  tests/golden/edlib_infix_vectors.json.gz holds its answers on 600 cases (made by
tests/golden/make_edlib_vectors.py from the compiled reference file); where oracle/_ref/libedlib_ref.so is present the
checker is also compared with it live.  The -m gpu tests hold the HIP aligner and the whole mapping path to the same."""
import ctypes as C
import gzip, json, os
import numpy as np
import pytest
from salmon_amd import api
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = np.full(256, 4, np.uint8)
for i, ch in enumerate(b"ACGT"):
    CODE[ch] = i


def vectors():
    with gzip.open(os.path.join(ROOT, "tests", "golden", "edlib_infix_vectors.json.gz"), "rb") as f:
        return json.loads(f.read().decode())


def checker_infix(q_ascii, t_ascii, k):
    L = orc.lib()
    L.orc_infix_align.restype = C.c_int
    L.orc_infix_align.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 3
    q = np.ascontiguousarray(CODE[np.frombuffer(q_ascii, np.uint8)]); t = np.ascontiguousarray(CODE[np.frombuffer(t_ascii, np.uint8)])
    v = [C.c_int(), C.c_int(), C.c_int()]
    ok = L.orc_infix_align(q.ctypes.data, len(q), t.ctypes.data, len(t), k, *[C.byref(x) for x in v])
    return (1, v[0].value, v[1].value, v[2].value) if ok else (0, -1, -1, -1)


def test_checker_infix_aligner_matches_reference_vectors():
    vs = vectors()
    assert len(vs) == 600 and sum(c["found"] for c in vs) > 300
    for i, c in enumerate(vs):
        got = checker_infix(c["q"].encode(), c["t"].encode(), c["k"])
        assert got == (c["found"], c["ed"], c["start"], c["end"]), "vector %d: checker %s, reference %s" % (i, got, c)


def test_checker_infix_aligner_matches_compiled_reference_live():
    path = os.path.join(ROOT, "oracle", "_ref", "libedlib_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    ref = C.CDLL(path); ip = C.POINTER(C.c_int)
    ref.ref_edlib_infix.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, ip, ip, ip, ip]
    rng = np.random.default_rng(99); A = np.frombuffer(b"ACGTN", np.uint8)
    for it in range(1500):
        m = int(rng.integers(1, 1001)); n = int(rng.integers(1, 257)); t = rng.integers(0, 4, m)
        if it % 3 and m >= n:
            s = int(rng.integers(0, m - n + 1)); q = t[s:s + n].copy()
            for _ in range(int(rng.integers(0, max(1, n // 3)))):
                p = int(rng.integers(0, len(q))); op = int(rng.integers(0, 3))
                if op == 0: q[p] = (q[p] + 1) % 4
                elif op == 1 and len(q) > 1: q = np.delete(q, p)
                else: q = np.insert(q, p, rng.integers(0, 4))
            q = q[:256]
            if it % 7 == 0: q[int(rng.integers(0, len(q)))] = 4
        else:
            q = rng.integers(0, 4, n)
        qs, ts = A[q].tobytes(), A[t].tobytes(); k = len(q) // 4
        v = [C.c_int() for _ in range(4)]
        ok = ref.ref_edlib_infix(qs, len(qs), ts, len(ts), k, *[C.byref(x) for x in v])
        want = (1, v[0].value, v[1].value, v[2].value) if ok else (0, -1, -1, -1)
        assert checker_infix(qs, ts, k) == want, "case %d" % it


def damaged_pairs(tx, n, seed, read_len=100):
    """Read pairs whose second (or first) mate has a substitution every ~17 bases — no 31-mer survives, so the mate has no
    seed and the fragment maps as an orphan; the mate is still within len/4 edits of the transcript.  A few mates get an
    indel as well, a few are replaced by random sequence (not recoverable)."""
    seq, off, tt, tp = tx.reads(n, read_len=read_len, seed=seed, sub_rate=0.0, indel_rate=0.0, junk_frac=0.0, threads=2)
    seq = seq.copy(); rng = np.random.default_rng(seed + 1)
    comp = {65: 67, 67: 71, 71: 84, 84: 65}
    kinds = np.zeros(n, np.uint8)
    for i in range(0, n, 2):            # every second pair is damaged
        e = 2 * i + int(rng.integers(0, 2)); a = int(off[e]); kind = int(rng.integers(0, 10)); kinds[i] = 1 + kind
        if kind == 9:
            seq[a:a + read_len] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, read_len)]
            continue
        for p in range(int(rng.integers(3, 14)), read_len, 17):
            seq[a + p] = comp.get(int(seq[a + p]), 65)
        if kind >= 6:                   # one deleted base: shift the tail left, pad with the last base
            p = int(rng.integers(20, read_len - 20)); seq[a + p:a + read_len - 1] = seq[a + p + 1:a + read_len].copy()
    return seq, off, tt, tp, kinds


def test_checker_recovers_orphans(small_world):
    w = small_world; n = 600
    seq, off, tt, tp, kinds = damaged_pairs(w["tx"], n, seed=314)
    rb = api.make_read_batch(seq, off, n, paired=True)
    o0 = api.quant_opts(); o1 = api.quant_opts(recover_orphans=1)
    ro0, aln0, mt0, st0 = orc.map_batch(w["oidx"], o0, rb, threads=4)
    ro1, aln1, mt1, st1 = orc.map_batch(w["oidx"], o1, rb, threads=4)
    assert st0["num_orphans_rescued"] == 0
    orphans0 = int(np.sum((mt0 == 1) | (mt0 == 2)))
    assert orphans0 > 0.35 * n                      # the damaged half maps as orphans without recovery
    assert st1["num_orphans_rescued"] > 0.7 * orphans0
    assert int(np.sum(mt1 == 4)) >= int(np.sum(mt0 == 4)) + st1["num_orphans_rescued"] * 0.9
    # untouched pairs are unaffected; recovered fragments report a proper pair on the anchor's transcript
    for f in range(1, n, 2):
        assert aln0[ro0[f]:ro0[f + 1]].tobytes() == aln1[ro1[f]:ro1[f + 1]].tobytes()
    rec = [f for f in range(0, n, 2) if mt0[f] in (1, 2) and mt1[f] == 4]
    assert len(rec) > 0.6 * orphans0
    hit = 0
    for f in rec:
        a1 = aln1[ro1[f]:ro1[f + 1]]; pairs = a1[a1["mate_status"] == 3]
        assert len(pairs) and np.all((pairs["frag_len"] > 0) & (pairs["frag_len"] <= 1000)) and np.all(pairs["fwd"] != pairs["mate_fwd"])
        hit += int(int(tp[f]) in np.minimum(pairs["pos"], pairs["mate_pos"]))
    assert hit > 0.9 * len(rec)                     # the recovered fragment starts where the simulator drew it (on the true isoform)


@pytest.mark.gpu
def test_device_infix_aligner_matches_reference_vectors(built):
    vs = vectors()
    out = api.debug_infix_align([c["q"].encode() for c in vs], [c["t"].encode() for c in vs], [c["k"] for c in vs])
    want = np.array([[c["found"], c["ed"], c["start"], c["end"]] for c in vs], np.int32)
    bad = np.nonzero(np.any(out != want, axis=1))[0]
    assert len(bad) == 0, "vector %d: device %s, reference %s" % (bad[0], out[bad[0]], want[bad[0]])


@pytest.mark.gpu
def test_device_infix_aligner_matches_checker_on_edge_shapes(built):
    # word boundaries of the bit-vector (63/64/65, 127/128/129, 255/256), windows shorter than the query, k = 0
    rng = np.random.default_rng(4242); A = np.frombuffer(b"ACGT", np.uint8); qs, ts, ks = [], [], []
    for n in (1, 2, 31, 63, 64, 65, 100, 127, 128, 129, 150, 191, 192, 193, 255, 256):
        for m in (1, n // 2 + 1, n, n + 1, n + 40, 1000):
            t = rng.integers(0, 4, m); s = int(rng.integers(0, max(1, m - n + 1))); q = t[s:s + n].copy()
            if len(q) < n: q = np.concatenate([q, rng.integers(0, 4, n - len(q))])
            for e in range(int(rng.integers(0, 4))): q[int(rng.integers(0, n))] = int(rng.integers(0, 4))
            for k in (0, n // 4):
                qs.append(A[q].tobytes()); ts.append(A[t].tobytes()); ks.append(k)
    out = api.debug_infix_align(qs, ts, ks)
    for i in range(len(qs)):
        assert tuple(out[i]) == checker_infix(qs[i], ts[i], ks[i]), "case %d (n=%d, m=%d, k=%d)" % (i, len(qs[i]), len(ts[i]), ks[i])


@pytest.mark.gpu
@pytest.mark.parametrize("read_len", [100, 150])
def test_orphan_recovery_matches_checker(small_world, read_len):
    w = small_world; n = 1200
    seq, off, tt, tp, kinds = damaged_pairs(w["tx"], n, seed=2718 + read_len, read_len=read_len)
    rb = api.make_read_batch(seq, off, n, paired=True)
    opts = api.quant_opts(recover_orphans=1)
    w["idx"].to_device(0)
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=2048)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
    ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=4)
    assert st_g == st_c and st_g["num_orphans_rescued"] > 0.3 * n
    assert np.array_equal(ro_g, ro_c) and np.array_equal(mt_g, mt_c)
    assert aln_g.tobytes() == aln_c.tobytes()
    cd_g = ctx.tap(4, api.CAND_DTYPE); _, _, _, cd_c = orc.map_taps(w["oidx"], opts, rb)
    assert cd_g.tobytes() == cd_c.tobytes()
    # the online stage takes recovered pairs like any other pair
    ctx.eq_accumulate(); eq_g = ctx.eq_finish()
    ost = orc.OrcState(w["oidx"], opts); ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"]); ost.finish(); eq_c = ost.eq_finish()
    assert np.array_equal(eq_g.tid, eq_c.tid) and np.array_equal(eq_g.count, eq_c.count) and np.array_equal(eq_g.wq, eq_c.wq)
    ctx.free(); ost.free()
