"""[r4] The limits rounds 1-3 had — read ends cut at 256 bases, uni-MEMs after the 32nd of an end dropped — are gone: reads of up to 1000 bases
are mapped whole (the packing stride of the context widens: 8 -> 16 -> 32 words), an end with more uni-MEMs than the slab has slots makes the
slab widen and takes the large-end path, and a read beyond 1000 bases is REFUSED, not cut.  2x300 pairs are held to the checker (every stage,
byte for byte) and to the exhaustive aligner; a haplotype family whose graph breaks every few bases produces the ends with > 32 uni-MEMs."""
import os
import numpy as np
import pytest
import orc, exh
from salmon_amd import api, synth, capi


@pytest.fixture(scope="module")
def world300(built):
    tx = synth.Txome(seed=21, n_genes=150, iso_per_gene=4, threads=4)
    names, seqs, lens = tx.tables()
    idx = api.SalmonIndex.build_mem_raw(tx.n, names, seqs, lens, threads=4)
    n = 1500
    seq, off, tt, tp = tx.reads(n, read_len=300, seed=5, threads=4)
    raw = dict(zip(tx.names(), tx.seqs()))
    refs = [raw[nm][:l] for nm, l in zip(idx.ref_names(), idx.ref_lens())]   # the index's references: duplicates dropped, poly-A tails clipped (as tests/golden/make_exhaustive.py)
    return dict(tx=tx, idx=idx, seq=seq, off=off, n=n, refs=refs)


def _haplotype_world(read_len=400, n_hap=24, n=400, seed=3):
    """One 3000-base locus, `n_hap` haplotypes that differ at sites 6 bases apart (each site: one of two alleles at random): a k-mer window covers
    five sites, so haplotypes share a k-mer here and differ there — the compacted graph breaks into unitigs of a handful of k-mers and an error-free
    read of 400 bases crosses dozens of them."""
    rng = np.random.default_rng(seed)
    base = rng.choice(list("ACGT"), 3000)
    alt = {"A": "C", "C": "G", "G": "T", "T": "A"}
    sites = np.arange(40, 2960, 6)
    haps = []
    for h in range(n_hap):
        s = base.copy(); flip = rng.integers(0, 2, len(sites)).astype(bool)
        s[sites[flip]] = [alt[c] for c in s[sites[flip]]]
        haps.append("".join(s))
    names = ["hap%d" % i for i in range(n_hap)]
    idx = api.SalmonIndex.build_mem(names, haps, threads=4, keep_duplicates=True, no_clip=True)
    recs = []
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for i in range(n):
        h = haps[int(rng.integers(0, n_hap))]; fl = int(rng.integers(read_len, 900)); st = int(rng.integers(0, len(h) - fl))
        frag = h[st:st + fl]
        recs.append(frag[:read_len]); recs.append("".join(comp[c] for c in reversed(frag[-read_len:])))
    seq = np.frombuffer("".join(recs).encode(), np.uint8).copy(); off = np.arange(0, 2 * n + 1, dtype=np.uint64) * np.uint64(read_len)
    return dict(idx=idx, seq=seq, off=off, n=n)


def test_checker_keeps_every_unimem_and_whole_reads(built):
    # CPU: the haplotype world really produces ends with more than 32 uni-MEMs, and 400-base reads are mapped over their whole length
    w = _haplotype_world(); oidx = orc.OrcIndex(w["idx"]); opts = api.quant_opts()
    rb = api.make_read_batch(w["seq"], w["off"], w["n"], paired=True)
    um, mm, ch, cd = orc.map_taps(oidx, opts, rb)
    per_end = np.bincount(um["end"], minlength=2 * w["n"])
    assert per_end.max() > 40 and (per_end > 32).sum() > 50, (per_end.max(), (per_end > 32).sum())
    assert (um["qpos"].astype(int) + um["len"].astype(int)).max() == 400          # uni-MEMs reach the last base of a 400-base read
    ro, aln, mt, st = orc.map_batch(oidx, opts, rb, threads=4)
    assert st["num_truncated_ends"] == 0 and st["num_mapped"] > 0.95 * w["n"] and int(aln["read_len"].max()) == 400


def test_checker_on_2x300_pairs_agrees_with_the_exhaustive_aligner(world300):
    # CPU: reads three times the old packing limit against the all-positions aligner — the same label sets, the same scores
    w = world300; opts = api.quant_opts(); oidx = orc.OrcIndex(w["idx"]); k = 150
    rb = api.make_read_batch(w["seq"], w["off"], k, paired=True)
    ro, aln, mt, st = orc.map_batch(oidx, opts, rb, threads=os.cpu_count() or 8)
    lo, lt, ls, kind = exh.labels(w["refs"], w["seq"], w["off"], k, opts, threads=os.cpu_count() or 8)
    c = exh.compare(lo, lt, ro, aln["tid"]); assert c["agreement"] >= 0.98, {x: v for x, v in c.items() if x != "examples"}
    assert int(aln["read_len"].max()) == 300 and st["num_truncated_ends"] == 0


def _fields_equal(a, b, fields, what):
    assert len(a) == len(b), (what, len(a), len(b))
    for f in fields: assert np.array_equal(a[f], b[f]), (what, f)


@pytest.mark.gpu
def test_2x300_pairs_equal_the_checker_at_every_stage_and_the_exhaustive_aligner(world300):
    w = world300; w["idx"].to_device(0); opts = api.quant_opts(); oidx = orc.OrcIndex(w["idx"])
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=2048)
    rb = api.make_read_batch(w["seq"], w["off"], w["n"], paired=True)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)                               # the first batch with 300-base reads: the stride goes 8 -> 16 words and the batch is packed again
    ro_c, aln_c, mt_c, st_c = orc.map_batch(oidx, opts, rb, threads=8)
    assert np.array_equal(ro_g, ro_c) and np.array_equal(mt_g, mt_c) and st_g == st_c and st_g["num_truncated_ends"] == 0
    _fields_equal(aln_g, aln_c, list(api.ALN_DTYPE.names), "alignments")
    assert int(aln_g["read_len"].max()) == 300 and st_g["num_mapped"] > 0.9 * w["n"]
    um_c, mm_c, ch_c, cd_c = orc.map_taps(oidx, opts, rb)
    _fields_equal(ctx.tap(1, api.UNIMEM_DTYPE), um_c, ["end", "qpos", "len", "unitig", "uoff", "fw"], "uni-MEMs")
    _fields_equal(ctx.tap(2, api.MEM_DTYPE), mm_c, ["end", "tid", "rpos", "qpos", "len", "fw"], "MEMs")
    _fields_equal(ctx.tap(3, api.CHAIN_DTYPE), ch_c, ["end", "tid", "pos", "last_end", "fw", "n_mems", "score"], "chains")
    # a second batch goes straight through with the widened stride; 100-base reads in the same context still map as before
    ro2, aln2, _, st2 = ctx.map_batch(rb); assert np.array_equal(ro2, ro_c) and aln2.tobytes() == aln_c.tobytes()
    s100, o100, _, _ = w["tx"].reads(1000, read_len=100, seed=9, threads=4); rb100 = api.make_read_batch(s100, o100, 1000, paired=True)
    r1, a1, _, _ = ctx.map_batch(rb100); r1c, a1c, _, _ = orc.map_batch(oidx, opts, rb100, threads=8)
    assert np.array_equal(r1, r1c) and a1.tobytes() == a1c.tobytes()
    # the exhaustive aligner (no index, no seeds, no band) on the first 60 pairs: the same label sets wherever it finds a concordant pair, the same scores
    k = 200
    lo, lt, ls, kind = exh.labels(w["refs"], w["seq"], w["off"], k, opts, threads=os.cpu_count() or 8)
    c = exh.compare(lo, lt, ro_g[:k + 1], aln_g["tid"])
    assert c["agreement"] >= 0.95, {x: v for x, v in c.items() if x != "examples"}
    shared = differs = 0
    for f in range(k):
        a = dict(zip(lt[int(lo[f]):int(lo[f + 1])].tolist(), ls[int(lo[f]):int(lo[f + 1])].tolist()))
        for x in aln_g[int(ro_g[f]):int(ro_g[f + 1])]:
            if int(x["tid"]) in a and (x["mate_status"] == 3) == (kind[f] == 1):
                shared += 1; differs += a[int(x["tid"])] != int(x["score"]) + (int(x["mate_score"]) if x["mate_status"] == 3 else 0)
    assert shared >= k and differs == 0, (shared, differs)
    # the whole path: eq-classes equal the checker's
    ctx.reset(); ctx.map_batch(rb, fetch=False); ctx.eq_accumulate(); eq_g = ctx.eq_finish()
    ost = orc.OrcState(oidx, opts); ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"]); ost.finish(); eq_c = ost.eq_finish()
    assert np.array_equal(eq_g.tid, eq_c.tid) and np.array_equal(eq_g.count, eq_c.count) and np.array_equal(eq_g.wq, eq_c.wq)
    ctx.free()


@pytest.mark.gpu
def test_ends_with_more_than_32_unimems_widen_the_slab_and_equal_the_checker(built):
    w = _haplotype_world(); w["idx"].to_device(0); opts = api.quant_opts(); oidx = orc.OrcIndex(w["idx"])
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=1024)
    rb = api.make_read_batch(w["seq"], w["off"], w["n"], paired=True)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
    ro_c, aln_c, mt_c, st_c = orc.map_batch(oidx, opts, rb, threads=8)
    um_c, mm_c, ch_c, cd_c = orc.map_taps(oidx, opts, rb)
    um_g = ctx.tap(1, api.UNIMEM_DTYPE)
    assert np.bincount(um_g["end"]).max() > 40                                   # nothing was dropped after the 32nd
    _fields_equal(um_g, um_c, ["end", "qpos", "len", "unitig", "uoff", "fw"], "uni-MEMs")
    _fields_equal(ctx.tap(2, api.MEM_DTYPE), mm_c, ["end", "tid", "rpos", "qpos", "len", "fw"], "MEMs")
    _fields_equal(ctx.tap(3, api.CHAIN_DTYPE), ch_c, ["end", "tid", "pos", "last_end", "fw", "n_mems", "score"], "chains")
    assert np.array_equal(ro_g, ro_c) and aln_g.tobytes() == aln_c.tobytes() and st_g == st_c
    ctx.free()


@pytest.mark.gpu
def test_reads_beyond_1000_bases_are_refused_not_cut(world300):
    w = world300; w["idx"].to_device(0)
    ctx = api.QuantContext(w["idx"], api.quant_opts(), device=0, max_batch_reads=64)
    long_read = np.frombuffer((w["refs"][0] * 3)[:1200], np.uint8)
    seq = np.concatenate([long_read, w["seq"][:300]]); off = np.array([0, 1200, 1500], np.uint64)
    with pytest.raises(capi.SalmonHipError, match="1200 bases"):
        ctx.map_batch(api.make_read_batch(seq, off, 1, paired=True))
    # 1000 bases exactly is fine (stride 32 words)
    seq = np.concatenate([long_read[:1000], w["seq"][:300]]); off = np.array([0, 1000, 1300], np.uint64)
    ro, aln, mt, st = ctx.map_batch(api.make_read_batch(seq, off, 1, paired=True))
    assert st["num_truncated_ends"] == 0
    # and with --recoverOrphans reads over 256 bases are refused by name
    c2 = api.QuantContext(w["idx"], api.quant_opts(recover_orphans=1), device=0, max_batch_reads=64)
    with pytest.raises(capi.SalmonHipError, match="recoverOrphans"):
        c2.map_batch(api.make_read_batch(w["seq"][:600], np.array([0, 300, 600], np.uint64), 1, paired=True))
    ctx.free(); c2.free()
