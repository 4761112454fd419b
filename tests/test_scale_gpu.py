"""GPU path vs the CPU checker at a bench-like size: 20 000 transcripts, 200 000 read pairs in two batches, the online
model crossing pre-burn-in -> aux params -> burned in.  Everything is compared bit for bit: alignments, counters, the
online model, the eq-class table with its fixed-point weight sums, projected counts, VBEM iterations and alphas."""
import hashlib
import numpy as np
import pytest
from salmon_amd import api, synth
import orc

pytestmark = pytest.mark.gpu
N = 200000


# burn-in inside the second of two batches; [r6] burn-in inside the second of four: the third and fourth run the burned-in chain (k_chain: one launch per batch,
# ten mini-batches of the default 5 000 in two groups of the default W = 8, 160 workgroups behind a counter barrier)
@pytest.mark.parametrize("B,burnin", [(100000, 120000), (50000, 60000)])
def test_gpu_equals_checker_at_20k_transcripts_200k_pairs(built, B, burnin):
    tx = synth.Txome(seed=5, n_genes=3200, iso_per_gene=8, threads=16)
    names, seqs, lens = tx.tables()
    idx = api.SalmonIndex.build_mem_raw(tx.n, names, seqs, lens, threads=16).to_device(0)
    assert idx.num_refs >= 19000
    oidx = orc.OrcIndex(idx)
    seq, off, tt, tp = tx.reads(N, read_len=100, seed=6, threads=16)
    opts = api.quant_opts(num_burnin_frags=burnin)       # burn-in falls inside the second batch
    ctx = api.QuantContext(idx, opts, device=0, max_batch_reads=B)
    ost = orc.OrcState(oidx, opts)
    tot_g = tot_c = None
    import os
    thr = os.cpu_count() or 8
    for i in range(N // B):
        lo, hi = i * B, (i + 1) * B
        s = seq[lo * 200: hi * 200]; o = (off[2 * lo: 2 * hi + 1] - off[2 * lo]).copy()
        rb = api.make_read_batch(s, o, B, paired=True)
        ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
        ctx.eq_accumulate()
        ro_c, aln_c, mt_c, st_c = orc.map_batch(oidx, opts, rb, threads=thr)
        ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"])
        assert st_g == st_c
        assert np.array_equal(ro_g, ro_c) and np.array_equal(mt_g, mt_c)
        assert hashlib.sha256(aln_g.tobytes()).hexdigest() == hashlib.sha256(aln_c.tobytes()).hexdigest()
        assert st_g["num_mapped"] > 0.9 * B
    ost.finish()
    assert ctx.summary() == ost.summary() and ctx.summary()["burned_in"]
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    assert len(eq_g.count) > 10000
    for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
        assert np.array_equal(getattr(eq_g, f), getattr(eq_c, f)), f
    mg, mc = ctx.model(), ost.model()
    for a, b, what in zip(mg, mc[:4], ["log mass", "unique counts", "total counts", "log effective length"]):
        assert np.array_equal(a, b), what
    assert np.array_equal(ctx.fld(), mc[4])
    assert np.array_equal(ctx.lib_counts(), ost.lib_counts())
    pg = api.normalize_alphas(eq_g, mg[0], mg[1], mg[2]); pc = orc.normalize_alphas(idx.num_refs, eq_c, mc[0], mc[1], mc[2])
    assert np.array_equal(pg, pc)
    ag, rg = ctx.em_optimize(np.exp(mg[3]), pg, api.em_opts())
    ac, rc = orc.em_optimize(eq_c, np.exp(mc[3]), pc, api.em_opts())
    assert rg["iters"] == rc["iters"] and rg["converged"] and np.array_equal(ag, ac)
    ctx.free(); ost.free()
