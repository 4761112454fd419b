"""N>1 semantics on CPU (SPEC §MG), online model enabled: two gloo ranks shard the reads, each learns its own model, the tables are
all-gathered and merged and the per-transcript state reduced (masses: logAdd in rank order through the library's
sq_merge_log_masses); the result must equal the R-rank checker (orc_state_merge) bit for bit.  The product path does the same
exchange over RCCL in C (sq_dist_*, hip/dist.hip); tests/test_dist_gpu.py drives that on the GPU box."""
import os, socket
import numpy as np
import pytest


def merge_tables_host(tables):
    """numpy reference of sq_eq_merge: key = (h1, h2); counts and wq add; output in canonical order."""
    acc = {}
    for t in tables:
        for c in range(len(t.count)):
            a, b = int(t.off[c]), int(t.off[c + 1])
            key = (int(t.h1[c]), int(t.h2[c]))
            if key not in acc:
                acc[key] = [t.tid[a:b].copy(), t.bins[a:b].copy(), t.wq[a:b].astype(object), int(t.count[c])]
            else:
                assert np.array_equal(acc[key][0], t.tid[a:b]) and np.array_equal(acc[key][1], t.bins[a:b])
                acc[key][2] = acc[key][2] + t.wq[a:b].astype(object); acc[key][3] += int(t.count[c])
    keys = sorted(acc, key=lambda k: (int(acc[k][0][0]), k[0], k[1]))   # canonical order: (first tid, h1, h2)
    return keys, [acc[k] for k in keys]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch, torch.distributed as dist
    from salmon_amd import api, synth, dist as sqdist
    import orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tx = synth.Txome(seed=9, n_genes=50, iso_per_gene=5, threads=1)
    names, seqs, lens = tx.tables()
    idx = api.SalmonIndex.build_mem_raw(tx.n, names, seqs, lens, threads=1)
    oidx = orc.OrcIndex(idx)
    N = 1200; per = N // world
    seq, off, _, _ = tx.reads(N, read_len=100, seed=3, threads=1)
    # the online model is ON (small burn-in, so the shards cross pre-burn-in -> aux params -> burned in): every rank learns its own
    # model on its own shard, exactly what SPEC §MG defines
    opts = api.quant_opts(mini_batch_size=100, num_pre_burnin_frags=80, num_burnin_frags=350)
    def run(lo, hi):
        s = seq[lo * 200: hi * 200]; o = (off[2 * lo: 2 * hi + 1] - off[2 * lo]).copy()
        rb = api.make_read_batch(s, o, hi - lo, paired=True)
        ro, aln, mt, st = orc.map_batch(oidx, opts, rb, threads=1)
        ost = orc.OrcState(oidx, opts); ost.eq_accumulate(ro, aln, st["num_with_joint_hits"]); ost.finish()
        return ost
    mine = run(rank * per, (rank + 1) * per)
    eq = mine.eq_finish(); lm, uq, tc, le, _ = mine.model()
    dev = torch.device("cpu")
    tables = sqdist.all_gather_tables(eq, dist, dev)
    lm2, uq2, tc2, le2 = sqdist.reduce_model(lm, uq, tc, le, dist, dev)
    keys, rows = merge_tables_host(tables)
    ok = True; msg = ""
    if rank == 0:
        # the R-rank checker: every rank's state, folded into rank 0's in rank order (orc_state_merge = SPEC §MG)
        states = [run(r * per, (r + 1) * per) for r in range(world)]
        assert states[0].summary()["burned_in"]
        for r in range(1, world): states[0].merge(states[r])
        full = states[0].eq_finish(); lmf, uqf, tcf, lef, _ = states[0].model()
        fk = [(int(a), int(b)) for a, b in zip(full.h1, full.h2)]
        ok = fk == keys
        for c, row in enumerate(rows):
            a, b = int(full.off[c]), int(full.off[c + 1])
            ok = ok and np.array_equal(full.tid[a:b],
                row[0]) and int(full.count[c]) == row[3] and [int(x) for x in full.wq[a:b]] == [int(x) for x in row[2]]
        ok = ok and np.array_equal(uq2, uqf) and np.array_equal(tc2, tcf) and np.array_equal(lm2, lmf) and np.array_equal(le2, lef)
        single = run(0, N)        # one rank over everything is a different realisation of the online phase (SPEC §MG) with the same totals
        ok = ok and int(single.eq_finish().count.sum()) == int(full.count.sum())
        msg = "classes=%d" % len(keys)
    # every rank must hold identical merged keys: compare a digest
    digest = torch.tensor([hash(tuple(keys)) % (2**31)], dtype=torch.int64)
    ds = [torch.zeros_like(digest) for _ in range(world)]; dist.all_gather(ds, digest)
    ok = ok and all(int(d) == int(ds[0]) for d in ds)
    q.put((rank, bool(ok), msg))
    dist.destroy_process_group()


def test_two_rank_gloo_eq_table_reduction(built):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


def test_shared_burn_in_prefix_makes_the_n_rank_table_the_one_rank_table(built):
    """[r4] SPEC §MG: every rank runs the batches up to the end of the burn-in itself (the model is learned once, as in the reference), ranks other
    than 0 then drop what the prefix counted; the rest of the batches are dealt round-robin.  Checker level (no GPU): the merged class table, the
    unique / total counts and the fragment counters of a 3-rank job equal the one-rank job's exactly; the masses are a different realisation."""
    import orc
    from salmon_amd import api, synth
    tx = synth.Txome(seed=9, n_genes=50, iso_per_gene=5, threads=1)
    names, seqs, lens = tx.tables()
    idx = api.SalmonIndex.build_mem_raw(tx.n, names, seqs, lens, threads=1); oidx = orc.OrcIndex(idx)
    N, B = 3000, 200; seq, off, _, _ = tx.reads(N, read_len=100, seed=3, threads=1)
    opts = api.quant_opts(mini_batch_size=50, num_pre_burnin_frags=80, num_burnin_frags=500)
    def batch(b):
        lo, hi = b * B, (b + 1) * B
        rb = api.make_read_batch(seq[lo * 200: hi * 200], (off[2 * lo: 2 * hi + 1] - off[2 * lo]).copy(), B, paired=True)
        ro, aln, mt, st = orc.map_batch(oidx, opts, rb, threads=2); return ro, aln, st["num_with_joint_hits"]
    batches = [batch(b) for b in range(N // B)]
    one = orc.OrcState(oidx, opts)
    for ro, aln, j in batches: one.eq_accumulate(ro, aln, j)
    one.finish(); eq1 = one.eq_finish(); lm1, uq1, tc1, le1, _ = one.model(); s1 = one.summary()
    R = 3; ranks = [orc.OrcState(oidx, opts) for _ in range(R)]; prefix = 0
    while not ranks[0].summary()["burned_in"]:          # the shared prefix: the same batches on every rank
        for st in ranks: st.eq_accumulate(*batches[prefix])
        prefix += 1
    assert 2 <= prefix < len(batches) - R and all(st.summary()["burned_in"] for st in ranks)
    for st in ranks[1:]: st.drop_counts()
    for i, b in enumerate(range(prefix, len(batches))): ranks[i % R].eq_accumulate(*batches[b])
    for st in ranks: st.finish()
    assert len(ranks[1].eq_finish().count) < len(ranks[0].eq_finish().count)               # a dropped rank holds only what it mapped after the prefix
    for st in ranks[1:]: ranks[0].merge(st)
    eqn = ranks[0].eq_finish(); lmn, uqn, tcn, len_, _ = ranks[0].model(); sn = ranks[0].summary()
    for f in ("off", "tid", "bins", "count", "wq", "h1", "h2", "w"): assert np.array_equal(getattr(eqn, f), getattr(eq1, f)), f
    assert np.array_equal(uqn, uq1) and np.array_equal(tcn, tc1) and np.array_equal(len_, le1)
    assert sn["num_assigned"] == s1["num_assigned"] and sn["num_observed"] == s1["num_observed"] and sn["num_compatible"] == s1["num_compatible"]
    assert not np.array_equal(lmn, lm1) and np.isfinite(lmn[np.isfinite(lm1)]).all()      # masses: the same transcripts carry mass, the values are another realisation
    # and the inference on the merged table starts from a different point but lands within the optimiser's tolerance
    a1, r1 = orc.em_optimize(eq1, np.exp(le1), orc.normalize_alphas(len(lm1), eq1, lm1, uq1, tc1), api.em_opts())
    an, rn = orc.em_optimize(eqn, np.exp(len_), orc.normalize_alphas(len(lmn), eqn, lmn, uqn, tcn), api.em_opts())
    big = a1 >= 10.0
    assert np.quantile(np.abs(an[big] - a1[big]) / a1[big], 0.9) < 2e-2
