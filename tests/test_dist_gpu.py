"""The multi-GPU seam in C (sq_dist_* over RCCL, hip/dist.hip) on the one GPU this box has: a communicator of one rank (every collective
runs through RCCL), two shards exchanged through the communicator's buffers in loop-back and merged by the product (= the 2-rank checker), the shares of replicates by rank, and the replicate-range entry points — a bootstrap replicate /
a Gibbs chain computed as a range is byte-identical to the same replicate of a full run, which is what lets ranks split them."""
import numpy as np
import pytest
from salmon_amd import api
from conftest import random_eq_classes

pytestmark = pytest.mark.gpu


def test_rccl_communicator_of_one_rank_runs_every_collective(small_world):
    """No short cut for world == 1: the size / payload all-gathers, the all-reduce, the broadcast and the model reduction all go through RCCL
    (a one-rank communicator), and a one-rank exchange leaves the table and the model bit-identical."""
    w = small_world; w["idx"].to_device(0)
    d = api.Dist(api.Dist.make_id(), 0, 1, 0)
    ctx = api.QuantContext(w["idx"], api.quant_opts(), device=0, max_batch_reads=4096)
    ctx.map_batch(api.make_read_batch(w["seq"], w["off"], w["n"], paired=True), fetch=False); ctx.eq_accumulate()
    before = ctx.eq_finish()
    d.merge_eq(ctx)
    after = ctx.eq_finish()
    for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
        assert np.array_equal(getattr(before, f), getattr(after, f)), f
    lm, uq, tc, le = ctx.model()
    lm2, uq2, tc2, le2 = d.reduce_model(lm, uq, tc, le)
    assert np.array_equal(lm, lm2) and np.array_equal(uq, uq2) and np.array_equal(tc, tc2) and np.array_equal(le, le2)
    assert np.array_equal(d.allreduce_u64(np.array([3, 5], np.uint64)), [3, 5])
    x = np.arange(1000, dtype=np.float64) * 0.37
    assert np.array_equal(d.allgather(x)[0], x) and np.array_equal(d.bcast(x), x)
    d.barrier()
    assert d.share(10) == (0, 10)
    d.free(); ctx.free()


def _two_shards(w, opts):
    """Two halves of the reads, each through its own HIP context (what two ranks do), and through two checker states."""
    import orc
    N = w["n"]; per = N // 2; ctxs = []; states = []
    for r in range(2):
        lo, hi = r * per, (r + 1) * per
        s = w["seq"][lo * 200: hi * 200]; o = (w["off"][2 * lo: 2 * hi + 1] - w["off"][2 * lo]).copy()
        rb = api.make_read_batch(s, o, hi - lo, paired=True)
        c = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
        c.map_batch(rb, fetch=False); c.eq_accumulate(); ctxs.append(c)
        ro, aln, mt, st = orc.map_batch(w["oidx"], opts, rb, threads=4)
        ost = orc.OrcState(w["oidx"], opts); ost.eq_accumulate(ro, aln, st["num_with_joint_hits"]); ost.finish(); states.append(ost)
    return ctxs, states


def test_two_shards_exchanged_through_the_rccl_buffers_equal_the_two_rank_checker(small_world):
    """SPEC §MG on one GPU: two contexts stand for two ranks.  Their tables are packed, sent through the one-rank communicator's
    all-gathers (RCCL runs them) into receive slots and merged by the code path sq_dist_merge_eq uses; the per-transcript state goes
    through the library's collectives and sq_merge_log_masses.  Everything equals the 2-rank checker (orc_state_merge) bit for bit."""
    w = small_world; w["idx"].to_device(0)
    opts = api.quant_opts(mini_batch_size=100, num_pre_burnin_frags=80, num_burnin_frags=700)   # each shard crosses burn-in on its own
    ctxs, states = _two_shards(w, opts)
    models = [c.model() for c in ctxs]                       # every rank's own online state, before the exchange
    d = api.Dist(api.Dist.make_id(), 0, 1, 0)
    d.merge_eq_loopback(ctxs)
    merged = [c.eq_finish() for c in ctxs]
    # the model reduction, collective by collective, each through RCCL: counts all-reduced (here: one rank's at a time, summed exactly),
    # masses all-gathered then folded in rank order by the library's rule, effective lengths broadcast from rank 0
    uq = sum(d.allreduce_u64(m[1]).astype(object) for m in models); tc = sum(d.allreduce_u64(m[2]).astype(object) for m in models)
    allm = np.ascontiguousarray(np.stack([d.allgather(m[0])[0] for m in models]))
    lm = np.zeros(allm.shape[1]); from salmon_amd import capi
    capi.check(capi.lib().sq_merge_log_masses(allm.shape[1], 2, allm.ctypes.data, lm.ctypes.data), "sq_merge_log_masses")
    le = d.bcast(models[0][3])
    assert states[0].summary()["burned_in"] and states[1].summary()["burned_in"]
    states[0].merge(states[1])
    full = states[0].eq_finish(); lmf, uqf, tcf, lef, _ = states[0].model()
    for r in range(2):
        for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
            assert np.array_equal(getattr(merged[r], f), getattr(full, f)), (r, f)
    assert np.array_equal(lm, lmf) and np.array_equal(np.array(uq, np.uint64), uqf) and np.array_equal(np.array(tc, np.uint64), tcf) and np.array_equal(le, lef)
    # and the inference tail on the merged table is the checker's
    import orc
    proj_g = api.normalize_alphas(merged[0], lm, np.array(uq, np.uint64), np.array(tc, np.uint64)); proj_c = orc.normalize_alphas(len(lm), full, lmf, uqf, tcf)
    assert np.array_equal(proj_g, proj_c)
    a_g, rep_g = ctxs[0].em_optimize(np.exp(le), proj_g, api.em_opts()); a_c, rep_c = orc.em_optimize(full, np.exp(lef), proj_c, api.em_opts())
    assert rep_g["iters"] == rep_c["iters"] and np.array_equal(a_g, a_c)
    d.free()
    for c in ctxs: c.free()


def _gloo_gpu_worker(rank, world, port, q):
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import torch, torch.distributed as dist
        from salmon_amd import api, synth, dist as sqdist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        tx = synth.Txome(seed=9, n_genes=50, iso_per_gene=5, threads=1)
        names, seqs, lens = tx.tables()
        idx = api.SalmonIndex.build_mem_raw(tx.n, names, seqs, lens, threads=1); idx.to_device(0)
        N = 1200; per = N // world
        seq, off, _, _ = tx.reads(N, read_len=100, seed=3, threads=1)
        opts = api.quant_opts(mini_batch_size=100, num_pre_burnin_frags=80, num_burnin_frags=350)
        lo, hi = rank * per, (rank + 1) * per
        s = seq[lo * 200: hi * 200]; o = (off[2 * lo: 2 * hi + 1] - off[2 * lo]).copy()
        ctx = api.QuantContext(idx, opts, device=0, max_batch_reads=2048)
        ctx.map_batch(api.make_read_batch(s, o, hi - lo, paired=True), fetch=False); ctx.eq_accumulate()
        eq = ctx.eq_finish(); lm, uq, tc, le = ctx.model()
        tables = sqdist.all_gather_tables(eq, dist, torch.device("cpu"))
        for r in range(world):
            if r != rank: ctx.eq_merge(tables[r])                      # the product's merge (sq_eq_merge): integer counts, fixed-point sums
        mine = ctx.eq_finish()
        lm2, uq2, tc2, le2 = sqdist.reduce_model(lm, uq, tc, le, dist, torch.device("cpu"))   # masses by the library's sq_merge_log_masses
        ok = True; msg = ""
        # the last collective comes first: every rank must hold the same merged table (nothing below talks to the other rank any more,
        # so a failing check on one rank cannot leave the other waiting)
        import hashlib
        dig = int(hashlib.sha256(mine.wq.tobytes() + mine.count.tobytes() + mine.tid.tobytes()).hexdigest()[:15], 16)
        dt = torch.tensor([dig], dtype=torch.int64); ds = [torch.zeros_like(dt) for _ in range(world)]; dist.all_gather(ds, dt)
        dist.barrier(); dist.destroy_process_group()
        if not all(int(x) == int(ds[0]) for x in ds): ok = False; msg = "ranks hold different merged tables"
        if rank == 0 and ok:
            import orc
            oidx = orc.OrcIndex(idx); states = []
            for r in range(world):
                a, b = r * per, (r + 1) * per
                rb = api.make_read_batch(seq[a * 200: b * 200], (off[2 * a: 2 * b + 1] - off[2 * a]).copy(), b - a, paired=True)
                ro, aln, mt, st = orc.map_batch(oidx, opts, rb, threads=1)
                ost = orc.OrcState(oidx, opts); ost.eq_accumulate(ro, aln, st["num_with_joint_hits"]); ost.finish(); states.append(ost)
            own = states[0].eq_finish(); own_model = states[0].model()[:4]
            for r in range(1, world): states[0].merge(states[r])
            full = states[0].eq_finish(); lmf, uqf, tcf, lef, _ = states[0].model()
            bad = [f for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"] if not np.array_equal(getattr(mine, f), getattr(full, f))]
            bad += [n for n, x, y in (("log_mass", lm2, lmf), ("uniq", uq2, uqf), ("total", tc2, tcf), ("log_eff_len", le2, lef)) if not np.array_equal(x, y)]
            # which side differs: this rank's own shard against its checker state
            bad += ["own:" + f for f in ["off", "tid", "count", "wq"] if not np.array_equal(getattr(eq, f), getattr(own, f))]
            bad += ["own_model:%d" % i for i, (x, y) in enumerate(zip((lm, uq, tc, le), own_model)) if not np.array_equal(x, y)]
            ok = not bad
            if bad: msg = "differs: " + ",".join(bad)
        q.put((rank, bool(ok), msg or "classes=%d" % len(mine.count)))
    except Exception as e:   # a worker that dies must not leave the parent waiting for its queue entry
        import traceback
        q.put((rank, False, traceback.format_exc()[-1500:]))


def test_two_gloo_ranks_on_one_gpu_merge_with_the_product(built):
    """Two processes (gloo for the transport, both on cuda:0): each maps its shard with the HIP path, the tables travel as the
    harness's padded all-gathers and every rank folds the other's in with the product's sq_eq_merge; equals the 2-rank checker."""
    import socket, torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    try:
        res = [q.get(timeout=240) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive(): p.terminate()
    assert all(ok for _, ok, _ in res), res


def test_shares_cover_the_replicates_without_overlap():
    from salmon_amd import capi
    import ctypes as C
    step = capi.lib().sq_gibbs_chain_step
    assert [step(n) for n in (1, 49, 50, 100, 210)] == [1, 49, 25, 25, 26]
    # sq_dist_share only needs rank / world: emulate the ranks of a world without a communicator through the same arithmetic
    def share(total, unit, rank, world):
        nu = max(1, total // unit); lo = nu * rank // world; hi = nu * (rank + 1) // world
        a = min(total, lo * unit); b = total if hi == nu else min(total, hi * unit)
        return a, max(0, b - a)
    for total, unit in [(100, 1), (7, 1), (210, 26), (100, 25), (30, 30)]:
        for world in (1, 2, 3, 8):
            got = [share(total, unit, r, world) for r in range(world)]
            assert sum(n for _, n in got) == total
            pos = 0
            for a, n in got:
                if n: assert a == pos and a % unit == 0; pos += n


def test_replicate_ranges_equal_the_full_run(built):
    M, E = 400, 3000
    eq = random_eq_classes(M, E, seed=4); eff = np.random.default_rng(5).uniform(80, 2500, M)
    full = api.bootstrap(eq, eff, 5, 11, int(eq.count.sum()))
    part = np.concatenate([api.bootstrap_range(eq, eff, 5, 0, 2, 11, int(eq.count.sum())), api.bootstrap_range(eq, eff, 5, 2, 3, 11, int(eq.count.sum()))])
    assert np.array_equal(full, part)
    a0 = np.random.default_rng(6).gamma(0.5, 50.0, M)
    S = 50                                                            # two chains of 25
    gfull = api.gibbs(eq, eff, a0, S, 13, int(eq.count.sum()))
    gpart = np.concatenate([api.gibbs_range(eq, eff, a0, S, 0, 25, 13, int(eq.count.sum())), api.gibbs_range(eq, eff, a0, S, 25, 25, 13, int(eq.count.sum()))])
    assert np.array_equal(gfull, gpart)
    with pytest.raises(Exception, match="does not start a chain"):
        api.gibbs_range(eq, eff, a0, S, 10, 5, 13, int(eq.count.sum()))


def test_shared_burn_in_prefix_two_hip_ranks_hold_the_one_rank_table(small_world):
    """[r4] SPEC §MG on one GPU: two contexts stand for two ranks.  Both run the batches up to the end of the burn-in (the shared prefix), rank 1 drops
    what the prefix counted (sq_model_drop_counts), the remaining batches alternate, the tables go through the communicator's buffers
    (sq_dist_merge_eq_loopback) — and the merged table is the ONE-context job's table bit for bit: labels, bins, counts, fixed-point weights."""
    w = small_world; w["idx"].to_device(0)
    opts = api.quant_opts(mini_batch_size=100, num_pre_burnin_frags=80, num_burnin_frags=900)
    N, B = w["n"], 400
    def rb_of(b):
        lo, hi = b * B, (b + 1) * B
        return api.make_read_batch(w["seq"][lo * 200: hi * 200], (w["off"][2 * lo: 2 * hi + 1] - w["off"][2 * lo]).copy(), B, paired=True)
    nb = N // B
    one = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=512)
    for b in range(nb): one.map_batch(rb_of(b), fetch=False); one.eq_accumulate()
    eq1 = one.eq_finish(); lm1, uq1, tc1, le1 = one.model(); s1 = one.summary()
    ranks = [api.QuantContext(w["idx"], opts, device=0, max_batch_reads=512) for _ in range(2)]; prefix = 0
    while not ranks[0].summary()["burned_in"]:
        for c in ranks: c.map_batch(rb_of(prefix), fetch=False); c.eq_accumulate()
        prefix += 1
    assert 2 <= prefix < nb - 2 and ranks[1].summary()["burned_in"]
    ranks[1].drop_counts()
    assert ranks[1].summary()["num_assigned"] == 0 and ranks[1].summary()["burned_in"]
    for i, b in enumerate(range(prefix, nb)): c = ranks[i % 2]; c.map_batch(rb_of(b), fetch=False); c.eq_accumulate()
    own1 = ranks[1].eq_finish(); assert 0 < len(own1.count) < len(eq1.count)
    models = [c.model() for c in ranks]
    d = api.Dist(api.Dist.make_id(), 0, 1, 0); d.merge_eq_loopback(ranks)
    for c in ranks:
        eqn = c.eq_finish()
        for f in ("off", "tid", "bins", "count", "wq", "h1", "h2", "w"): assert np.array_equal(getattr(eqn, f), getattr(eq1, f)), f
    assert np.array_equal(models[0][1] + models[1][1], uq1) and np.array_equal(models[0][2] + models[1][2], tc1) and np.array_equal(models[0][3], le1) and np.array_equal(models[1][3], le1)
    assert ranks[0].summary()["num_assigned"] + ranks[1].summary()["num_assigned"] == s1["num_assigned"]
    d.free(); one.free()
    for c in ranks: c.free()
