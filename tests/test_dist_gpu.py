"""The multi-GPU seam in C (sq_dist_* over RCCL, hip/dist.hip) on the one GPU this box has: a communicator of one rank (every collective
runs through RCCL, merging nothing), the shares of replicates by rank, and the replicate-range entry points — a bootstrap replicate /
a Gibbs chain computed as a range is byte-identical to the same replicate of a full run, which is what lets ranks split them."""
import numpy as np
import pytest
from salmon_amd import api
from conftest import random_eq_classes

pytestmark = pytest.mark.gpu


def test_rccl_communicator_of_one_rank_runs_every_collective(small_world):
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
    d.barrier()
    assert d.share(10) == (0, 10)
    d.free(); ctx.free()


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
