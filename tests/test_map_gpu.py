"""GPU mapping pipeline vs the CPU checker, stage by stage and end to end (bit-exact; -m gpu)."""
import numpy as np
import pytest
from salmon_amd import api, capi, synth
import orc

pytestmark = pytest.mark.gpu


def _fields_equal(a, b, fields, what):
    assert len(a) == len(b), "%s: %d vs %d records" % (what, len(a), len(b))
    for f in fields:
        if not np.array_equal(a[f], b[f]):
            i = int(np.nonzero(a[f] != b[f])[0][0])
            raise AssertionError("%s: field %s differs first at %d: gpu=%s cpu=%s" % (what, f, i, a[i], b[i]))


@pytest.fixture(scope="module")
def gpu_world(small_world):
    w = small_world
    w["idx"].to_device(0)
    opts = api.quant_opts()
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=8192)
    rb = api.make_read_batch(w["seq"], w["off"], w["n"], paired=True)
    return dict(w=w, opts=opts, ctx=ctx, rb=rb)


def test_stages_match_checker(gpu_world):
    g = gpu_world; w = g["w"]
    ro, aln, mt, st = g["ctx"].map_batch(g["rb"])
    um_c, mm_c, ch_c, cd_c = orc.map_taps(w["oidx"], g["opts"], g["rb"])
    um_g = g["ctx"].tap(capi.lib and 1, api.UNIMEM_DTYPE)
    _fields_equal(um_g, um_c, ["end", "qpos", "len", "unitig", "uoff", "fw"], "uni-MEMs")
    mm_g = g["ctx"].tap(2, api.MEM_DTYPE)
    _fields_equal(mm_g, mm_c, ["end", "tid", "rpos", "qpos", "len", "fw"], "MEMs")
    ch_g = g["ctx"].tap(3, api.CHAIN_DTYPE)
    _fields_equal(ch_g, ch_c, ["end", "tid", "pos", "last_end", "fw", "n_mems", "score"], "chains")
    cd_g = g["ctx"].tap(4, api.CAND_DTYPE)
    _fields_equal(cd_g, cd_c, ["frag", "tid", "lpos", "rpos", "lfw", "rfw", "mate_status", "valid", "lscore", "rscore", "frag_len"], "candidates")


def test_alignments_and_stats_match_checker(gpu_world):
    g = gpu_world; w = g["w"]
    ro_g, aln_g, mt_g, st_g = g["ctx"].map_batch(g["rb"])
    ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], g["opts"], g["rb"], threads=4)
    assert np.array_equal(ro_g, ro_c)
    assert np.array_equal(mt_g, mt_c)
    _fields_equal(aln_g, aln_c, list(api.ALN_DTYPE.names), "alignments")
    assert st_g == st_c
    assert st_g["num_mapped"] > 0.9 * w["n"] * 0.98


def test_small_and_ragged_batches(gpu_world):
    # empty batch, a single pair, reads shorter than k, reads with N, unequal lengths
    g = gpu_world; w = g["w"]
    rb0 = api.make_read_batch(np.zeros(1, np.uint8), np.zeros(1, np.uint64), 0, paired=True)
    ro, aln, mt, st = g["ctx"].map_batch(rb0)
    assert len(aln) == 0 and st["num_reads"] == 0
    seqs = []
    base = bytes(w["seq"][:200].tobytes())
    r1, r2 = base[:100], base[100:200]
    seqs += [r1, r2]                                   # ordinary pair
    seqs += [r1[:20], r2[:25]]                         # both shorter than k
    seqs += [r1[:60] + b"N" + r1[61:], r2]             # an N in the middle
    seqs += [b"N" * 100, r2]                           # all-N mate -> orphan
    seqs += [r1[:75], r2[:90]]                         # ragged lengths
    seqs += [r1.lower(), r2]                           # lower case
    seqs += [b"", r2]                                  # empty mate
    seq = np.frombuffer(b"".join(seqs), np.uint8).copy()
    off = np.zeros(len(seqs) + 1, np.uint64); off[1:] = np.cumsum([len(s) for s in seqs])
    rb = api.make_read_batch(seq, off, len(seqs) // 2, paired=True)
    ro_g, aln_g, mt_g, st_g = g["ctx"].map_batch(rb)
    ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], g["opts"], rb, threads=1)
    assert np.array_equal(ro_g, ro_c) and np.array_equal(mt_g, mt_c) and st_g == st_c
    _fields_equal(aln_g, aln_c, list(api.ALN_DTYPE.names), "alignments")
    assert ro_g[1] - ro_g[0] >= 1 and ro_g[2] - ro_g[1] == 0


@pytest.mark.parametrize("libtype", ["IU", "ISF", "ISR"])
def test_library_types(small_world, libtype):
    w = small_world
    opts = api.set_libtype(api.quant_opts(), libtype)
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=2048)
    rb = api.make_read_batch(w["seq"], w["off"], 1500, paired=True)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
    ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=4)
    assert np.array_equal(ro_g, ro_c) and st_g == st_c
    _fields_equal(aln_g, aln_c, list(api.ALN_DTYPE.names), "alignments")
    ctx.free()


def test_single_end(small_world):
    w = small_world
    opts = api.set_libtype(api.quant_opts(), "U")
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
    rb = api.make_read_batch(w["seq"], w["off"], 3000, paired=False)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
    ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=4)
    assert np.array_equal(ro_g, ro_c) and st_g == st_c
    _fields_equal(aln_g, aln_c, list(api.ALN_DTYPE.names), "alignments")
    ctx.free()


@pytest.mark.parametrize("lt", ["U", "SF"])
def test_single_end_online_model_and_eq_classes(small_world, lt):
    # single-end libraries take the ambiguous-fragment-length branch (LogCMFCache::getAmbigFragLengthProb with the live,
    # uncached FLD::cmf before burn-in and the cached one after): model, eq-classes and weights must match the checker
    w = small_world
    opts = api.set_libtype(api.quant_opts(mini_batch_size=500, num_pre_burnin_frags=400, num_burnin_frags=2500), lt)
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
    ost = orc.OrcState(w["oidx"], opts)
    for lo, hi in [(0, 2200), (2200, 6000)]:
        seq = w["seq"][lo * 100: hi * 100]; off = (w["off"][lo: hi + 1] - w["off"][lo]).copy()
        rb = api.make_read_batch(seq, off, hi - lo, paired=False)
        ctx.map_batch(rb, fetch=False); ctx.eq_accumulate()
        ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=4)
        ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"])
    ost.finish()
    assert ctx.summary() == ost.summary() and ctx.summary()["burned_in"]
    for a, b in zip(ctx.model(), ost.model()[:4]):
        assert np.array_equal(a, b)
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
        assert np.array_equal(getattr(eq_g, f), getattr(eq_c, f)), f
    ctx.free(); ost.free()


def test_online_model_and_eq_classes_match_checker(small_world):
    w = small_world
    # small burn-in so the test crosses pre-burn-in -> aux params -> burned-in regimes
    opts = api.quant_opts(mini_batch_size=500, num_pre_burnin_frags=400, num_burnin_frags=2200)
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
    ost = orc.OrcState(w["oidx"], opts)
    for lo, hi in [(0, 1700), (1700, 4000)]:
        seq = w["seq"][lo * 200: hi * 200]; off = (w["off"][2 * lo: 2 * hi + 1] - w["off"][2 * lo]).copy()
        rb = api.make_read_batch(seq, off, hi - lo, paired=True)
        ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
        ctx.eq_accumulate()
        ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=4)
        ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"])
    ost.finish()
    sg, sc = ctx.summary(), ost.summary()
    assert sg == sc and sg["burned_in"]
    lm_g, uq_g, tc_g, le_g = ctx.model()
    lm_c, uq_c, tc_c, le_c, fld_c = ost.model()
    assert np.array_equal(uq_g, uq_c) and np.array_equal(tc_g, tc_c)
    assert np.array_equal(lm_g, lm_c), float(np.nanmax(np.abs(np.where(np.isinf(lm_c), 0, lm_g - lm_c))))
    assert np.array_equal(le_g, le_c)
    assert np.array_equal(ctx.fld(), fld_c)
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
        assert np.array_equal(getattr(eq_g, f), getattr(eq_c, f)), f
    assert int(eq_g.count.sum()) == sg["num_assigned"]
    ctx.free(); ost.free()


def test_eq_merge_is_exact_and_order_free(small_world):
    # shard the reads over two contexts, merge the tables both ways: identical bits (multi-GPU reduction)
    w = small_world
    opts = api.quant_opts(num_burnin_frags=10**9, num_pre_burnin_frags=10**9)   # model-independent weights
    def run(lo, hi):
        ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
        seq = w["seq"][lo * 200: hi * 200]; off = (w["off"][2 * lo: 2 * hi + 1] - w["off"][2 * lo]).copy()
        ctx.map_batch(api.make_read_batch(seq, off, hi - lo, paired=True), fetch=False); ctx.eq_accumulate()
        return ctx
    a, b, full = run(0, 2000), run(2000, 4000), run(0, 4000)
    ea, eb = a.eq_finish(), b.eq_finish()
    a.eq_merge(eb); b.eq_merge(ea)
    m1, m2, ef = a.eq_finish(), b.eq_finish(), full.eq_finish()
    for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
        assert np.array_equal(getattr(m1, f), getattr(m2, f)), f
        assert np.array_equal(getattr(m1, f), getattr(ef, f)), f
    for c in (a, b, full):
        c.free()


def test_pipelined_lanes_match_sequential(small_world):
    # sq_map_submit / sq_map_wait (two batches in flight on two lanes) must give the same alignments, model and
    # eq-classes as sq_map_batch called batch by batch
    w = small_world
    opts = api.quant_opts(mini_batch_size=500, num_pre_burnin_frags=400, num_burnin_frags=2200)
    cuts = [(0, 900), (900, 2100), (2100, 2600), (2600, 4000)]
    keep = []   # a read batch holds raw pointers: the arrays must outlive it
    def rbs():
        out = []
        for lo, hi in cuts:
            seq = w["seq"][lo * 200: hi * 200]; off = (w["off"][2 * lo: 2 * hi + 1] - w["off"][2 * lo]).copy()
            keep.append((seq, off)); out.append(api.make_read_batch(seq, off, hi - lo, paired=True))
        return out
    seq_ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
    want = []
    for rb in rbs():
        ro, aln, mt, st = seq_ctx.map_batch(rb); seq_ctx.eq_accumulate(); want.append((ro, aln, mt, st))
    pip = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
    batches = rbs()
    pip.map_submit(batches[0], fetch=True); pip.map_submit(batches[1], fetch=True)
    for i in range(len(batches)):
        ro, aln, mt, st = pip.map_wait()
        assert st == want[i][3]
        assert np.array_equal(ro, want[i][0]) and np.array_equal(mt, want[i][2])
        _fields_equal(aln, want[i][1], list(api.ALN_DTYPE.names), "alignments of batch %d" % i)
        if i in (1, 2):   # taps refer to the batch just waited for (lane 1, then lane 0)
            assert len(pip.tap(3, api.CHAIN_DTYPE)) == st["num_chains"]
        pip.eq_accumulate()
        if i + 2 < len(batches):
            pip.map_submit(batches[i + 2], fetch=True)
    assert pip.summary() == seq_ctx.summary()
    for a, b in zip(pip.model(), seq_ctx.model()):
        assert np.array_equal(a, b)
    e1, e2 = pip.eq_finish(), seq_ctx.eq_finish()
    for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
        assert np.array_equal(getattr(e1, f), getattr(e2, f)), f
    pip.free(); seq_ctx.free()


@pytest.mark.parametrize("read_len,sub,indel", [(150, 0.02, 0.004), (250, 0.01, 0.002), (75, 0.03, 0.0)])
def test_noisy_and_long_reads_match_checker(small_world, read_len, sub, indel):
    # error-rich reads drive the banded DP (global + extension regions, early exits) and the mismatch-skip walk;
    # 150 / 250 bp exercise the multi-word read packing (256 bases max per end)
    w = small_world
    N = 1500
    seq, off, _, _ = w["tx"].reads(N, read_len=read_len, seed=77 + read_len, sub_rate=sub, indel_rate=indel, threads=4)
    opts = api.quant_opts()
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=2048)
    rb = api.make_read_batch(seq, off, N, paired=True)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
    ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=4)
    assert st_g == st_c and st_g["num_dp_alignments"] > 0
    assert np.array_equal(ro_g, ro_c) and np.array_equal(mt_g, mt_c)
    _fields_equal(aln_g, aln_c, list(api.ALN_DTYPE.names), "alignments")
    ctx.free()


def test_device_resident_merge_matches_host_merge(small_world):
    # multi-GPU reduction without host staging: export as device pointers (wrapped zero-copy by torch), copy as an
    # all_gather would, merge with sq_eq_merge_device; must equal the host-table merge bit for bit
    import torch
    from salmon_amd.dist import _DevArray, FIELDS
    w = small_world
    opts = api.quant_opts(num_burnin_frags=10**9, num_pre_burnin_frags=10**9)
    def run(lo, hi):
        ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
        seq = w["seq"][lo * 200: hi * 200]; off = (w["off"][2 * lo: 2 * hi + 1] - w["off"][2 * lo]).copy()
        ctx.map_batch(api.make_read_batch(seq, off, hi - lo, paired=True), fetch=False); ctx.eq_accumulate()
        return ctx
    a, b, a2 = run(0, 2000), run(2000, 4000), run(0, 2000)
    ex = b.eq_export_device(); dev = torch.device("cuda", 0)
    recv = {}
    for f in FIELDS:
        ptr, n, dt = ex[f]; is64 = np.dtype(dt).itemsize == 8
        view = torch.as_tensor(_DevArray(ptr, n, "<i8" if is64 else "<i4"), device=dev)
        recv[f] = view.clone()                                   # stands in for the all_gather receive buffer
    hb = b.eq_finish()
    assert np.array_equal(recv["tid"].cpu().numpy().view(np.uint32), hb.tid) and np.array_equal(recv["h1"].cpu().numpy().view(np.uint64), hb.h1)
    a.eq_merge_device(ex["E"], ex["L"], {f: recv[f].data_ptr() for f in FIELDS})
    a2.eq_merge(hb)
    m1, m2 = a.eq_finish(), a2.eq_finish()
    for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
        assert np.array_equal(getattr(m1, f), getattr(m2, f)), f
    for c in (a, b, a2):
        c.free()


def test_decoy_aware_mapping_matches_checker(built):
    # decoys (SalmonMappingUtils.hpp:82-151,407-485): sequences listed after the targets; a fragment whose best hit is a decoy is
    # counted as a decoy fragment and target hits below decoyThreshold x bestDecoy are dropped.  Decoys here embed mutated
    # copies of transcripts between random flanks, so reads drawn from a decoy hit both the decoy and the transcript.
    rng = np.random.default_rng(5)
    tx = synth.Txome(seed=13, n_genes=40, iso_per_gene=3, threads=2)
    names = list(tx.names()); seqs = [s.decode() for s in tx.seqs()]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    def rnd(n): return "".join(rng.choice(list("ACGT"), n))
    def mutate(s, rate): return "".join((rng.choice([c for c in "ACGT" if c != ch]) if rng.random() < rate else ch) for ch in s)
    decoys = [rnd(400) + mutate(seqs[j], 0.01) + rnd(400) for j in rng.choice(len(seqs), 12, replace=False)]
    first_decoy = len(seqs)
    idx = api.SalmonIndex.build_mem(names + ["decoy%d" % i for i in range(len(decoys))], seqs + decoys, threads=2, first_decoy=first_decoy,
        keep_duplicates=True)
    oidx = orc.OrcIndex(idx)
    recs = []
    def pair(s):
        fl = int(rng.integers(180, 320)); p = int(rng.integers(0, len(s) - fl))
        r1 = s[p:p + 100]; r2 = "".join(comp[c] for c in reversed(s[p + fl - 100:p + fl]))
        return (r1, r2) if rng.random() < 0.5 else (r2, r1)
    for _ in range(1500): recs += pair(decoys[int(rng.integers(len(decoys)))])        # decoy-derived fragments
    for _ in range(1500): recs += pair(seqs[int(rng.integers(len(seqs)))])            # transcript-derived fragments
    seq = np.frombuffer("".join(recs).encode(), np.uint8).copy(); off = np.arange(0, len(recs) + 1, dtype=np.uint64) * np.uint64(100)
    n = len(recs) // 2
    opts = api.quant_opts(mini_batch_size=500, num_pre_burnin_frags=400, num_burnin_frags=1800)
    ctx = api.QuantContext(idx, opts, device=0, max_batch_reads=4096)
    rb = api.make_read_batch(seq, off, n, paired=True)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
    ro_c, aln_c, mt_c, st_c = orc.map_batch(oidx, opts, rb, threads=4)
    assert st_g == st_c and st_g["num_decoy_fragments"] > 200
    assert np.array_equal(ro_g, ro_c) and np.array_equal(mt_g, mt_c)
    _fields_equal(aln_g, aln_c, list(api.ALN_DTYPE.names), "alignments")
    assert np.all(aln_g["tid"] < first_decoy)            # decoys never reach the alignment lists
    ctx.eq_accumulate()
    ost = orc.OrcState(oidx, opts); ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"]); ost.finish()
    assert ctx.summary() == ost.summary()
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
        assert np.array_equal(getattr(eq_g, f), getattr(eq_c, f)), f
    ctx.free(); ost.free()


OPTION_VARIANTS = [
    dict(hard_filter=1), dict(allow_dovetail=1), dict(allow_orphans=0), dict(disable_chaining_heuristic=1),
    dict(range_factorization_bins=0), dict(range_factorization_bins=8), dict(consensus_slack=0.1, min_score_fraction=0.8),
    dict(mismatch_seed_skip=5, max_occs_per_hit=20), dict(match_score=1, mismatch_penalty=-3, gap_open=4, gap_extend=1, bandwidth=8),
    dict(score_exp=2.0, min_aln_prob=1e-3, decoy_threshold=0.9), dict(no_length_correction=1), dict(no_eff_length_correction=1),
    dict(use_frag_len_dist=0), dict(model_single_frag_prob=0), dict(ignore_incompat=0, incompat_prior=-20.0, _lib="ISF"),
    dict(pre_merge_chain_sub_thresh=0.9, post_merge_chain_sub_thresh=0.95, orphan_chain_sub_thresh=0.5), dict(frag_len_max=400, fld_mean=200.0,
        fld_sd=40.0),
    dict(forgetting_factor=0.8, seed=12345), dict(recover_orphans=1), dict(recover_orphans=1, max_read_occs=2, allow_dovetail=1),
    dict(_mimic="bt2"), dict(_mimic="strict"),      # --mimicBT2 / --mimicStrictBT2 (QuantOptionsUtils.cpp:256-294)
]


@pytest.mark.parametrize("variant", OPTION_VARIANTS, ids=lambda v: ",".join("%s=%s" % kv for kv in v.items()))
def test_option_variants_match_checker(small_world, variant):
    # every mapping / model option the C ABI exposes (the SalmonOpts fields initMapperSettings and processMiniBatch read),
    # away from its default: alignments, counters, online model and eq-classes must still equal the checker's
    w = small_world
    kw = {k: v for k, v in variant.items() if not k.startswith("_")}
    opts = api.quant_opts(mini_batch_size=500, num_pre_burnin_frags=400, num_burnin_frags=1500, **kw)
    if "_lib" in variant: api.set_libtype(opts, variant["_lib"])
    if "_mimic" in variant: api.mimic_bt2(opts, strict=variant["_mimic"] == "strict")
    N = 2500
    seq, off, _, _ = w["tx"].reads(N, read_len=100, seed=909, sub_rate=0.015, indel_rate=0.002, threads=4)
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
    rb = api.make_read_batch(seq, off, N, paired=True)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
    ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=4)
    assert st_g == st_c
    assert np.array_equal(ro_g, ro_c) and np.array_equal(mt_g, mt_c)
    _fields_equal(aln_g, aln_c, list(api.ALN_DTYPE.names), "alignments")
    ctx.eq_accumulate()
    ost = orc.OrcState(w["oidx"], opts); ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"]); ost.finish()
    assert ctx.summary() == ost.summary()
    assert np.array_equal(ctx.lib_counts(), ost.lib_counts())      # fragments per observed library format
    for a, b in zip(ctx.model(), ost.model()[:4]):
        assert np.array_equal(a, b)
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
        assert np.array_equal(getattr(eq_g, f), getattr(eq_c, f)), f
    ctx.free(); ost.free()


def test_limits_capacity_and_long_reads(small_world):
    w = small_world
    opts = api.quant_opts()
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=512)
    # a batch larger than the context was created for is refused, not truncated
    rb = api.make_read_batch(w["seq"], w["off"], 600, paired=True)
    with pytest.raises(Exception) as ei:
        ctx.map_batch(rb)
    assert "exceeds ctx capacity" in str(ei.value)
    # [r4] read ends longer than 256 bases are mapped whole: the context widens its packing stride (tests/test_long_reads.py has the details)
    seq, off, _, _ = w["tx"].reads(300, read_len=300, seed=5, threads=2)
    rb = api.make_read_batch(seq, off, 300, paired=True)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
    ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=2)
    assert st_g == st_c and np.array_equal(ro_g, ro_c)
    _fields_equal(aln_g, aln_c, list(api.ALN_DTYPE.names), "alignments")
    assert len(aln_g) > 0 and int(aln_g["read_len"].max()) == 300 and st_g["num_truncated_ends"] == 0
    ctx.free()


def test_reserved_class_table_and_workspace(small_world):
    # sq_ctx_reserve: a larger class table (made while empty), pre-sized export buffers and EM workspace; results are
    # unchanged (canonical class order does not depend on the table size); reserving more once classes exist is refused
    w = small_world
    opts = api.quant_opts()
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
    ctx.reserve(4_000_000, 20_000_000)
    rb = api.make_read_batch(w["seq"], w["off"], w["n"], paired=True)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb); ctx.eq_accumulate()
    ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=4)
    ost = orc.OrcState(w["oidx"], opts); ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"]); ost.finish()
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
        assert np.array_equal(getattr(eq_g, f), getattr(eq_c, f)), f
    with pytest.raises(Exception, match="already holds"):
        ctx.reserve(12_000_000, 0)
    lm, uq, tc, le = ctx.model()
    proj = api.normalize_alphas(eq_g, lm, uq, tc)
    a1, r1 = ctx.em_optimize(np.exp(le), proj, api.em_opts())          # ctx workspace (arena), lent stream
    a2, r2 = api.em_optimize(eq_g, np.exp(le), proj, api.em_opts(), device=0)   # private workspace
    a3, r3 = ctx.em_optimize(np.exp(le), proj, api.em_opts())          # the arena is reused
    assert r1["iters"] == r2["iters"] == r3["iters"] and np.array_equal(a1, a2) and np.array_equal(a1, a3)
    ctx.reset(); ctx.reserve(0, 0)                                     # after a reset the table is empty again
    ctx.free(); ost.free()


def test_repeat_families_cover_every_mem_size_class(built):
    # The fused projection / sort / chaining kernels pick a lane-group size by the MEM count of a read end (mem_kernels.h:
    # <= 16 / 32 / 64 in 16-lane groups, <= 256 and <= 1024 one wave per end, larger -> compact radix-sort path).  Two repeat families put reads in every class: family A (900
    # transcripts share a 160-base element that comes in two variants differing by one base: a read across the variant
    # site collects three uni-MEMs of 900 / ~450 / 900 occurrences = ~2250 MEMs), family B (150 copies of another element).
    rng = np.random.default_rng(17)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    def rnd(n): return "".join(rng.choice(list("ACGT"), n))
    repA = rnd(160); repA2 = repA[:80] + comp[repA[80]] + repA[81:]; repB = rnd(160)
    seqs = []
    for i in range(1300):
        body = rnd(int(rng.integers(500, 900)))
        if i < 900: body = body[:250] + (repA if i % 2 else repA2) + body[250:]
        elif i < 1050: body = body[:250] + repB + body[250:]
        seqs.append(body)
    names = ["t%d" % i for i in range(len(seqs))]
    idx = api.SalmonIndex.build_mem(names, seqs, threads=4, keep_duplicates=True).to_device(0)
    oidx = orc.OrcIndex(idx)
    recs = []
    def pair(s, p, fl):
        r1 = s[p:p + 100]; r2 = "".join(comp[c] for c in reversed(s[p + fl - 100:p + fl]))
        return [r1, r2] if rng.random() < 0.5 else [r2, r1]
    for _ in range(150): recs += pair(seqs[int(rng.integers(0, 900))], int(rng.integers(225, 245)), int(rng.integers(200, 300)))      # mate 1 spans the variant site
    for _ in range(150): recs += pair(seqs[int(rng.integers(0, 900))], int(rng.integers(170, 200)), int(rng.integers(200, 300)))      # partly inside the element
    for _ in range(200): recs += pair(seqs[int(rng.integers(900, 1050))], int(rng.integers(200, 300)), int(rng.integers(200, 300)))   # family B
    for _ in range(300): recs += pair(seqs[int(rng.integers(1050, 1300))], int(rng.integers(0, 200)), int(rng.integers(200, 300)))    # unique transcripts
    seq = np.frombuffer("".join(recs).encode(), np.uint8).copy(); off = np.arange(0, len(recs) + 1, dtype=np.uint64) * np.uint64(100)
    n = len(recs) // 2
    opts = api.quant_opts()
    ctx = api.QuantContext(idx, opts, device=0, max_batch_reads=1024)
    rb = api.make_read_batch(seq, off, n, paired=True)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
    um_c, mm_c, ch_c, cd_c = orc.map_taps(oidx, opts, rb, cap=1 << 23)
    mm_g = ctx.tap(2, api.MEM_DTYPE)
    per_end = np.bincount(mm_g["end"], minlength=2 * n)
    assert per_end.max() > 1024 and np.any((per_end > 64) & (per_end <= 256)) and np.any((per_end > 256) & (per_end <= 1024)) and np.any((per_end > 0) & (per_end <= 64))
    _fields_equal(mm_g, mm_c, ["end", "tid", "rpos", "qpos", "len", "fw"], "MEMs")
    ch_g = ctx.tap(3, api.CHAIN_DTYPE)
    _fields_equal(ch_g, ch_c, ["end", "tid", "pos", "last_end", "fw", "n_mems", "score"], "chains")
    ro_c, aln_c, mt_c, st_c = orc.map_batch(oidx, opts, rb, threads=8)
    assert st_g == st_c and np.array_equal(ro_g, ro_c) and np.array_equal(mt_g, mt_c)
    _fields_equal(aln_g, aln_c, list(api.ALN_DTYPE.names), "alignments")
    ctx.free()


@pytest.mark.parametrize("overhang", [None, 0])
def test_large_ends_on_a_decoy_chromosome(built, overhang, monkeypatch):
    # [r6] overhang 0: the tiles of k_lg_dp2 / k_lg_accept2 load no records behind their own, so every cluster across a tile's edge continues in global memory
    if overhang is not None: monkeypatch.setenv("SQ_LG_OVERHANG", str(overhang))
    # [r4] The large class (more than 1024 MEMs per end) is chained by flat passes over all its records (mem_kernels.h: k_lg_*): clusters of
    # MEMs no chain can cross, the reference's acceptance order restored by two stable sorts.  A decoy "chromosome" carries 665 copies of a
    # 160-base element (two variants), some of them back to back (several copies in one cluster: the DP and the clash rule see neighbours),
    # so a read across the variant site has ~1600 MEMs on ONE reference; 300 transcripts carry the element too (many small groups in the same end).
    rng = np.random.default_rng(23)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    def rnd(n): return "".join(rng.choice(list("ACGT"), n))
    rep = rnd(160); rep2 = rep[:80] + comp[rep[80]] + rep[81:]
    chrom = []
    for i in range(560):
        chrom.append(rnd(int(rng.integers(260, 500))))
        for _ in range(2 if i % 8 == 0 else 1): chrom.append(rep if rng.random() < 0.5 else rep2)      # i % 8 == 0: two copies back to back
        if i % 16 == 3: chrom.append(rnd(int(rng.integers(5, 30))) + (rep if i % 32 == 3 else rep2))   # ... or a few bases apart
    chrom = "".join(chrom) + rnd(300)
    seqs = []
    for i in range(500):
        body = rnd(int(rng.integers(500, 900)))
        if i < 300: body = body[:250] + (rep if i % 2 else rep2) + body[250:]
        seqs.append(body)
    ncopies = chrom.count(rep) + chrom.count(rep2) + 300
    assert 900 < ncopies <= 1000          # under maxOccsPerHit, or the element's uni-MEMs would be dropped
    seqs.append(chrom)
    names = ["t%d" % i for i in range(len(seqs) - 1)] + ["chrD"]
    idx = api.SalmonIndex.build_mem(names, seqs, threads=4, keep_duplicates=True).to_device(0)
    oidx = orc.OrcIndex(idx)
    recs = []
    def pair(s, p, fl):
        r1 = s[p:p + 100]; r2 = "".join(comp[c] for c in reversed(s[p + fl - 100:p + fl]))
        return [r1, r2] if rng.random() < 0.5 else [r2, r1]
    starts = [m for m in range(len(chrom) - 400) if chrom.startswith(rep, m) or chrom.startswith(rep2, m)]
    for _ in range(200): m = starts[int(rng.integers(len(starts)))]; recs += pair(chrom, max(0, m - int(rng.integers(5, 25))), int(rng.integers(200, 300)))   # across the variant site
    for _ in range(100): m = starts[int(rng.integers(len(starts)))]; recs += pair(chrom, max(0, m - int(rng.integers(60, 90))), int(rng.integers(200, 300)))  # partly inside
    for _ in range(100): recs += pair(seqs[int(rng.integers(0, 300))], int(rng.integers(225, 245)), int(rng.integers(200, 300)))
    for _ in range(100): recs += pair(seqs[int(rng.integers(300, 500))], int(rng.integers(0, 200)), int(rng.integers(200, 300)))
    seq = np.frombuffer("".join(recs).encode(), np.uint8).copy(); off = np.arange(0, len(recs) + 1, dtype=np.uint64) * np.uint64(100)
    n = len(recs) // 2
    opts = api.quant_opts()
    ctx = api.QuantContext(idx, opts, device=0, max_batch_reads=1024)
    rb = api.make_read_batch(seq, off, n, paired=True)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
    um_c, mm_c, ch_c, cd_c = orc.map_taps(oidx, opts, rb, cap=1 << 23)
    mm_g = ctx.tap(2, api.MEM_DTYPE)
    per_end = np.bincount(mm_g["end"], minlength=2 * n)
    big = np.flatnonzero(per_end > 1024)
    assert len(big) > 100
    chr_tid = int(mm_g["tid"].max())
    on_chr = np.bincount(mm_g["end"][mm_g["tid"] == chr_tid], minlength=2 * n)
    assert np.count_nonzero(on_chr > 1024) > 50                           # (end, reference) groups beyond every LDS class
    _fields_equal(mm_g, mm_c, ["end", "tid", "rpos", "qpos", "len", "fw"], "MEMs")
    ch_g = ctx.tap(3, api.CHAIN_DTYPE)
    _fields_equal(ch_g, ch_c, ["end", "tid", "pos", "last_end", "fw", "n_mems", "score"], "chains")
    ro_c, aln_c, mt_c, st_c = orc.map_batch(oidx, opts, rb, threads=8)
    assert st_g == st_c and np.array_equal(ro_g, ro_c) and np.array_equal(mt_g, mt_c)
    _fields_equal(aln_g, aln_c, list(api.ALN_DTYPE.names), "alignments")
    ctx.free()


def _stranded_pairs(seqs, n, rng, flip_from=None):
    # fragments read in ISF orientation (mate 1 = forward strand of the transcript, mate 2 = reverse complement of the fragment's end);
    # from pair index `flip_from` on every third pair is turned into ISR (the mates swapped)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}; recs = []
    for i in range(n):
        s = seqs[int(rng.integers(len(seqs)))]
        while len(s) < 340: s = seqs[int(rng.integers(len(seqs)))]
        fl = int(rng.integers(200, 320)); p = int(rng.integers(0, len(s) - fl))
        r1 = s[p:p + 100]; r2 = "".join(comp[c] for c in reversed(s[p + fl - 100:p + fl]))
        if flip_from is not None and i >= flip_from and i % 3 == 0: r1, r2 = r2, r1
        recs += [r1, r2]
    seq = np.frombuffer("".join(recs).encode(), np.uint8).copy(); off = np.arange(0, len(recs) + 1, dtype=np.uint64) * np.uint64(100)
    return seq, off


def test_library_type_autodetect_matches_checker(small_world):
    # `-l A` (LibraryTypeDetector.hpp; SalmonQuantify.cpp:496-501,692-704; SPEC D8): the library starts as IU; once 50 000 alignments have been
    # seen the most likely format (here ISF) is what the online model expects — the ISR pairs that follow are then incompatible and dropped
    w = small_world; rng = np.random.default_rng(23)
    seqs = [s.decode() for s in w["tx"].seqs()]
    N = 24000
    seq, off = _stranded_pairs(seqs, N, rng, flip_from=16000)
    opts = api.quant_opts(mini_batch_size=1000, num_pre_burnin_frags=400, num_burnin_frags=9000, lib_autodetect=1)
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=10240)
    ost = orc.OrcState(w["oidx"], opts)
    seen = []
    for lo, hi in [(0, 8000), (8000, 18000), (18000, 24000)]:      # ~3.6 alignments per fragment: the 50 000th sample falls inside the second batch
        s = seq[lo * 200: hi * 200]; o = (off[2 * lo: 2 * hi + 1] - off[2 * lo]).copy()
        rb = api.make_read_batch(s, o, hi - lo, paired=True)
        ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb); ctx.eq_accumulate()
        ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=4)
        ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"])
        assert st_g == st_c and np.array_equal(ro_g, ro_c)
        _fields_equal(aln_g, aln_c, list(api.ALN_DTYPE.names), "alignments")
        seen.append(ctx.summary()["lib_detected"])
    ost.finish()
    sg, sc = ctx.summary(), ost.summary()
    assert sg == sc and sg["lib_detected"] == 1 and sg["lib_format_id"] == (1 | (2 << 1) | (0 << 3))      # ISF
    assert seen == [0, 1, 1]
    assert sg["num_assigned"] < sg["num_observed"] - 2000                     # the ISR pairs after detection were dropped as incompatible
    for a, b in zip(ctx.model(), ost.model()[:4]):
        assert np.array_equal(a, b)
    assert np.array_equal(ctx.lib_counts(), ost.lib_counts())
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
        assert np.array_equal(getattr(eq_g, f), getattr(eq_c, f)), f
    ctx.free(); ost.free()


def test_c4_shape_decoy_genome_2x150_matches_checker(built):
    """configs[3] at reduced scale: the transcriptome followed by a synthetic genome as decoys (every gene's exons with introns, 45 % repeat
    families — tools/synth.cpp sqs_genome_generate), 2x150 bp pairs of which 5 % come from gene loci of the genome.  100 000 pairs through
    the HIP path and through the checker: every array equal; genomic pairs end up as decoy fragments, transcript pairs do not."""
    tx = synth.Txome(seed=31, n_genes=800, iso_per_gene=4, threads=4)
    g = synth.Genome(tx, seed=3, total_nt=40_000_000, n_chrom=5, repeat_frac=0.45, threads=4)
    names, seqs, lens = g.append_tables(tx)
    idx = api.SalmonIndex.build_mem_raw(tx.n + g.n, names, seqs, lens, threads=8, first_decoy=tx.n)
    nt = idx.first_decoy                                   # sequence-identical transcripts are dropped (SPEC §I); the decoys follow the rest
    assert 0.8 * tx.n < nt <= tx.n and idx.num_refs == nt + g.n
    oidx = orc.OrcIndex(idx)
    N = 100000
    seq, off, tt, tp = g.reads(tx, N, read_len=150, seed=2, genomic_frac=0.05, threads=4)
    opts = api.quant_opts()
    rb = api.make_read_batch(seq, off, N, paired=True)
    ctx = api.QuantContext(idx, opts, device=0, max_batch_reads=N)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
    ro_c, aln_c, mt_c, st_c = orc.map_batch(oidx, opts, rb, threads=8)
    assert st_g == st_c
    assert np.array_equal(ro_g, ro_c) and np.array_equal(mt_g, mt_c)
    _fields_equal(aln_g, aln_c, list(api.ALN_DTYPE.names), "alignments")
    assert np.all(aln_g["tid"] < nt)
    genomic = tt == 0xFFFFFFFE; txp = tt < 0xFFFFFFF0
    assert (mt_g[genomic] == 6).mean() > 0.95 and (mt_g[txp] == 6).mean() < 0.01      # SQ_MT_DECOY
    assert st_g["num_truncated_ends"] == 0
    ctx.eq_accumulate()
    ost = orc.OrcState(oidx, opts); ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"]); ost.finish()
    assert ctx.summary() == ost.summary()
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
        assert np.array_equal(getattr(eq_g, f), getattr(eq_c, f)), f
    ctx.free(); ost.free()


@pytest.mark.parametrize("variant", ["default", "W1", "W3", "gc", "single_end", "incompat_prior", "no_len_corr", "ISF"])
def test_batches_after_burn_in_take_the_split_path_and_match_checker(small_world, variant):
    """After burn-in the online stage runs as one model-independent launch per mapped batch (k_frag_static) plus, per group of W mini-batches,
    the mass terms (k_frag_dynamic / k_apply_dynamic).  Four batches of 1000 pairs with burn-in inside the first: batches 2-4 take that path.
    Model arrays, counters, library-format counts, the class table (labels, bins, counts, fixed-point weights) equal the checker's."""
    w = small_world; idx = w["idx"]; idx.to_device(0)
    kw = dict(mini_batch_size=100, num_pre_burnin_frags=80, num_burnin_frags=600)
    paired = True
    if variant == "W1": kw["mini_batches_in_flight"] = 1
    if variant == "W3": kw["mini_batches_in_flight"] = 3
    if variant == "gc": kw["gc_bias"] = 1
    if variant == "incompat_prior": kw.update(incompat_prior=-9.2, ignore_incompat=0)
    if variant == "no_len_corr": kw["no_length_correction"] = 1
    opts = api.quant_opts(**kw)
    if variant == "single_end": api.set_libtype(opts, "U"); paired = False
    if variant in ("ISF", "incompat_prior"): api.set_libtype(opts, "ISF")
    if variant == "ISF": opts.num_burnin_frags = 300      # the library is unstranded: about half the fragments are compatible and assigned
    ctx = api.QuantContext(idx, opts, device=0, max_batch_reads=4096)
    ost = orc.OrcState(w["oidx"], opts)
    B = 1000
    for i in range(4):
        lo, hi = i * B, (i + 1) * B
        if paired:
            s = w["seq"][lo * 200: hi * 200]; o = (w["off"][2 * lo: 2 * hi + 1] - w["off"][2 * lo]).copy()
        else:   # mate 1 of every pair as a single-end library
            s = np.concatenate([w["seq"][(2 * j) * 100:(2 * j + 1) * 100] for j in range(lo, hi)]); o = np.arange(0, B + 1, dtype=np.uint64) * np.uint64(100)
        rb = api.make_read_batch(s, o, B, paired=paired)
        ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb); ctx.eq_accumulate()
        ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=4); ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"])
        assert st_g == st_c and aln_g.tobytes() == aln_c.tobytes()
        if i == 0: assert ctx.summary()["burned_in"]
    ost.finish()
    assert ctx.summary() == ost.summary()
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]:
        assert np.array_equal(getattr(eq_g, f), getattr(eq_c, f)), f
    mg, mc = ctx.model(), ost.model()
    for a, b, what in zip(mg, mc[:4], ["log mass", "unique counts", "total counts", "log effective length"]):
        assert np.array_equal(a, b), what
    assert np.array_equal(ctx.fld(), mc[4]) and np.array_equal(ctx.lib_counts(), ost.lib_counts())
    if variant == "gc": assert np.array_equal(ctx.gc_observed(), ost.gc_observed())
    ctx.free(); ost.free()


def _pack_rule(seq_bytes, off, nrec, rw):
    # the packing rule of k_pack (map_kernels.h): A, C, G, T in either case -> 0, 1, 2, 3, two bits a base, 32 bases a word; any other byte packs as 0 and sets its bit in the mask
    code = np.full(256, 255, np.uint8)
    for ch, v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]): code[ch] = v
    rec = 1 + rw + rw // 2; out = np.zeros((nrec, rec), np.uint64)
    for e in range(nrec):
        s = seq_bytes[int(off[e]):int(off[e + 1])]; c = code[s]; L = len(s)
        bad = c == 255
        out[e, 0] = np.uint64(L) | (np.uint64(1 if bad.any() else 0) << np.uint64(32))
        c2 = np.where(bad, 0, c).astype(np.uint64)
        for i in range(L):
            out[e, 1 + i // 32] |= c2[i] << np.uint64(2 * (i % 32))
            if bad[i]: out[e, 1 + rw + i // 64] |= np.uint64(1) << np.uint64(i % 64)
    return out.reshape(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["2x100", "2x151_odd_offsets", "ragged_20_256", "ragged_140_170_across_the_lds_cap", "device_buffer_at_an_odd_address"])
def test_packed_reads_follow_the_packing_rule(small_world, shape):
    # [r6] k_pack8_staged (a wave's text through LDS) and the thread-per-end code it falls back to, held to the rule itself through the SQ_TAP_PACKED tap: read starts at every
    # byte alignment, lengths from 20 to 256 (waves whose text exceeds the LDS cap take the fallback; so does any wave with a longer end), non-bases and lower case anywhere
    # (first / last base of a read, of a word, of the batch), a last wave of fewer than 64 ends, and a device-resident batch whose buffer starts at an odd address
    w = small_world; rng = np.random.default_rng(77)
    n = 1000 + 37                                   # pairs: 2 074 ends = 32 waves and a partial one
    if shape == "2x100": lens = np.full(2 * n, 100)
    elif shape == "2x151_odd_offsets": lens = np.full(2 * n, 151)
    elif shape == "ragged_20_256": lens = rng.integers(20, 257, 2 * n)
    elif shape == "ragged_140_170_across_the_lds_cap": lens = rng.integers(140, 171, 2 * n)
    else: lens = rng.integers(31, 140, 2 * n)
    off = np.zeros(2 * n + 1, np.uint64); off[1:] = np.cumsum(lens)
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(off[-1])).copy()
    lower = rng.random(len(seq)) < 0.1; seq[lower] |= 0x20
    junk = rng.random(len(seq)) < 0.01; seq[junk] = rng.choice(np.frombuffer(b"NnRY.-*@\x00\xff", np.uint8), int(junk.sum()))
    for e in range(0, 2 * n, 7):                    # non-bases at the edges of reads and words
        a, b = int(off[e]), int(off[e + 1])
        seq[a if e % 3 == 0 else b - 1] = ord("N")
        if b - a > 40: seq[a + (31 if e % 2 else 32)] = ord("n")
    seq[0] = ord("N"); seq[-1] = ord("N")
    opts = api.quant_opts()
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=2048)
    if shape == "device_buffer_at_an_odd_address":
        import torch
        pad = 5; dbuf = torch.zeros(len(seq) + pad + 64, dtype=torch.uint8, device="cuda:0"); dbuf[pad:pad + len(seq)] = torch.from_numpy(seq).to("cuda:0")
        doff = torch.from_numpy(off.astype(np.int64)).to("cuda:0"); torch.cuda.synchronize()
        rb = api.make_read_batch(int(dbuf.data_ptr()) + pad, int(doff.data_ptr()), n, paired=True, on_device=True)
    else:
        rb = api.make_read_batch(seq, off, n, paired=True)
    ctx.map_batch(rb)
    got = ctx.tap(5, np.uint64)
    want = _pack_rule(seq, off, 2 * n, 8)
    assert len(got) == len(want)
    bad = np.flatnonzero(got != want)
    assert len(bad) == 0, "end %d word %d: %x != %x" % (bad[0] // 13, bad[0] % 13, int(got[bad[0]]), int(want[bad[0]]))
    ctx.free()
