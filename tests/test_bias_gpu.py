"""Fragment-GC bias (row f-3, --gcBias): the observed model collected while mapping, the expected model + bias-corrected effective
lengths (updateEffectiveLengths, SalmonUtils.cpp:1208-1985, gc branches) and the EM with the bias hook at iteration 11
(CollapsedEMOptimizer.cpp:901-928) — HIP path vs the CPU checker, bit for bit — plus what the correction must do on data with a
planted GC bias."""
import numpy as np
import pytest
from salmon_amd import api, synth
import orc

pytestmark = pytest.mark.gpu


def _run(w, seq, off, n):
    opts = api.quant_opts(gc_bias=1, mini_batch_size=1000, num_pre_burnin_frags=400, num_burnin_frags=3000)
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=max(4096, n))
    rb = api.make_read_batch(seq, off, n, paired=True)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb); ctx.eq_accumulate()
    ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=8)
    ost = orc.OrcState(w["oidx"], opts); ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"]); ost.finish()
    return ctx, ost


def test_gc_models_effective_lengths_and_em_match_checker(small_world):
    w = small_world; w["idx"].to_device(0)
    ctx, ost = _run(w, w["seq"], w["off"], w["n"])
    assert ctx.summary() == ost.summary()
    g_g, g_c = ctx.gc_observed(), ost.gc_observed()
    assert np.array_equal(g_g, g_c) and g_g.sum() > 0.5 * w["n"] and g_g.shape == (3, 25)       # every properly paired, assigned fragment is an observation
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    lm, uq, tc, le = ctx.model(); fld = ctx.fld(); mc = ost.model()
    assert np.array_equal(fld, mc[4])
    proj = api.normalize_alphas(eq_g, lm, uq, tc); eff = np.exp(le)
    # the effective-length update on its own, from the same inputs
    a0 = np.maximum(proj, 0.0)
    e_g, r_g = api.bias_gc_eff_lengths(w["idx"], g_g, fld, a0, eff)
    e_c, r_c = orc.bias_gc_eff_lengths(w["oidx"], g_c, mc[4], a0, eff)
    assert r_g["num_processed"] == r_c["num_processed"] > 50
    assert np.array_equal(r_g["gc_bias"], r_c["gc_bias"]) and np.array_equal(e_g, e_c)
    assert 150 < r_g["fld_low"] < 250 < r_g["fld_high"] < 400
    # the whole optimisation with the hook
    al_g, ef_g, rep_g = ctx.em_optimize_gc(eff, proj, g_g, fld, api.em_opts())
    al_c, ef_c, rep_c = orc.em_optimize_gc(eq_c, eff, proj, w["oidx"], g_c, mc[4], api.em_opts())
    assert rep_g["iters"] == rep_c["iters"] and np.array_equal(ef_g, ef_c) and np.array_equal(al_g, al_c)
    assert abs(al_g.sum() - ctx.summary()["num_assigned"]) < 1e-6 * al_g.sum()
    plain, _ = ctx.em_optimize(eff, proj, api.em_opts())
    assert not np.array_equal(plain, al_g)
    ctx.free(); ost.free()


def test_gc_correction_on_a_gc_filtered_library(small_world):
    # a library that lost most of its GC-rich fragments: the HIP path must still equal the checker, the bias factors stay inside the
    # reference's clamp [1/1000, 1000], and processed transcripts get new, positive effective lengths no shorter than the barrier allows.
    # (In gc-only mode the reference compares the observed fragments of the LOW-context bin with all expected fragments —
    # SalmonUtils.cpp:1562-1565 leaves the context counts empty — so the factors are not a clean function of GC; parity is the test.)
    w = small_world; w["idx"].to_device(0)
    seq, off, tt, tp = w["tx"].reads(20000, read_len=100, seed=31, threads=4)
    recs = seq.reshape(-1, 100); gc = ((recs == ord("G")) | (recs == ord("C"))).mean(axis=1); pair_gc = 0.5 * (gc[0::2] + gc[1::2])
    rng = np.random.default_rng(5); keep = rng.random(len(pair_gc)) < np.clip(1.6 - 2.4 * pair_gc, 0.05, 1.0)
    idxs = np.nonzero(keep)[0]; n = len(idxs)
    sel = np.stack([recs[2 * idxs], recs[2 * idxs + 1]], axis=1).reshape(-1)
    off2 = np.arange(0, 2 * n + 1, dtype=np.uint64) * np.uint64(100)
    ctx, ost = _run(w, np.ascontiguousarray(sel), off2, n)
    assert np.array_equal(ctx.gc_observed(), ost.gc_observed())
    eq = ctx.eq_finish(); lm, uq, tc, le = ctx.model(); fld = ctx.fld()
    proj = api.normalize_alphas(eq, lm, uq, tc); eff = np.exp(le)
    e_new, rep = api.bias_gc_eff_lengths(w["idx"], ctx.gc_observed(), fld, np.maximum(proj, 0.0), eff)
    e_chk, rep_c = orc.bias_gc_eff_lengths(w["oidx"], ost.gc_observed(), ost.model()[4], np.maximum(proj, 0.0), eff)
    assert np.array_equal(e_new, e_chk) and np.array_equal(rep["gc_bias"], rep_c["gc_bias"])
    b = rep["gc_bias"]
    assert np.all(b >= 1e-3) and np.all(b <= 1e3) and len(np.unique(np.round(b, 9))) >= 4 and rep["num_processed"] > 50
    assert np.all(e_new > 0) and np.any(np.abs(e_new - eff) > 1.0)
    ctx.free(); ost.free()


def test_cli_gcbias_writes_corrected_effective_lengths(built, tmp_path):
    import os, subprocess, json, fixtures
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "salmon_amd", "bin", "salmon-hip")
    g = fixtures.G
    subprocess.check_call([exe, "index", "-t", os.path.join(g, "transcripts.fa.gz"), "-i", str(tmp_path / "idx"), "-p", "2"])
    for out, extra in (("plain", []), ("gc", ["--gcBias"])):
        subprocess.check_call([exe, "quant", "-i", str(tmp_path / "idx"), "-l", "IU", "-1", os.path.join(g, "reads_1.fq.gz"), "-2", os.path.join(g, "reads_2.fq.gz"),
                               "-o", str(tmp_path / out)] + extra)
    rows = {o: [l.split("\t") for l in open(tmp_path / o / "quant.sf").read().splitlines()[1:]] for o in ("plain", "gc")}
    assert open(tmp_path / "plain" / "quant.sf").read() == open(os.path.join(g, "golden_quant.sf")).read()
    e0 = np.array([float(r[2]) for r in rows["plain"]]); e1 = np.array([float(r[2]) for r in rows["gc"]])
    n0 = np.array([float(r[4]) for r in rows["plain"]]); n1 = np.array([float(r[4]) for r in rows["gc"]])
    assert np.any(np.abs(e1 - e0) > 1.0) and abs(n1.sum() - n0.sum()) < 1e-3 * n0.sum() and np.corrcoef(n0, n1)[0, 1] > 0.99
    assert json.load(open(tmp_path / "gc" / "aux_info" / "meta_info.json"))["gc_bias_correct"] is True
    r = subprocess.run([exe, "quant", "-i", str(tmp_path / "idx"), "-l", "U", "-r", os.path.join(g, "reads_1.fq.gz"), "-o", str(tmp_path / "se"), "--gcBias"], capture_output=True, text=True)
    # [r3] single-end libraries are corrected too (fragments taken at the conditional mean length of the prior), and --seqBias with --gcBias
    assert r.returncode == 0, r.stderr[-800:]
    assert json.load(open(tmp_path / "se" / "aux_info" / "meta_info.json"))["gc_bias_correct"] is True
    r = subprocess.run([exe, "quant", "-i", str(tmp_path / "idx"), "-l", "IU", "-1", os.path.join(g, "reads_1.fq.gz"), "-2", os.path.join(g, "reads_2.fq.gz"),
                        "-o", str(tmp_path / "sg"), "--gcBias", "--seqBias"], capture_output=True, text=True)
    assert r.returncode == 0 and "fragments sampled for the read-start context models" in r.stderr, r.stderr[-800:]
    m = json.load(open(tmp_path / "sg" / "aux_info" / "meta_info.json")); assert m["gc_bias_correct"] is True and m["seq_bias_correct"] is True
    n2 = np.array([float(l.split("\t")[4]) for l in open(tmp_path / "sg" / "quant.sf").read().strip().split("\n")[1:]])
    assert abs(n2.sum() - n0.sum()) < 1e-3 * n0.sum() and np.corrcoef(n0, n2)[0, 1] > 0.98
    r = subprocess.run([exe, "quant", "-i", str(tmp_path / "idx"), "-l", "IU", "-1", os.path.join(g, "reads_1.fq.gz"), "-2", os.path.join(g, "reads_2.fq.gz"),
                        "-o", str(tmp_path / "all3"), "--gcBias", "--seqBias", "--posBias"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    m = json.load(open(tmp_path / "all3" / "aux_info" / "meta_info.json")); assert m["salmon_hip"]["pos_bias_correct"] is True and m["seq_bias_correct"] is True
    n3 = np.array([float(l.split("\t")[4]) for l in open(tmp_path / "all3" / "quant.sf").read().strip().split("\n")[1:]])
    assert abs(n3.sum() - n0.sum()) < 1e-3 * n0.sum() and np.corrcoef(n0, n3)[0, 1] > 0.97 and not np.array_equal(n3, n2)


@pytest.mark.parametrize("with_gc", [False, True])
def test_seq_bias_models_effective_lengths_and_em_match_checker(small_world, with_gc):
    """--seqBias (alone and with --gcBias): the observed read-start context models collected by the online stage (one sampled alignment per
    fragment, first num_bias_samples in read order), the expected context models, the expected GC model with its context bins, the
    corrected effective lengths and the EM with the bias hook — HIP path vs the checker, bit for bit (SPEC §B2)."""
    w = small_world; w["idx"].to_device(0)
    opts = api.quant_opts(seq_bias=1, gc_bias=1 if with_gc else 0, num_bias_samples=2500, mini_batch_size=1000, num_pre_burnin_frags=400, num_burnin_frags=3000)
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
    ost = orc.OrcState(w["oidx"], opts)
    for lo in (0, 2000):   # two batches: the cap of 2500 sampled fragments falls inside the second
        hi = lo + 2000
        s = w["seq"][lo * 200: hi * 200]; o = (w["off"][2 * lo: 2 * hi + 1] - w["off"][2 * lo]).copy()
        rb = api.make_read_batch(s, o, hi - lo, paired=True)
        ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb); ctx.eq_accumulate()
        ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=8); ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"])
    ost.finish()
    fw_g, rc_g, n_g = ctx.seq_observed(); fw_c, rc_c, n_c = ost.seq_observed()
    assert n_g == n_c == 2500 and np.array_equal(fw_g, fw_c) and np.array_equal(rc_g, rc_c)
    assert int(fw_g[:4].sum()) == 2500 and int(rc_g[:4].sum()) == 2500          # every sampled fragment adds one context to each model (column 0 has 4 cells)
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    lm, uq, tc, le = ctx.model(); fld = ctx.fld(); mc = ost.model()
    assert np.array_equal(fld, mc[4])
    gcg = ctx.gc_observed() if with_gc else None; gcc = ost.gc_observed() if with_gc else None
    proj = api.normalize_alphas(eq_g, lm, uq, tc); eff = np.exp(le); a0 = np.maximum(proj, 0.0)
    e_g, m_g, r_g = api.bias_seq_eff_lengths(w["idx"], fw_g, rc_g, fld, a0, eff, gc_obs=gcg)
    e_c, m_c, r_c = orc.bias_seq_eff_lengths(w["oidx"], fw_c, rc_c, mc[4], a0, eff, gc_obs=gcc)
    assert r_g["num_processed"] == r_c["num_processed"] > 50
    for k, name in enumerate(["expected fw", "expected rc", "observed fw", "observed rc"]):
        assert np.array_equal(m_g[k], m_c[k]), name
    assert np.array_equal(e_g, e_c) and np.all(e_g > 0) and np.any(np.abs(e_g - eff) > 0.5)
    # conditional probabilities: every context's four cells sum to one
    p = np.exp(m_g[2].reshape(9, 64)); assert abs(p[0, :4].sum() - 1) < 1e-9 and abs(p[5, 8:12].sum() - 1) < 1e-9
    al_g, ef_g, rep_g = ctx.em_optimize_seq(eff, proj, fw_g, rc_g, fld, gc_obs=gcg)
    al_c, ef_c, rep_c = orc.em_optimize_bias(eq_c, eff, proj, w["oidx"], fw_c, rc_c, mc[4], gc_obs=gcc)
    assert rep_g["iters"] == rep_c["iters"] and np.array_equal(ef_g, ef_c) and np.array_equal(al_g, al_c)
    ctx.free(); ost.free()


def test_seq_bias_single_end_library_matches_checker(small_world):
    # single-end libraries sample one context per read (SalmonQuantify.cpp:2211-2257: the start of a reverse read is pos + readLen there)
    w = small_world; w["idx"].to_device(0)
    opts = api.quant_opts(seq_bias=1, mini_batch_size=1000, num_pre_burnin_frags=400, num_burnin_frags=3000); api.set_libtype(opts, "U")
    n = 3000
    s = np.concatenate([w["seq"][(2 * j) * 100:(2 * j + 1) * 100] for j in range(n)]); o = np.arange(0, n + 1, dtype=np.uint64) * np.uint64(100)
    rb = api.make_read_batch(s, o, n, paired=False)
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb); ctx.eq_accumulate()
    ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=8)
    ost = orc.OrcState(w["oidx"], opts); ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"]); ost.finish()
    fw_g, rc_g, n_g = ctx.seq_observed(); fw_c, rc_c, n_c = ost.seq_observed()
    assert n_g == n_c > 1000 and np.array_equal(fw_g, fw_c) and np.array_equal(rc_g, rc_c)
    assert int(fw_g[:4].sum() + rc_g[:4].sum()) == n_g                       # one context per sampled read, in one of the two models
    eq_g = ctx.eq_finish(); lm, uq, tc, le = ctx.model(); fld = ctx.fld()
    proj = api.normalize_alphas(eq_g, lm, uq, tc); eff = np.exp(le); a0 = np.maximum(proj, 0.0)
    e_g, m_g, r_g = api.bias_seq_eff_lengths(w["idx"], fw_g, rc_g, fld, a0, eff)
    e_c, m_c, r_c = orc.bias_seq_eff_lengths(w["oidx"], fw_c, rc_c, ost.model()[4], a0, eff)
    assert np.array_equal(m_g, m_c) and np.array_equal(e_g, e_c)
    ctx.free(); ost.free()


def test_gc_bias_single_end_library_matches_checker(small_world):
    # a single-end library observes every fragment at the conditional mean length of the prior distribution (SalmonQuantify.cpp:952-971)
    w = small_world; w["idx"].to_device(0)
    opts = api.quant_opts(gc_bias=1, mini_batch_size=1000, num_pre_burnin_frags=400, num_burnin_frags=3000); api.set_libtype(opts, "U")
    n = 3000
    s = np.concatenate([w["seq"][(2 * j) * 100:(2 * j + 1) * 100] for j in range(n)]); o = np.arange(0, n + 1, dtype=np.uint64) * np.uint64(100)
    rb = api.make_read_batch(s, o, n, paired=False)
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb); ctx.eq_accumulate()
    ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=8)
    ost = orc.OrcState(w["oidx"], opts); ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"]); ost.finish()
    g_g, g_c = ctx.gc_observed(), ost.gc_observed()
    assert np.array_equal(g_g, g_c) and g_g.sum() > 0.3 * n
    eq_g = ctx.eq_finish(); lm, uq, tc, le = ctx.model(); fld = ctx.fld()
    proj = api.normalize_alphas(eq_g, lm, uq, tc); eff = np.exp(le); a0 = np.maximum(proj, 0.0)
    e_g, r_g = api.bias_gc_eff_lengths(w["idx"], g_g, fld, a0, eff); e_c, r_c = orc.bias_gc_eff_lengths(w["oidx"], g_c, ost.model()[4], a0, eff)
    assert np.array_equal(e_g, e_c) and np.array_equal(r_g["gc_bias"], r_c["gc_bias"])
    ctx.free(); ost.free()


@pytest.mark.parametrize("combo", ["pos", "pos+gc", "pos+seq", "pos+seq+gc"])
def test_pos_bias_models_effective_lengths_and_em_match_checker(small_world, combo):
    """--posBias, alone and with the other corrections (SPEC §P): the observed read-start models by transcript length class collected by the
    online stage (general kernels before burn-in, the split path after), the expected models, the splines, the corrected effective lengths
    and the EM with the bias hook — HIP path vs the checker, bit for bit."""
    w = small_world; w["idx"].to_device(0)
    gc, sq = "gc" in combo, "seq" in combo
    opts = api.quant_opts(pos_bias=1, gc_bias=1 if gc else 0, seq_bias=1 if sq else 0, num_bias_samples=2500, mini_batch_size=500, num_pre_burnin_frags=400, num_burnin_frags=1200,
                          mini_batches_in_flight=3)
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
    ost = orc.OrcState(w["oidx"], opts)
    for lo in (0, 2000):   # burn-in ends inside the first batch: the second one takes the static / dynamic kernels
        hi = lo + 2000
        s = w["seq"][lo * 200: hi * 200]; o = (w["off"][2 * lo: 2 * hi + 1] - w["off"][2 * lo]).copy()
        rb = api.make_read_batch(s, o, hi - lo, paired=True)
        ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb); ctx.eq_accumulate()
        ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=8); ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"])
    ost.finish()
    assert ctx.summary() == ost.summary()
    q_g, c_g = api.length_classes(w["idx"]); q_c, c_c = orc.length_classes(w["oidx"])
    assert np.array_equal(q_g, q_c) and np.array_equal(c_g, c_c) and len(q_g) == 5 and set(np.unique(c_g)) == {0, 1, 2, 3, 4}
    p_g, p_c = ctx.pos_observed(), ost.pos_observed()
    assert np.array_equal(p_g, p_c) and p_g.shape == (2, 5, 20)
    na = ctx.summary()["num_assigned"]
    assert 0.97 * na < p_g[0].sum() <= na + 1e-6 and 0.97 * na < p_g[1].sum() <= na + 1e-6 and p_g.sum() >= na - 1e-6   # a proper pair adds mass 1 to each model, an orphan to one of them
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    lm, uq, tc, le = ctx.model(); fld = ctx.fld(); mc = ost.model()
    assert np.array_equal(fld, mc[4])
    gcg = ctx.gc_observed() if gc else None; gcc = ost.gc_observed() if gc else None
    sg = ctx.seq_observed()[:2] if sq else None; sc = ost.seq_observed()[:2] if sq else None
    proj = api.normalize_alphas(eq_g, lm, uq, tc); eff = np.exp(le); a0 = np.maximum(proj, 0.0)
    e_g, sm_g, pm_g, r_g = api.bias_eff_lengths(w["idx"], fld, a0, eff, gc_obs=gcg, seq=sg, pos_obs=p_g, threads=3)
    e_c, pm_c, r_c = orc.bias_eff_lengths(w["oidx"], mc[4], a0, eff, gc_obs=gcc, seq=sc, pos_obs=p_c, threads=3)
    assert r_g["num_processed"] == r_c["num_processed"] > 50
    assert np.array_equal(pm_g.reshape(4, 100), pm_c), "normalised positional models (observed 5', 3', expected 5', 3')"
    assert np.allclose(pm_g.reshape(4, 5, 20).sum(axis=2), 1.0)
    assert np.array_equal(e_g, e_c) and np.all(e_g > 0) and np.any(np.abs(e_g - eff) > 0.5)
    e1, _, _, _ = api.bias_eff_lengths(w["idx"], fld, a0, eff, gc_obs=gcg, seq=sg, pos_obs=p_g, threads=8)
    assert not np.array_equal(e1, e_g)                                                   # the initial mass per bin is 1 + threads
    al_g, ef_g, rep_g = ctx.em_optimize_bias(eff, proj, fld, gc_obs=gcg, seq=sg, pos_obs=p_g, threads=3)
    al_c, ef_c, rep_c = orc.em_optimize_bias_pos(eq_c, eff, proj, w["oidx"], mc[4], gc_obs=gcc, seq=sc, pos_obs=p_c, threads=3)
    assert rep_g["iters"] == rep_c["iters"] and np.array_equal(ef_g, ef_c) and np.array_equal(al_g, al_c)
    ctx.free(); ost.free()


def test_pos_bias_single_end_library_and_cli(small_world, built, tmp_path):
    # single-end reads: the forward ones feed the 5' model, the others the 3' model (SalmonQuantify.cpp:917-933)
    w = small_world; w["idx"].to_device(0)
    opts = api.quant_opts(pos_bias=1, mini_batch_size=1000, num_pre_burnin_frags=400, num_burnin_frags=3000); api.set_libtype(opts, "U")
    n = 3000
    s = np.concatenate([w["seq"][(2 * j) * 100:(2 * j + 1) * 100] for j in range(n)]); o = np.arange(0, n + 1, dtype=np.uint64) * np.uint64(100)
    rb = api.make_read_batch(s, o, n, paired=False)
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=4096)
    ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb); ctx.eq_accumulate()
    ro_c, aln_c, mt_c, st_c = orc.map_batch(w["oidx"], opts, rb, threads=8)
    ost = orc.OrcState(w["oidx"], opts); ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"]); ost.finish()
    p_g, p_c = ctx.pos_observed(), ost.pos_observed()
    na = ctx.summary()["num_assigned"]
    assert np.array_equal(p_g, p_c) and abs(p_g.sum() - na) < 1e-3 * na and p_g[0].sum() > 0.2 * na and p_g[1].sum() > 0.2 * na
    eq_g = ctx.eq_finish(); lm, uq, tc, le = ctx.model(); fld = ctx.fld()
    proj = api.normalize_alphas(eq_g, lm, uq, tc); eff = np.exp(le); a0 = np.maximum(proj, 0.0)
    e_g, _, pm_g, _ = api.bias_eff_lengths(w["idx"], fld, a0, eff, pos_obs=p_g)
    e_c, pm_c, _ = orc.bias_eff_lengths(w["oidx"], ost.model()[4], a0, eff, pos_obs=p_c)
    assert np.array_equal(pm_g.reshape(4, 100), pm_c) and np.array_equal(e_g, e_c)
    ctx.free(); ost.free()


def test_bias_hook_fires_at_the_first_convergence_when_that_comes_before_iteration_11(small_world):
    # CollapsedEMOptimizer.cpp:901 `itNum > targetIt or converged`: with a tolerance everything meets the hook comes after the first update
    w = small_world; w["idx"].to_device(0)
    ctx, ost = _run(w, w["seq"], w["off"], w["n"])
    g_g, g_c = ctx.gc_observed(), ost.gc_observed()
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    lm, uq, tc, le = ctx.model(); fld = ctx.fld(); mc = ost.model()
    proj = api.normalize_alphas(eq_g, lm, uq, tc); eff = np.exp(le)
    loose = api.em_opts(); loose.rel_diff_tolerance = 1e9; loose.min_iter = 30
    al_g, ef_g, rep_g = ctx.em_optimize_gc(eff, proj, g_g, fld, loose)
    al_c, ef_c, rep_c = orc.em_optimize_gc(eq_c, eff, proj, w["oidx"], g_c, mc[4], loose)
    assert rep_g["iters"] == rep_c["iters"] == 30 and np.array_equal(ef_g, ef_c) and np.array_equal(al_g, al_c)
    al_d, ef_d, rep_d = ctx.em_optimize_gc(eff, proj, g_g, fld, api.em_opts())
    assert not np.array_equal(ef_d, ef_g)            # the default run calls the hook with the alphas of iteration 11, this one with those of iteration 1
    ctx.free(); ost.free()
