"""C1 = BASELINE.json configs[0]: the reference's bundled sample data (sample_data.tgz: 15 transcripts, 10 000 simulated
read pairs whose names carry the true transcript), committed as tests/golden/c1/ by tests/golden/make_c1.py because the
GPU box has no /root/reference.  The CPU checker must keep reproducing the committed digests; the HIP path — through the
C ABI and through the stand-alone `salmon-hip` binary — must equal them and recover the true transcripts."""
import hashlib, os, subprocess
import numpy as np
import pytest
from salmon_amd import api
import fixtures


def test_checker_reproduces_c1_digests(built):
    m = fixtures.c1_meta()
    r = fixtures.c1_run_checker(threads=4)
    assert r["n"] == m["n_pairs"] == 10000 and r["idx"].num_refs == m["num_refs"]
    assert r["stats"] == m["stats"] and r["summary"] == m["summary"]
    assert hashlib.sha256(r["aln"].tobytes()).hexdigest() == m["alignments_sha256"]
    assert hashlib.sha256(r["read_off"].tobytes()).hexdigest() == m["read_off_sha256"]
    assert fixtures.eq_digest(r["eq"]) == m["eq_sha256"] and len(r["eq"].count) == m["num_eq_classes"]
    assert r["rep"]["iters"] == m["em_iters"] and [float(a).hex() for a in r["alphas"]] == m["alphas_hex"]
    # the data's own ground truth (the bar tests/test_oracle_e2e.py holds the checker to on the un-committed original)
    assert r["stats"]["num_mapped"] >= 0.9 * r["n"] and r["recall"] >= 0.98 and r["corr"] > 0.98


@pytest.mark.gpu
def test_gpu_api_on_c1_equals_checker_and_recovers_truth(built):
    m = fixtures.c1_meta(); d = fixtures.c1_load()
    idx = api.SalmonIndex.build_mem(d["names"], d["seqs"], threads=2).to_device(0)
    ctx = api.QuantContext(idx, api.quant_opts(), device=0, max_batch_reads=16384)
    ro, aln, mt, st = ctx.map_batch(api.make_read_batch(d["seq"], d["off"], d["n"], paired=True))
    assert st == m["stats"]
    assert hashlib.sha256(aln.tobytes()).hexdigest() == m["alignments_sha256"]
    assert hashlib.sha256(ro.tobytes()).hexdigest() == m["read_off_sha256"]
    ctx.eq_accumulate()
    eq = ctx.eq_finish()
    assert fixtures.eq_digest(eq) == m["eq_sha256"] and ctx.summary() == m["summary"]
    lm, uq, tc, le = ctx.model()
    proj = api.normalize_alphas(eq, lm, uq, tc)
    alphas, rep = ctx.em_optimize(np.exp(le), proj, api.em_opts())
    assert rep["iters"] == m["em_iters"] and [float(a).hex() for a in alphas] == m["alphas_hex"]
    rec, r = fixtures.truth_scores(idx, d["truth"], ro, aln, alphas)
    assert st["num_mapped"] >= 0.9 * d["n"] and rec >= 0.98 and r > 0.98
    ctx.free()


@pytest.mark.gpu
def test_gpu_cli_on_c1_writes_the_golden_quant_sf(built, tmp_path):
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "salmon_amd", "bin", "salmon-hip")
    g = fixtures.C1
    subprocess.check_call([exe, "index", "-t", os.path.join(g, "transcripts.fa.gz"), "-i", str(tmp_path / "idx"), "-p", "2"])
    subprocess.check_call([exe, "quant", "-i", str(tmp_path / "idx"), "-l", "IU", "-1", os.path.join(g, "reads_1.fq.gz"),
                           "-2", os.path.join(g, "reads_2.fq.gz"), "-o", str(tmp_path / "out")])
    assert open(tmp_path / "out" / "quant.sf").read() == open(os.path.join(g, "golden_quant.sf")).read()
    # NumReads against the truth in the read names
    d = fixtures.c1_load()
    rows = [l.split("\t") for l in open(tmp_path / "out" / "quant.sf").read().splitlines()[1:]]
    got = {r[0]: float(r[4]) for r in rows}
    true = {}
    for t in d["truth"]: true[t] = true.get(t, 0) + 1
    names = [r[0] for r in rows]
    assert np.corrcoef([got[n] for n in names], [true.get(n, 0) for n in names])[0, 1] > 0.98
