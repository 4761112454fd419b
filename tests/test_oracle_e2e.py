"""The checker on the reference's bundled sample data (configs[0]): read names carry the
truth (`@<n>:<txp>:<pos>:<fraglen>`), so mapping and quantification can be checked against it.
Skipped where /root/reference is absent (the GPU box)."""
import os, tarfile
import numpy as np
import pytest
from salmon_amd import api
import orc

SRC = "/root/reference/sample_data.tgz"


@pytest.fixture(scope="module")
def sample(built, tmp_path_factory):
    if not os.path.exists(SRC):
        pytest.skip("reference sample data not present")
    d = tmp_path_factory.mktemp("sd")
    tarfile.open(SRC).extractall(d)
    sd = os.path.join(d, "sample_data")
    api.SalmonIndex.build(os.path.join(sd, "transcripts.fasta"), os.path.join(d, "idx"), threads=2)
    idx = api.SalmonIndex.load(os.path.join(d, "idx"))
    recs, truth = [], []
    with open(os.path.join(sd, "reads_1.fastq")) as f1, open(os.path.join(sd, "reads_2.fastq")) as f2:
        l1, l2 = f1.read().split("\n"), f2.read().split("\n")
    for i in range(0, len(l1) - 3, 4):
        recs.append(l1[i + 1].encode()); recs.append(l2[i + 1].encode()); truth.append(l1[i][1:].split(":")[1].split("/")[0])
    seq = np.frombuffer(b"".join(recs), np.uint8).copy()
    off = np.zeros(len(recs) + 1, np.uint64); off[1:] = np.cumsum([len(r) for r in recs])
    return dict(idx=idx, seq=seq, off=off, n=len(truth), truth=truth)


def test_sample_data_maps_to_the_true_transcripts(sample):
    idx = sample["idx"]; oidx = orc.OrcIndex(idx)
    opts = api.quant_opts()
    rb = api.make_read_batch(sample["seq"], sample["off"], sample["n"], paired=True)
    ro, aln, mt, st = orc.map_batch(oidx, opts, rb, threads=4)
    names = idx.ref_names(); name2tid = {n: i for i, n in enumerate(names)}
    assert sample["n"] == 10000
    hit = sum(1 for i in range(sample["n"]) if name2tid.get(sample["truth"][i], -1) in aln["tid"][int(ro[i]):int(ro[i + 1])])
    assert st["num_mapped"] >= 0.9 * sample["n"]
    assert hit >= 0.98 * st["num_mapped"]
    # quantify and compare NumReads with the true per-transcript fragment counts
    ost = orc.OrcState(oidx, opts); ost.eq_accumulate(ro, aln, st["num_with_joint_hits"]); ost.finish()
    eq = ost.eq_finish(); lm, uq, tc, le, fld = ost.model()
    proj = orc.normalize_alphas(idx.num_refs, eq, lm, uq, tc)
    alphas, rep = orc.em_optimize(eq, np.exp(le), proj, api.em_opts())
    true = np.zeros(idx.num_refs)
    for t in sample["truth"]:
        if t in name2tid: true[name2tid[t]] += 1
    r = np.corrcoef(alphas, true)[0, 1]
    assert r > 0.98
    assert abs(alphas.sum() - ost.summary()["num_assigned"]) < 1e-6 * alphas.sum()


def test_eq_class_file_round_trip_and_bootstrap_writer(built, tmp_path):
    # `salmon quant -e` interchange (readEquivCounts) against our own --dumpEqWeights writer, and the
    # bootstraps.gz / names.tsv.gz layout (raw f64[M] per replicate)
    import gzip
    from conftest import random_eq_classes
    from salmon_amd import synth
    tx = synth.Txome(seed=21, n_genes=10, iso_per_gene=3, threads=1)
    names, seqs, lens = tx.tables()
    idx = api.SalmonIndex.build_mem_raw(tx.n, names, seqs, lens, threads=1)
    M = idx.num_refs
    eq = random_eq_classes(M, 200, seed=3)
    p = str(tmp_path / "eq_classes.txt.gz")
    api.write_eq_classes(p, idx, eq, with_weights=True)
    rnames, eff, eq2 = api.read_eq_classes(p)
    assert rnames == list(idx.ref_names()) and np.all(eff == 100.0)
    assert np.array_equal(eq2.off, eq.off) and np.array_equal(eq2.tid, eq.tid) and np.array_equal(eq2.count, eq.count)
    assert np.array_equal(eq2.w, eq.w)          # %.17g round-trips doubles exactly
    # optional trailing "name effLen" pairs
    with gzip.open(p, "at") as f:
        f.write("%s\t123.5\n" % rnames[2])
    _, eff2, _ = api.read_eq_classes(p)
    assert eff2[2] == 123.5 and eff2[0] == 100.0
    # ambig_info.tsv: unique = counts of single-transcript classes, ambiguous = counts of the multi-transcript classes a transcript is in
    api.write_ambig_info(str(tmp_path / "ambig_info.tsv"), M, eq)
    lines = open(tmp_path / "ambig_info.tsv").read().splitlines()
    assert lines[0] == "UniqueCount\tAmbigCount" and len(lines) == M + 1
    uq = np.zeros(M, np.int64); am = np.zeros(M, np.int64)
    for c in range(len(eq.count)):
        ts = eq.tid[int(eq.off[c]):int(eq.off[c + 1])]
        if len(ts) == 1: uq[ts[0]] += int(eq.count[c])
        else: am[ts] += int(eq.count[c])
    assert [tuple(map(int, l.split("\t"))) for l in lines[1:]] == list(zip(uq.tolist(), am.tolist()))
    rows = np.random.default_rng(1).uniform(0, 50, (3, M))
    assert api.write_bootstraps(str(tmp_path / "aux_info"), rnames, rows) == 3
    raw = gzip.open(tmp_path / "aux_info" / "bootstrap" / "bootstraps.gz").read()
    assert np.array_equal(np.frombuffer(raw, np.float64).reshape(3, M), rows)
    assert gzip.open(tmp_path / "aux_info" / "bootstrap" / "names.tsv.gz").read().decode() == "\t".join(rnames) + "\n"
