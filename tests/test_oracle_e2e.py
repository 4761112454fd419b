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
