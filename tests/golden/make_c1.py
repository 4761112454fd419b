#!/usr/bin/env python
"""Makes the C1 fixture (BASELINE.json configs[0]) from the reference's bundled sample data.  Run in the build container:

    python tests/golden/make_c1.py

Input: /root/reference/sample_data.tgz (transcripts.fasta + 10 000 simulated read pairs whose names carry the truth,
`@<n>:<transcript>:<pos>:<fraglen>/<mate>`).  The GPU box has no /root/reference, so the data travel as a committed
fixture under tests/golden/c1/: the transcripts, the two mate files (names and bases kept; base qualities — which
salmon never reads — replaced by a constant so the files compress) and the CPU checker's outputs on them
(alignment / eq-class digests, NumReads, quant.sf).  tests/test_c1.py holds the checker and the HIP path to these.
"""
import gzip, hashlib, json, os, sys, tarfile, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from salmon_amd import api
import orc
import fixtures

SRC = "/root/reference/sample_data.tgz"
OUT = os.path.join(HERE, "c1")


def main():
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as d:
        tarfile.open(SRC).extractall(d)
        sd = os.path.join(d, "sample_data")
        fa = open(os.path.join(sd, "transcripts.fasta")).read()
        with gzip.GzipFile(os.path.join(OUT, "transcripts.fa.gz"), "wb", mtime=0) as f:
            f.write(fa.encode())
        for m in (1, 2):
            lines = open(os.path.join(sd, "reads_%d.fastq" % m)).read().split("\n")
            with gzip.GzipFile(os.path.join(OUT, "reads_%d.fq.gz" % m), "wb", mtime=0) as f:
                for i in range(0, len(lines) - 3, 4):
                    f.write(("%s\n%s\n+\n%s\n" % (lines[i], lines[i + 1], "I" * len(lines[i + 1]))).encode())
    res = fixtures.c1_run_checker(threads=8)
    idx, st, eq, alphas, rep, summ = res["idx"], res["stats"], res["eq"], res["alphas"], res["rep"], res["summary"]
    eff = res["eff"]; M = idx.num_refs
    denom = float(np.sum((alphas / st["num_with_joint_hits"]) / eff))
    tpm = (alphas / st["num_with_joint_hits"]) / eff / denom * 1e6
    with open(os.path.join(OUT, "golden_quant.sf"), "w") as f:
        f.write("Name\tLength\tEffectiveLength\tTPM\tNumReads\n")
        rn = idx.ref_names(); cl = idx.ref_complete_lens()
        for i in range(M):
            f.write("%s\t%d\t%.3f\t%f\t%.3f\n" % (rn[i], cl[i], eff[i], tpm[i], alphas[i]))
    meta = {"n_pairs": res["n"], "num_refs": M, "stats": st, "summary": summ, "em_iters": rep["iters"],
            "alignments_sha256": hashlib.sha256(res["aln"].tobytes()).hexdigest(),
            "read_off_sha256": hashlib.sha256(res["read_off"].tobytes()).hexdigest(),
            "eq_sha256": fixtures.eq_digest(eq), "num_eq_classes": len(eq.count),
            "alphas_hex": [float(a).hex() for a in alphas], "true_recall": res["recall"], "corr_with_truth": res["corr"]}
    json.dump(meta, open(os.path.join(OUT, "c1_meta.json"), "w"), indent=0, sort_keys=True)
    print("wrote C1 fixture: %d refs, %d pairs, %d alignments, %d eq-classes, %d VBEM iterations, recall %.4f, r %.4f" % (M,
        res["n"], len(res["aln"]), len(eq.count), rep["iters"], res["recall"], res["corr"]))


if __name__ == "__main__":
    main()
