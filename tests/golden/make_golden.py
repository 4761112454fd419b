#!/usr/bin/env python
"""Regenerates tests/golden/* from the CPU checker (oracle/).  Run in the build container:

    python tests/golden/make_golden.py

Inputs are produced by tools/synth.cpp (seeded); expected outputs come from the checker's full
pipeline (map -> online model / eq-classes -> normalizeAlphas -> VBEM).  The committed fixtures let
the GPU box (which has no /root/reference and must not depend on regenerating data) check the
product end to end, including the stand-alone `salmon-hip` CLI.
"""
import gzip, hashlib, json, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from salmon_amd import api, synth
import orc

N_PAIRS, READ_LEN = 3000, 75


def main():
    tx = synth.Txome(seed=42, n_genes=40, iso_per_gene=4, threads=2)
    names, seqs = tx.names(), tx.seqs()
    with gzip.open(os.path.join(HERE, "transcripts.fa.gz"), "wt") as f:
        for n, s in zip(names, seqs):
            f.write(">%s\n%s\n" % (n, s.decode()))
    seq, off, tt, tp = tx.reads(N_PAIRS, read_len=READ_LEN, seed=43, threads=2)
    # sprinkle some Ns / lower case so the parser and N handling are covered
    seq = seq.copy(); seq[5::977] = ord("N"); seq[11::1501] = ord("a")
    for mate in (0, 1):
        with gzip.open(os.path.join(HERE, "reads_%d.fq.gz" % (mate + 1)), "wt") as f:
            for i in range(N_PAIRS):
                a = int(off[2 * i + mate]); b = int(off[2 * i + mate + 1])
                f.write("@r%d/%d\n%s\n+\n%s\n" % (i, mate + 1, seq[a:b].tobytes().decode(), "I" * (b - a)))
    idx = api.SalmonIndex.build_mem(names, seqs, threads=2)
    oidx = orc.OrcIndex(idx)
    opts = api.quant_opts()
    rb = api.make_read_batch(seq, off, N_PAIRS, paired=True)
    ro, aln, mt, st = orc.map_batch(oidx, opts, rb, threads=4)
    ost = orc.OrcState(oidx, opts); ost.eq_accumulate(ro, aln, st["num_with_joint_hits"]); ost.finish()
    eq = ost.eq_finish(); lm, uq, tc, le, fld = ost.model()
    M = idx.num_refs
    proj = orc.normalize_alphas(M, eq, lm, uq, tc)
    eff = np.exp(le)
    alphas, rep = orc.em_optimize(eq, eff, proj, api.em_opts())
    denom = float(np.sum((alphas / st["num_with_joint_hits"]) / eff))
    tpm = (alphas / st["num_with_joint_hits"]) / eff / denom * 1e6
    with open(os.path.join(HERE, "golden_quant.sf"), "w") as f:
        f.write("Name\tLength\tEffectiveLength\tTPM\tNumReads\n")
        rn = idx.ref_names(); cl = idx.ref_complete_lens()
        for i in range(M):
            f.write("%s\t%d\t%.3f\t%f\t%.3f\n" % (rn[i], cl[i], eff[i], tpm[i], alphas[i]))
    col = {",".join(map(str, k)): v for k, v in eq.collapsed().items()}
    meta = {"n_pairs": N_PAIRS, "read_len": READ_LEN, "num_refs": M, "stats": st, "summary": ost.summary(), "em_iters": rep["iters"],
            "alignments_sha256": hashlib.sha256(aln.tobytes()).hexdigest(), "read_off_sha256": hashlib.sha256(ro.tobytes()).hexdigest(),
            "num_eq_classes": len(eq.count), "collapsed_eq_classes": col,
            "alphas_hex": [float(a).hex() for a in alphas]}
    json.dump(meta, open(os.path.join(HERE, "golden_meta.json"), "w"), indent=0, sort_keys=True)
    print("wrote golden fixtures: %d refs, %d alignments, %d eq-classes, %d VBEM iterations" % (M, len(aln), len(eq.count), rep["iters"]))


if __name__ == "__main__":
    main()
