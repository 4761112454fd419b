"""Loaders for the committed golden fixtures (tests/golden/)."""
import gzip, json, os
import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fasta():
    names, seqs = [], []
    with gzip.open(os.path.join(G, "transcripts.fa.gz"), "rt") as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                names.append(line[1:].split()[0]); seqs.append([])
            elif line:
                seqs[-1].append(line)
    return names, ["".join(s) for s in seqs]


def load_reads():
    mates = []
    for m in (1, 2):
        with gzip.open(os.path.join(G, "reads_%d.fq.gz" % m), "rt") as f:
            lines = f.read().split("\n")
        mates.append(lines[1::4])
    n = len(mates[0])
    recs = []
    for i in range(n):
        recs.append(mates[0][i].encode()); recs.append(mates[1][i].encode())
    seq = np.frombuffer(b"".join(recs), np.uint8).copy()
    off = np.zeros(2 * n + 1, np.uint64); off[1:] = np.cumsum([len(r) for r in recs])
    return seq, off, n


def meta():
    return json.load(open(os.path.join(G, "golden_meta.json")))


# ---- C1 fixture (the reference's bundled sample data; tests/golden/make_c1.py) ------------------------------------
C1 = os.path.join(G, "c1")


def c1_meta():
    return json.load(open(os.path.join(C1, "c1_meta.json")))


def c1_load():
    """-> dict(names, seqs, seq, off, n, truth): transcripts, interleaved read pairs, true transcript name per pair."""
    names, seqs = [], []
    with gzip.open(os.path.join(C1, "transcripts.fa.gz"), "rt") as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                names.append(line[1:].split()[0]); seqs.append([])
            elif line:
                seqs[-1].append(line)
    mates, truth = [], []
    for m in (1, 2):
        with gzip.open(os.path.join(C1, "reads_%d.fq.gz" % m), "rt") as f:
            lines = f.read().split("\n")
        mates.append(lines[1::4])
        if m == 1:
            truth = [h[1:].split(":")[1].split("/")[0] for h in lines[0::4] if h]
    n = len(truth)
    recs = []
    for i in range(n):
        recs.append(mates[0][i].encode()); recs.append(mates[1][i].encode())
    seq = np.frombuffer(b"".join(recs), np.uint8).copy()
    off = np.zeros(2 * n + 1, np.uint64); off[1:] = np.cumsum([len(r) for r in recs])
    return dict(names=names, seqs=["".join(s) for s in seqs], seq=seq, off=off, n=n, truth=truth)


def eq_digest(eq):
    """sha256 over the canonical-order class table: offsets, transcript ids, bins, counts and fixed-point weight sums."""
    import hashlib
    h = hashlib.sha256()
    for f in ("off", "tid", "bins", "count", "wq"):
        h.update(np.ascontiguousarray(getattr(eq, f)).tobytes())
    return h.hexdigest()


def truth_scores(idx, truth, read_off, aln, alphas):
    """(recall of the true transcript among the alignments of mapped pairs, Pearson r of NumReads with the true counts)."""
    name2tid = {n: i for i, n in enumerate(idx.ref_names())}
    hit = mapped = 0
    for i, t in enumerate(truth):
        a0, a1 = int(read_off[i]), int(read_off[i + 1])
        if a1 > a0:
            mapped += 1; hit += int(name2tid.get(t, -1) in aln["tid"][a0:a1])
    true = np.zeros(idx.num_refs)
    for t in truth:
        if t in name2tid: true[name2tid[t]] += 1
    return hit / max(1, mapped), float(np.corrcoef(alphas, true)[0, 1])


def c1_run_checker(threads=4):
    """The CPU checker's whole pipeline on the C1 fixture."""
    from salmon_amd import api
    import orc
    d = c1_load()
    idx = api.SalmonIndex.build_mem(d["names"], d["seqs"], threads=2)
    oidx = orc.OrcIndex(idx); opts = api.quant_opts()
    rb = api.make_read_batch(d["seq"], d["off"], d["n"], paired=True)
    ro, aln, mt, st = orc.map_batch(oidx, opts, rb, threads=threads)
    ost = orc.OrcState(oidx, opts); ost.eq_accumulate(ro, aln, st["num_with_joint_hits"]); ost.finish()
    eq = ost.eq_finish(); lm, uq, tc, le, fld = ost.model()
    proj = orc.normalize_alphas(idx.num_refs, eq, lm, uq, tc)
    alphas, rep = orc.em_optimize(eq, np.exp(le), proj, api.em_opts())
    rec, r = truth_scores(idx, d["truth"], ro, aln, alphas)
    return dict(idx=idx, n=d["n"], read_off=ro, aln=aln, stats=st, eq=eq, alphas=alphas, rep=rep, summary=ost.summary(), eff=np.exp(le),
                recall=rec, corr=r, data=d)
