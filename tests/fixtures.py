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
