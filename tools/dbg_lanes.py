import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from salmon_amd import api, synth
tx = synth.Txome(seed=7, n_genes=120, iso_per_gene=6, threads=4)
names, seqs, lens = tx.tables()
idx = api.SalmonIndex.build_mem_raw(tx.n, names, seqs, lens, threads=4)
seq, off, _, _ = tx.reads(4000, read_len=100, seed=11, threads=4)
opts = api.quant_opts(mini_batch_size=500, num_pre_burnin_frags=400, num_burnin_frags=2200)
cuts = [(0, 900), (900, 2100), (2100, 2600), (2600, 4000)]
mode = sys.argv[1] if len(sys.argv) > 1 else "seq"
ctx = api.QuantContext(idx, opts, device=0, max_batch_reads=4096)
def rb(lo, hi):
    s = seq[lo * 200: hi * 200]; o = (off[2 * lo: 2 * hi + 1] - off[2 * lo]).copy()
    return api.make_read_batch(s, o, hi - lo, paired=True), s, o
keep = []
if mode == "seq":
    for lo, hi in cuts:
        b = rb(lo, hi); keep.append(b)
        r = ctx.map_batch(b[0], fetch=(len(sys.argv) > 2)); print("mapped", lo, hi, r[3]["num_alignments"], flush=True)
        if "sync" in sys.argv: torch.cuda.synchronize()
        ctx.eq_accumulate()
        if "sync" in sys.argv: print("summary", ctx.summary(), flush=True)
else:
    bs = [rb(lo, hi) for lo, hi in cuts]
    ctx.map_submit(bs[0][0], fetch=True); ctx.map_submit(bs[1][0], fetch=True)
    for i in range(4):
        r = ctx.map_wait(); print("waited", i, r[3]["num_alignments"], flush=True)
        ctx.eq_accumulate(); print("summary", ctx.summary(), flush=True)
        if i + 2 < 4: ctx.map_submit(bs[i + 2][0], fetch=True)
print("eq", len(ctx.eq_finish().count)); ctx.free(); print("ok")
