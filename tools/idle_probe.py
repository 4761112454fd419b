"""Experiment: where does the 15-30 ms gap between the eq-class export and the first EM kernel come from?
Runs the bench job (nb batches) and times tiny GPU round trips (torch stream) at points of the tail."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from salmon_amd import api, synth
tx = synth.Txome(seed=1, n_genes=20000, iso_per_gene=10, threads=32)
names, seqs, lens = tx.tables()
idx = api.SalmonIndex.build_mem_raw(tx.n, names, seqs, lens, threads=64); idx.to_device(0)
B = 1000000; opts = api.quant_opts(); ctx = api.QuantContext(idx, opts, device=0, max_batch_reads=B); ctx.reserve()
dev = torch.device("cuda", 0)
off_d = torch.from_numpy(np.arange(0, 2 * B + 1, dtype=np.int64) * 100).to(dev)
bs = []
for s in range(10):
    seq, off, _, _ = tx.reads(B, read_len=100, seed=2, first_pair=s * B, threads=64, truth=False); bs.append(torch.from_numpy(seq).to(dev))
rbs = [api.make_read_batch(int(b.data_ptr()), int(off_d.data_ptr()), B, paired=True, on_device=True) for b in bs]
x = torch.zeros(16, device=dev)
def probe():
    t = time.perf_counter(); x.add_(1); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
probe()
for mode, nb in (("plain", 3), ("plain", 10), ("probe", 10), ("plain", 10), ("probe", 10), ("plain", 6)):
    ctx.reset()
    for rb in rbs[:nb]:
        ctx.map_batch(rb, fetch=False); ctx.eq_accumulate()
    t0 = time.perf_counter(); eq = ctx.eq_finish(); t_eq = (time.perf_counter() - t0) * 1e3
    p1 = probe() if mode == "probe" else -1
    lm, uq, tc, le = ctx.model()
    proj = api.normalize_alphas(eq, lm, uq, tc)
    p2 = probe() if mode == "probe" else -1
    t1 = time.perf_counter()
    a, rep = ctx.em_optimize(np.exp(le), proj, api.em_opts())
    t2 = time.perf_counter()
    print("%-6s nb=%2d eq_finish %.1f ms  probe1 %.2f  probe2 %.2f  em_call %.1f  device %.1f  overhead %.1f ms" % (mode, nb, t_eq, p1, p2, (t2 - t1) * 1e3, rep["device_ms"], (t2 - t1) * 1e3 - rep["device_ms"]))
