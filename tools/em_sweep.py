"""Experiment: EM iteration time on the c2 class table for the block-plan sizes of k_class3 / k_l13 (SQ_EM_CL / SQ_EM_L1).  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from salmon_amd import api, synth, capi
class A: pass
a = A(); a.index_cache = "/tmp/ixc"
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W = bench.World(a, 60000, 4, 0.0, 0.0, 32, 0, api, synth); idx = W.idx
B = 5000000; ctx = api.QuantContext(idx, api.quant_opts(), device=0, max_batch_reads=B); ctx.reserve(1000000, 0)
dev = torch.device("cuda", 0); off_d = torch.from_numpy(np.arange(0, 2 * B + 1, dtype=np.int64) * 100).to(dev)
for i in range(nb):
    t = torch.from_numpy(W.reads(B, 100, i * B)).to(dev); rb = api.make_read_batch(int(t.data_ptr()), int(off_d.data_ptr()), B, paired=True, on_device=True)
    ctx.map_batch(rb, fetch=False); ctx.eq_accumulate(); del t
eq = ctx.eq_finish(); lm, uq, tc, le = ctx.model(); eff = np.exp(le); proj = api.normalize_alphas(eq, lm, uq, tc)
print("classes", len(eq.count), "labels", len(eq.tid), "M", idx.num_refs, flush=True)
base = None
for cl in (2048, 1024, 512):
    for l1 in (2048, 1024, 512):
        os.environ["SQ_EM_CL"] = str(cl); os.environ["SQ_EM_L1"] = str(l1)
        best = 1e9
        for rep in range(3):
            out, r = api.em_steps(eq, eff, np.maximum(proj, 1e-3), 400, api.em_opts(), device=0); best = min(best, r["ms_per_iter"])
        if base is None: base = out
        print("CL %d L1 %d: %.2f us per iteration%s" % (cl, l1, best * 1e3, "" if np.array_equal(out, base) else "  RESULT DIFFERS"), flush=True)
