"""Run by tests/test_poison_gpu.py with SQ_POISON=1 in the environment: every device work buffer starts filled with 0xA5 instead of the zeros
fresh GPU memory usually holds, so a kernel that reads a word nobody wrote gives itself away.  Prints a JSON object: stage -> equal to the
checker.  (TEST INFRASTRUCTURE; not imported by the product.)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from salmon_amd import api, synth
import orc


def main():
    tx = synth.Txome(seed=9, n_genes=50, iso_per_gene=5, threads=2)
    names, seqs, lens = tx.tables()
    idx = api.SalmonIndex.build_mem_raw(tx.n, names, seqs, lens, threads=2); idx.to_device(0)
    oidx = orc.OrcIndex(idx)
    N = 1800
    seq, off, _, _ = tx.reads(N, read_len=100, seed=3, threads=2)
    opts = api.quant_opts(mini_batch_size=100, num_pre_burnin_frags=80, num_burnin_frags=350)
    ctx = api.QuantContext(idx, opts, device=0, max_batch_reads=2048)
    ost = orc.OrcState(oidx, opts)
    res = {}
    for b in range(3):   # batch 0 crosses burn-in (chain of mini-batches), batches 1-2 take the split path
        lo, hi = b * 600, (b + 1) * 600
        s = seq[lo * 200: hi * 200]; o = (off[2 * lo: 2 * hi + 1] - off[2 * lo]).copy()
        rb = api.make_read_batch(s, o, hi - lo, paired=True)
        ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
        if b == 0:
            um_c, mm_c, ch_c, cd_c = orc.map_taps(oidx, opts, rb)
            for name, what, dt, ref, fields in (("unimems", 1, api.UNIMEM_DTYPE, um_c, ["end", "qpos", "len", "unitig", "uoff", "fw"]),
                                                ("mems", 2, api.MEM_DTYPE, mm_c, ["end", "tid", "rpos", "qpos", "len", "fw"]),
                                                ("chains", 3, api.CHAIN_DTYPE, ch_c, ["end", "tid", "pos", "last_end", "fw", "n_mems", "score"]),
                                                ("candidates", 4, api.CAND_DTYPE, cd_c, ["frag", "tid", "lpos", "rpos", "lfw", "rfw", "mate_status", "valid", "lscore", "rscore", "frag_len"])):
                g = ctx.tap(what, dt)
                res[name] = bool(len(g) == len(ref) and all(np.array_equal(g[f], ref[f]) for f in fields))
        ctx.eq_accumulate()
        ro_c, aln_c, mt_c, st_c = orc.map_batch(oidx, opts, rb, threads=2)
        ost.eq_accumulate(ro_c, aln_c, st_c["num_with_joint_hits"])
        res["batch%d_alignments" % b] = bool(np.array_equal(ro_g, ro_c) and aln_g.tobytes() == aln_c.tobytes() and np.array_equal(mt_g, mt_c))
        res["batch%d_counters" % b] = bool(st_g == st_c)
    ost.finish()
    res["summary"] = bool(ctx.summary() == ost.summary())
    eq_g, eq_c = ctx.eq_finish(), ost.eq_finish()
    res["eq_table"] = bool(all(np.array_equal(getattr(eq_g, f), getattr(eq_c, f)) for f in ["off", "tid", "count", "wq", "bins", "h1", "h2", "w"]))
    mg, mc = ctx.model(), ost.model()
    res["model"] = bool(all(np.array_equal(a, b) for a, b in zip(mg, mc[:4])))
    res["fld"] = bool(np.array_equal(ctx.fld(), mc[4])); res["lib_counts"] = bool(np.array_equal(ctx.lib_counts(), ost.lib_counts()))
    pg = api.normalize_alphas(eq_g, mg[0], mg[1], mg[2]); pc = orc.normalize_alphas(idx.num_refs, eq_c, mc[0], mc[1], mc[2])
    ag, rg = ctx.em_optimize(np.exp(mg[3]), pg, api.em_opts()); ac, rc = orc.em_optimize(eq_c, np.exp(mc[3]), pc, api.em_opts())
    res["em"] = bool(rg["iters"] == rc["iters"] and np.array_equal(ag, ac))
    print("POISON_RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
