"""processMiniBatch (a10–a12), first mini-batch of a run, against a plain-Python restatement of the in-tree code
(SalmonQuantify.cpp:426-1023): per-alignment log-probabilities (fragment coverage, start-position term, prior mass),
per-read normalisation, range-factorized eq-class labels and weights, mass / unique / total updates.  Restricted to
fragments whose alignments are all proper pairs, before burn-in and before the auxiliary models switch on
(< numPreBurninFrags assigned), where the model state is fully known: FLD = prior, mass = prior mass only."""
import math
import numpy as np
from salmon_amd import api
import orc


def test_first_mini_batch_matches_python_restatement(small_world):
    w = small_world; opts = api.quant_opts()
    rb = api.make_read_batch(w["seq"], w["off"], w["n"], paired=True)
    ro, aln, mt, st = orc.map_batch(w["oidx"], opts, rb, threads=4)
    # keep fragments whose alignments are all proper pairs (the orphan term needs the ambiguous-length table)
    keep = [r for r in range(w["n"]) if ro[r + 1] > ro[r] and np.all(aln["mate_status"][ro[r]:ro[r + 1]] == 3)]
    assert len(keep) > 0.9 * w["n"] and len(keep) < 5000           # one mini-batch, no auxiliary model yet
    sel = np.concatenate([np.arange(int(ro[r]), int(ro[r + 1])) for r in keep]); a2 = np.ascontiguousarray(aln[sel])
    off2 = np.zeros(len(keep) + 1, np.uint64); off2[1:] = np.cumsum([int(ro[r + 1] - ro[r]) for r in keep])
    s = orc.OrcState(w["oidx"], opts); s.eq_accumulate(off2, a2, len(keep))
    lm, uq, tc, le, fld = s.model(); eq = s.eq_finish(); summ = s.summary(); s.free()
    # ---- Python restatement
    lens = w["idx"].ref_lens().astype(np.int64); M = len(lens)
    prior = [math.log(0.005 * float(l)) for l in lens]              # Transcript.hpp:48-56, alpha = 0.005 (ReadExperiment.inl:114)
    classes = {}; mass_add = [0.0] * M; uniq = [0] * M; total = [0] * M
    def pedantic(a, T):                                             # ReadPair.hpp:149-168
        p1 = int(a["pos"]) if a["fwd"] else int(a["mate_pos"]); p1 = min(max(p1, 0), T)
        p2 = int(a["mate_pos"]) + int(a["mate_len"]) if a["fwd"] else int(a["pos"]) + int(a["read_len"]); p2 = min(max(p2, 0), T)
        return abs(p1 - p2)
    for r in range(len(keep)):
        al = a2[int(off2[r]):int(off2[r + 1])]
        tids, aux, lp = [], [], []
        for a in al:
            t = int(a["tid"]); T = int(lens[t]); ref_len = float(T) if T > 0 else 1.0
            flen = pedantic(a, T) if a["fwd"] != a["mate_fwd"] else int(a["frag_len"])
            log_cov = math.log(a["est_aln_prob"]) if a["est_aln_prob"] > 0 else 0.0          # :609-610
            log_frag = 0.0                                          # no FLD term before numPreBurninFrags (:661-681)
            start = -math.log(ref_len - flen + 1.0) if flen <= ref_len else math.log(0.375e-10)   # :749-757
            auxp = log_frag + log_cov                               # compatible under IU: logAlignCompatProb = 0 (:777)
            tids.append(t); aux.append(auxp); lp.append(prior[t] + auxp + start)               # mass(initialRound) = prior (:779)
        mx = max(lp); tot = mx + math.log(sum(math.exp(x - mx) for x in lp))
        ma = max(aux); den = ma + math.log(sum(math.exp(x - ma) for x in aux))
        wts = [math.exp(x - den) for x in aux]                      # :818-820
        n = len(tids); rc = int(math.sqrt(n) + opts.range_factorization_bins)
        label = tuple(tids) + tuple(int(x * rc) for x in wts)       # :845-853
        c = classes.setdefault(label, [0, [0.0] * n]); c[0] += 1
        for i in range(n): c[1][i] += wts[i]
        for i in range(n):
            mass_add[tids[i]] += math.exp(lp[i] - tot); total[tids[i]] += 1                    # :871-872, :986-989
        if n == 1: uniq[tids[0]] += 1
    # ---- compare
    assert summ["num_assigned"] == len(keep)
    got = {}
    for c in range(len(eq.count)):
        a, b = int(eq.off[c]), int(eq.off[c + 1])
        got[tuple(int(x) for x in eq.tid[a:b]) + tuple(int(x) for x in eq.bins[a:b])] = (int(eq.count[c]), eq.wq[a:b].astype(np.float64) / 2.0 ** 36)
    assert set(got) == set(classes)
    for lab, (cnt, wsum) in classes.items():
        assert got[lab][0] == cnt
        assert np.allclose(got[lab][1], wsum, rtol=0, atol=cnt * 2.0 ** -35 + 1e-9)             # fixed-point sums (2^-36 per addend)
    assert np.array_equal(tc, np.array(total, np.uint64)) and np.array_equal(uq, np.array(uniq, np.uint64))
    for t in range(M):                                              # mass = logAdd(LOG_0, logForgettingMass(0) = 0 + log(sum))
        if mass_add[t] > 0: assert abs(lm[t] - math.log(mass_add[t])) < 1e-7, t
        else: assert math.isinf(lm[t])
