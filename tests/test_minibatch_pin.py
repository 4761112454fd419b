"""[r6] Row a10 against the reference's own code: processMiniBatch<QuasiAlignment> (src/quant/SalmonQuantify.cpp:426-1023), EquivalenceClassBuilder::addGroup / finish
(include/salmon/internal/quant/EquivalenceClassBuilder.hpp:165-181,237-250) over the vendored libcuckoo map, the burn-in's updateTranscriptLengthsAtomic
(ReadExperiment.inl:62-94) and normalizeAlphas (src/util/SalmonUtils.cpp:461-529), compiled from where they lie under /root/reference into
oracle/_ref/libminibatch_ref.so (oracle/ref_minibatch_shim.cpp says what is the reference's and what is stood in for; pufferfish's QuasiAlignment accessors are stand-ins, so
row a6 stays unpinned).  One thread, one mini-batch in flight: the reference is then deterministic given its random engine, whose draws the shim reports.

The checker runs the same mini-batches in its REFERENCE-ORDER mode (SPEC D1r: every fragment's increments are applied before the next fragment reads the model, the
uniform draws are the reference's) — the per-alignment arithmetic, the labels, the burn-in, the counters are the code every other mode runs; only the moment at which
increments become visible differs from the default mode (SPEC D1, the documented deviation).  Held: after EVERY mini-batch the transcript masses, unique / total
counts, the FLD, effective lengths, the assigned count and the burned-in flag; at the end every class label (transcripts + range-factorisation bins) and count exactly,
the normalised weights to 1e-9; then normalizeAlphas' projected counts.  The mini-batches cross numPreBurninFrags (the auxiliary model switches on inside mini-batch 2)
and the burn-in (after mini-batch 3).  Skipped where the library was not built."""
import ctypes as C, os
import numpy as np
import pytest
from salmon_amd import api, capi
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class MbOpts(C.Structure):
    _fields_ = [("lib_type", C.c_uint32), ("lib_orientation", C.c_uint32), ("lib_strand", C.c_uint32), ("range_factorization_bins", C.c_uint32), ("num_burnin_frags", C.c_uint64),
                ("num_pre_burnin_frags", C.c_uint64), ("incompat_prior", C.c_double), ("forgetting_factor", C.c_double), ("fld_max", C.c_uint32), ("fld_mean", C.c_uint32), ("fld_sd", C.c_uint32),
                ("ignore_incompat", C.c_uint8), ("no_eff_length_correction", C.c_uint8), ("no_length_correction", C.c_uint8), ("no_frag_length_dist", C.c_uint8), ("no_single_frag_prob", C.c_uint8),
                ("rank_eq_classes", C.c_uint8), ("_pad", C.c_uint8 * 2), ("engine_seed", C.c_uint64)]


def _ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libminibatch_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libminibatch_ref.so not built (no /root/reference on this machine)")
    L = C.CDLL(path); vp = C.c_void_p
    L.ref_mb_create.restype = vp; L.ref_mb_create.argtypes = [C.c_uint32, vp, vp, C.POINTER(MbOpts)]
    L.ref_mb_free.argtypes = [vp]
    L.ref_mb_process.restype = C.c_uint64; L.ref_mb_process.argtypes = [vp, C.c_uint32, vp, vp, vp, vp]
    L.ref_mb_state.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    L.ref_mb_eq_finish.restype = C.c_uint64; L.ref_mb_eq_finish.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ref_mb_eq_fetch.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ref_mb_normalize_alphas.argtypes = [vp, C.c_uint64, vp]
    return L


@pytest.fixture(scope="module")
def mapped(built):
    """24 000 pairs of a small transcriptome mapped by the checker: proper pairs, orphans, multi-mappers."""
    from salmon_amd import synth
    tx = synth.Txome(seed=21, n_genes=150, iso_per_gene=6, threads=4); names, seqs, lens = tx.tables()
    idx = api.SalmonIndex.build_mem_raw(tx.n, names, seqs, lens, threads=4); oidx = orc.OrcIndex(idx)
    n = 24000; seq, off, _, _ = tx.reads(n, read_len=100, seed=31, threads=4)
    return dict(idx=idx, oidx=oidx, seq=seq, off=off, n=n)


VARIANTS = {
    "IU defaults": dict(lib="IU"),
    "ISF, incompatible alignments dropped": dict(lib="ISF", num_burnin_frags=8000, num_pre_burnin_frags=3000),     # (half the fragments of this unstranded data are dropped)
    "ISR, incompatible alignments kept with the prior": dict(lib="ISR", ignore_incompat=0, incompat_prior=float(np.log(1e-3))),
    "no range factorisation, no effective-length correction": dict(lib="IU", range_factorization_bins=0, no_eff_length_correction=1),
    "no fragment-length distribution, orphans at LOG_EPSILON": dict(lib="IU", use_frag_len_dist=0, model_single_frag_prob=0),
    "no length correction": dict(lib="IU", no_length_correction=1),
}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_mini_batches_follow_the_compiled_reference(mapped, name):
    L = _ref(); w = mapped; kw = dict(VARIANTS[name]); libname = kw.pop("lib")
    opts = api.quant_opts(**dict(dict(num_burnin_frags=13000), **kw)); api.set_libtype(opts, libname)
    rb = api.make_read_batch(w["seq"], w["off"], w["n"], paired=True)
    ro, aln, mt, st = orc.map_batch(w["oidx"], opts, rb, threads=4)
    assert (aln["mate_status"] != 3).sum() > 100 and (np.diff(ro.astype(np.int64)) > 1).sum() > 2000          # orphans and multi-mappers are there
    idx = w["idx"]; M = idx.num_refs; rl = np.ascontiguousarray(idx.ref_lens(), np.uint32); cl = np.ascontiguousarray(idx.ref_complete_lens(), np.uint32)
    mo = MbOpts(lib_type=opts.lib_type, lib_orientation=opts.lib_orientation, lib_strand=opts.lib_strand, range_factorization_bins=opts.range_factorization_bins,
                num_burnin_frags=opts.num_burnin_frags, num_pre_burnin_frags=opts.num_pre_burnin_frags, incompat_prior=opts.incompat_prior, forgetting_factor=opts.forgetting_factor,
                fld_max=1000, fld_mean=int(opts.fld_mean), fld_sd=int(opts.fld_sd), ignore_incompat=opts.ignore_incompat, no_eff_length_correction=opts.no_eff_length_correction,
                no_length_correction=opts.no_length_correction, no_frag_length_dist=0 if opts.use_frag_len_dist else 1, no_single_frag_prob=0 if opts.model_single_frag_prob else 1,
                rank_eq_classes=0, engine_seed=12345)
    R = L.ref_mb_create(M, rl.ctypes.data, cl.ctypes.data, C.byref(mo)); S = orc.OrcState(w["oidx"], opts)
    mb = 5000; nmb = 0; assigned_before = 0; crossed_aux = crossed_burn = False
    try:
        for r0 in range(0, w["n"], mb):
            r1 = min(w["n"], r0 + mb); a0, a1 = int(ro[r0]), int(ro[r1])
            off = np.ascontiguousarray(ro[r0:r1 + 1] - ro[r0], np.uint64); al = np.ascontiguousarray(aln[a0:a1])
            draws = np.zeros(max(1, a1 - a0)); lp = np.zeros(max(1, a1 - a0))
            used = L.ref_mb_process(R, r1 - r0, off.ctypes.data, al.ctypes.data, draws.ctypes.data, lp.ctypes.data)
            S.reference_order(draws); S.eq_accumulate(off, al, 0)
            assert S.draws_used() == used, (nmb, S.draws_used(), used)                   # the same alignments were kept
            lm_r, uq_r, tc_r, le_r, fld_r = np.zeros(M), np.zeros(M, np.uint64), np.zeros(M, np.uint64), np.zeros(M), np.zeros(1001)
            na = C.c_uint64(); bi = C.c_int(); nc = C.c_uint64()
            L.ref_mb_state(R, lm_r.ctypes.data, uq_r.ctypes.data, tc_r.ctypes.data, le_r.ctypes.data, fld_r.ctypes.data, C.byref(na), C.byref(bi), C.byref(nc))
            lm, uq, tc, le, fld = S.model(); summ = S.summary()
            assert summ["num_assigned"] == na.value and summ["burned_in"] == bool(bi.value) and summ["num_compatible"] == nc.value, (nmb, summ, na.value, bi.value, nc.value)
            assert np.array_equal(uq, uq_r) and np.array_equal(tc, tc_r), nmb
            assert np.array_equal(np.isinf(lm), np.isinf(lm_r)), nmb
            f = ~np.isinf(lm); assert np.allclose(lm[f], lm_r[f], rtol=0, atol=1e-9), (nmb, np.abs(lm[f] - lm_r[f]).max())
            assert np.allclose(fld[1:], fld_r[1:], rtol=0, atol=1e-9), (nmb, np.abs(fld[1:] - fld_r[1:]).max())      # bin 0: never read (a fragment has a length)
            if summ["burned_in"]: assert np.allclose(le, le_r, rtol=0, atol=1e-9), (nmb, np.abs(le - le_r).max())
            crossed_aux |= assigned_before < opts.num_pre_burnin_frags <= na.value; crossed_burn |= assigned_before < opts.num_burnin_frags <= na.value
            assigned_before = na.value; nmb += 1
        assert nmb >= 4 and crossed_aux and crossed_burn and bool(bi.value)
        # ---- the class table: EquivalenceClassBuilder::finish
        Lt = C.c_uint64(); E = L.ref_mb_eq_finish(R, C.byref(Lt)); eq = S.eq_finish()
        off_r = np.zeros(E + 1, np.uint64); lab_r = np.zeros(Lt.value, np.uint32); cnt_r = np.zeros(E, np.uint64); w_r = np.zeros(Lt.value); woff_r = np.zeros(E + 1, np.uint64)
        L.ref_mb_eq_fetch(R, off_r.ctypes.data, lab_r.ctypes.data, cnt_r.ctypes.data, w_r.ctypes.data, woff_r.ctypes.data)
        rf = opts.range_factorization_bins > 0
        ref = {}; got = {}
        for c in range(E):
            k = tuple(int(x) for x in lab_r[int(off_r[c]):int(off_r[c + 1])]); n = len(k) // 2 if rf else len(k)
            ref[(k[:n], k[n:])] = (int(cnt_r[c]), w_r[int(woff_r[c]):int(woff_r[c + 1])] * int(cnt_r[c]))              # weight SUMS (finish() divided them by their total = the count)
        for c in range(len(eq.count)):
            a, b = int(eq.off[c]), int(eq.off[c + 1])
            got[(tuple(int(x) for x in eq.tid[a:b]), tuple(int(x) for x in eq.bins[a:b]) if rf else ())] = (int(eq.count[c]), eq.w[a:b] * int(eq.count[c]))
        # SPEC D5: exp / log are include/sq_math.h's operation sequences here (<= 2 ulp from the reference's libm).  An auxiliary probability that lies ON a bin
        # boundary — six equally likely transcripts: 1/6 x int(sqrt(6) + 4) = 1 — falls to either side by its last bit, on both sides, in the reference by ITS libm's
        # last bit.  A class all of whose fragments sit on a boundary at some position (its mean probability x rangeCount is an integer to 1e-9 there) is therefore
        # counted with the bin above at that position, on both sides; nothing else is touched, and everything is then exact
        def canonical(table):
            out = {}; moved = 0
            for (tids, bins), (cnt, wsum) in table.items():
                if rf:
                    n = len(tids); x = wsum / cnt * int(np.sqrt(n) + opts.range_factorization_bins)
                    nb = tuple(int(round(x[i])) if abs(x[i] - round(x[i])) < 1e-9 else b for i, b in enumerate(bins)); moved += nb != bins; bins = nb
                c0, w0 = out.get((tids, bins), (0, 0.0)); out[(tids, bins)] = (c0 + cnt, w0 + wsum)
            return out, moved
        n_ref, n_got = len(ref), len(got)
        ref, moved_r = canonical(ref); got, moved_g = canonical(got)
        assert moved_r <= 0.1 * n_ref and moved_g <= 0.1 * n_got, (moved_r, moved_g)       # (this transcriptome has six isoforms per gene: the 1/6 case is as common as it gets)
        assert set(ref) == set(got), (len(ref), len(got), sorted(set(ref) ^ set(got))[:4])
        worst = 0.0
        for key, (cnt, wsum) in got.items():
            assert ref[key][0] == cnt, key
            worst = max(worst, float(np.abs(ref[key][1] - wsum).max() / cnt))
        assert worst < 1e-9, worst
        # ---- normalizeAlphas on what the mini-batches left
        nm = int(eq.count.sum()); proj_r = np.zeros(M); L.ref_mb_normalize_alphas(R, nm, proj_r.ctypes.data)
        proj = orc.normalize_alphas(M, eq, lm, uq, tc)
        assert np.array_equal(proj == 0, proj_r == 0) and np.allclose(proj, proj_r, rtol=1e-9, atol=1e-9), np.abs(proj - proj_r).max()
    finally:
        L.ref_mb_free(R); S.free()
