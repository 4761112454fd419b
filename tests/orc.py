"""ctypes binding of oracle/_build/liboracle.so — TEST INFRASTRUCTURE (the CPU checker).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import numpy as np
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from salmon_amd import capi, api  # struct layouts only

_PATH = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
_lib = None
P = C.POINTER


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(_PATH)
        vp = C.c_void_p
        L.orc_index_from_view.restype = vp; L.orc_index_from_view.argtypes = [P(capi.IndexView), P(C.c_char_p)]
        L.orc_index_free.argtypes = [vp]
        L.orc_index_num_kmers.restype = C.c_uint64; L.orc_index_num_kmers.argtypes = [vp]
        L.orc_index_lookup.restype = C.c_int; L.orc_index_lookup.argtypes = [vp, C.c_uint64, P(C.c_uint64), P(C.c_uint32), P(C.c_int)]
        L.orc_check_cdbg.restype = C.c_int; L.orc_check_cdbg.argtypes = [P(capi.IndexView)]
        L.orc_map_batch.argtypes = [vp, P(capi.QuantOpts), P(capi.ReadBatch), C.c_uint32, vp, vp, C.c_uint64, vp, P(capi.MapStats), P(C.c_uint64)]
        L.orc_map_taps.argtypes = [vp, P(capi.QuantOpts), P(capi.ReadBatch), vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64,
            P(C.c_uint64)]
        L.orc_state_create.restype = vp; L.orc_state_create.argtypes = [vp, P(capi.QuantOpts)]
        L.orc_state_free.argtypes = [vp]
        L.orc_eq_accumulate.argtypes = [vp, C.c_uint32, vp, vp, C.c_uint64]
        L.orc_state_finish.argtypes = [vp]
        L.orc_state_reference_order.argtypes = [vp, vp, C.c_uint64]; L.orc_state_draws_used.restype = C.c_uint64; L.orc_state_draws_used.argtypes = [vp]
        L.orc_state_merge.argtypes = [vp, vp]
        L.orc_state_gc_observed.argtypes = [vp, vp]
        L.orc_bias_gc_eff_lengths.restype = C.c_int; L.orc_bias_gc_eff_lengths.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, vp, vp]
        L.orc_em_optimize_gc.restype = C.c_int; L.orc_em_optimize_gc.argtypes = [P(capi.EqTable), P(capi.TxpIn), P(capi.EmOpts), vp, vp, vp, vp, vp, P(capi.EmReport)]
        L.orc_state_seq_observed.argtypes = [vp, vp, vp, P(C.c_uint64)]
        L.orc_bias_seq_eff_lengths.restype = C.c_int; L.orc_bias_seq_eff_lengths.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_uint32, vp, vp, vp, vp]
        L.orc_em_optimize_bias.restype = C.c_int; L.orc_em_optimize_bias.argtypes = [P(capi.EqTable), P(capi.TxpIn), P(capi.EmOpts), vp, vp, vp, vp, vp, vp, vp, P(capi.EmReport)]
        L.orc_state_pos_observed.argtypes = [vp, vp]
        L.orc_length_classes.restype = C.c_int; L.orc_length_classes.argtypes = [vp, vp, vp]
        L.orc_pos_bin.restype = C.c_int; L.orc_pos_bin.argtypes = [C.c_int32, C.c_uint32]
        L.orc_pos_project.argtypes = [vp, C.c_int32, vp, vp]
        L.orc_spline_eval.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp]
        L.orc_bias_eff_lengths.restype = C.c_int; L.orc_bias_eff_lengths.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_uint32, vp, C.c_uint32, vp, vp, vp, vp]
        L.orc_em_optimize_bias_pos.restype = C.c_int
        L.orc_em_optimize_bias_pos.argtypes = [P(capi.EqTable), P(capi.TxpIn), P(capi.EmOpts), vp, vp, vp, vp, vp, C.c_uint32, vp, vp, vp, P(capi.EmReport)]
        L.orc_state_summary.argtypes = [vp, P(capi.ModelSummary)]
        L.orc_state_lib_counts.argtypes = [vp, vp]
        L.orc_state_fetch.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_eq_finish.argtypes = [vp, P(capi.EqTable)]
        L.orc_normalize_alphas.argtypes = [C.c_uint32, P(capi.EqTable), vp, vp, vp, vp]
        L.orc_em_optimize.restype = C.c_int; L.orc_em_optimize.argtypes = [P(capi.EqTable), P(capi.TxpIn), P(capi.EmOpts), vp, P(capi.EmReport)]
        L.orc_em_steps.restype = C.c_int; L.orc_em_steps.argtypes = [P(capi.EqTable), P(capi.TxpIn), P(capi.EmOpts), vp, C.c_uint32, vp]
        L.orc_bootstrap.restype = C.c_int; L.orc_bootstrap.argtypes = [P(capi.EqTable), P(capi.TxpIn), P(capi.EmOpts), C.c_uint32, C.c_uint64,
            C.c_uint64, vp]
        L.orc_gibbs.restype = C.c_int; L.orc_gibbs.argtypes = [P(capi.EqTable), P(capi.TxpIn), P(capi.GibbsOpts), vp, C.c_uint32, C.c_uint64,
            C.c_uint64, vp]
        L.orc_em_time_iters.restype = C.c_double; L.orc_em_time_iters.argtypes = [P(capi.EqTable), P(capi.TxpIn), P(capi.EmOpts), C.c_uint32,
            C.c_uint32]
        L.orc_canonical_sum.restype = C.c_double; L.orc_canonical_sum.argtypes = [vp, C.c_uint64]
        for f in ("orc_exp", "orc_log", "orc_digamma"):
            getattr(L, f).restype = C.c_double; getattr(L, f).argtypes = [C.c_double]
        L.orc_log_add.restype = C.c_double; L.orc_log_add.argtypes = [C.c_double, C.c_double]
        L.orc_compatible_pe.restype = C.c_int; L.orc_compatible_pe.argtypes = [C.c_int] * 6
        L.orc_compatible_se.restype = C.c_int; L.orc_compatible_se.argtypes = [C.c_int] * 5
        L.orc_format_id.restype = C.c_int; L.orc_format_id.argtypes = [C.c_int] * 3
        L.orc_dp_align.restype = C.c_int; L.orc_dp_align.argtypes = [P(capi.QuantOpts), vp, C.c_int, vp, C.c_int, C.c_int]
        L.orc_fld_prior.argtypes = [C.c_double, C.c_double, vp, P(C.c_double)]
        L.orc_forgetting_mass.restype = C.c_double; L.orc_forgetting_mass.argtypes = [C.c_double, C.c_uint64]
        _lib = L
    return _lib


class OrcIndex:
    def __init__(self, salmon_index):
        self.src = salmon_index
        v = salmon_index.view()
        self.h = C.c_void_p(lib().orc_index_from_view(C.byref(v), None))

    def free(self):
        if self.h:
            lib().orc_index_free(self.h); self.h = None

    def lookup(self, kmer):
        u, off, fw = C.c_uint64(), C.c_uint32(), C.c_int()
        ok = lib().orc_index_lookup(self.h, int(kmer), C.byref(u), C.byref(off), C.byref(fw))
        return (u.value, off.value, bool(fw.value)) if ok else None


def check_cdbg(salmon_index):
    v = salmon_index.view()
    return lib().orc_check_cdbg(C.byref(v))


def map_batch(oidx, opts, rb, threads=1, aln_cap=None):
    n = rb.n
    cap = aln_cap or max(1024, 16 * n)
    read_off = np.zeros(n + 1, np.uint64); aln = np.zeros(cap, api.ALN_DTYPE); mt = np.zeros(n, np.uint8)
    st = capi.MapStats(); na = C.c_uint64()
    lib().orc_map_batch(oidx.h, C.byref(opts), C.byref(rb), threads, read_off.ctypes.data, aln.ctypes.data, cap, mt.ctypes.data, C.byref(st),
        C.byref(na))
    assert na.value <= cap, "oracle alignment buffer too small"
    return read_off, aln[: na.value], mt, st.as_dict()


def map_taps(oidx, opts, rb, cap=1 << 22):
    um = np.zeros(cap, api.UNIMEM_DTYPE); mm = np.zeros(cap, api.MEM_DTYPE); ch = np.zeros(cap, api.CHAIN_DTYPE); cd = np.zeros(cap, api.CAND_DTYPE)
    n = (C.c_uint64 * 4)()
    lib().orc_map_taps(oidx.h, C.byref(opts), C.byref(rb), um.ctypes.data, cap, mm.ctypes.data, cap, ch.ctypes.data, cap, cd.ctypes.data, cap, n)
    assert max(n) <= cap
    return um[: n[0]], mm[: n[1]], ch[: n[2]], cd[: n[3]]


class OrcState:
    def __init__(self, oidx, opts):
        self.oidx = oidx
        self.h = C.c_void_p(lib().orc_state_create(oidx.h, C.byref(opts)))
        self.M = oidx.src.num_refs

    def free(self):
        if self.h:
            lib().orc_state_free(self.h); self.h = None

    def eq_accumulate(self, read_off, aln, num_with_joint_hits=0):
        aln = np.ascontiguousarray(aln)
        lib().orc_eq_accumulate(self.h, len(read_off) - 1, read_off.ctypes.data, aln.ctypes.data, num_with_joint_hits)

    def finish(self):
        lib().orc_state_finish(self.h)

    def reference_order(self, draws):
        """SPEC D1r (pin of row a10): every fragment's increments applied before the next one reads the model; uniform draws from the caller."""
        self._draws = np.ascontiguousarray(draws, np.float64)
        lib().orc_state_reference_order(self.h, self._draws.ctypes.data, len(self._draws))

    def draws_used(self):
        return int(lib().orc_state_draws_used(self.h))

    def seq_observed(self):
        fw = np.zeros(576, np.uint64); rc = np.zeros(576, np.uint64); n = C.c_uint64()
        lib().orc_state_seq_observed(self.h, fw.ctypes.data, rc.ctypes.data, C.byref(n)); return fw, rc, int(n.value)

    def pos_observed(self):
        g = np.zeros(200); lib().orc_state_pos_observed(self.h, g.ctypes.data); return g.reshape(2, 5, 20)

    def gc_observed(self):
        g = np.zeros(75); lib().orc_state_gc_observed(self.h, g.ctypes.data); return g.reshape(3, 25)

    def drop_counts(self):
        lib().orc_state_drop_counts(self.h)

    def merge(self, other):
        """SPEC §MG: fold the state of the next rank into this one (call in rank order on rank 0's state)."""
        lib().orc_state_merge(self.h, other.h)

    def summary(self):
        s = capi.ModelSummary(); lib().orc_state_summary(self.h, C.byref(s))
        return dict(num_observed=int(s.num_observed), num_assigned=int(s.num_assigned), num_mapped_ub=int(s.num_mapped_ub),
            burned_in=bool(s.burned_in), num_compatible=int(s.num_compatible), lib_format_id=int(s.lib_format_id), lib_detected=int(s.lib_detected))

    def lib_counts(self):
        out = np.zeros(64, np.uint64); lib().orc_state_lib_counts(self.h, out.ctypes.data); return out

    def model(self):
        M = self.M
        lm, uq, tc, le, fld = np.zeros(M), np.zeros(M, np.uint64), np.zeros(M, np.uint64), np.zeros(M), np.zeros(1001)
        lib().orc_state_fetch(self.h, lm.ctypes.data, uq.ctypes.data, tc.ctypes.data, le.ctypes.data, fld.ctypes.data)
        return lm, uq, tc, le, fld

    def eq_finish(self):
        t = capi.EqTable(); lib().orc_eq_finish(self.h, C.byref(t))
        eq = api.EqClasses.alloc(int(t.num_classes), int(t.num_labels))
        tt = eq.table(); lib().orc_eq_finish(self.h, C.byref(tt))
        return eq


def normalize_alphas(M, eq, log_mass, uniq, total):
    out = np.zeros(M); t = eq.table()
    lib().orc_normalize_alphas(M, C.byref(t), log_mass.ctypes.data, uniq.ctypes.data, total.ctypes.data, out.ctypes.data)
    return out


def em_optimize(eq, eff_len, projected=None, opts=None, unique=None):
    o = opts or api.em_opts(); t = eq.table(); txp = api.make_txp_in(eff_len, projected, unique)
    out = np.zeros(txp.num_txp); rep = capi.EmReport()
    rc = lib().orc_em_optimize(C.byref(t), C.byref(txp), C.byref(o), out.ctypes.data, C.byref(rep))
    return out, dict(iters=rep.iters, converged=bool(rep.converged), max_rel_diff=rep.max_rel_diff, alpha_sum=rep.alpha_sum, rc=rc,
        num_degenerate=rep.num_degenerate)


def em_steps(eq, eff_len, alpha_in, iters, opts=None):
    o = opts or api.em_opts(); t = eq.table(); txp = api.make_txp_in(eff_len)
    a = np.ascontiguousarray(alpha_in, np.float64); out = np.zeros(txp.num_txp)
    lib().orc_em_steps(C.byref(t), C.byref(txp), C.byref(o), a.ctypes.data, iters, out.ctypes.data)
    return out


def em_time_iters(eq, eff_len, iters, threads, opts=None):
    o = opts or api.em_opts(); t = eq.table(); txp = api.make_txp_in(eff_len)
    return lib().orc_em_time_iters(C.byref(t), C.byref(txp), C.byref(o), iters, threads)


def bootstrap(eq, eff_len, B, seed, num_mapped, opts=None):
    o = opts or api.em_opts(); t = eq.table(); txp = api.make_txp_in(eff_len)
    out = np.zeros((B, txp.num_txp))
    rc = lib().orc_bootstrap(C.byref(t), C.byref(txp), C.byref(o), B, seed, num_mapped, out.ctypes.data)
    assert rc == 0
    return out


def gibbs(eq, eff_len, alpha_init, S, seed, num_mapped, gopts):
    t = eq.table(); txp = api.make_txp_in(eff_len)
    a = np.ascontiguousarray(alpha_init, np.float64); out = np.zeros((S, txp.num_txp))
    rc = lib().orc_gibbs(C.byref(t), C.byref(txp), C.byref(gopts), a.ctypes.data, S, seed, num_mapped, out.ctypes.data)
    assert rc == 0
    return out


def bias_gc_eff_lengths(oidx, gc_obs, log_pmf, alphas, eff_in):
    g = np.ascontiguousarray(gc_obs, np.float64).reshape(-1); lp = np.ascontiguousarray(log_pmf, np.float64)
    a = np.ascontiguousarray(alphas, np.float64); e = np.ascontiguousarray(eff_in, np.float64); out = np.zeros(len(a)); bias = np.zeros(25)
    n = lib().orc_bias_gc_eff_lengths(oidx.h, g.ctypes.data, lp.ctypes.data, len(a), a.ctypes.data, e.ctypes.data, out.ctypes.data, bias.ctypes.data)
    return out, dict(num_processed=n, gc_bias=bias)


def bias_seq_eff_lengths(oidx, seq_fw, seq_rc, log_pmf, alphas, eff_in, gc_obs=None):
    fw = np.ascontiguousarray(seq_fw, np.uint64); rc = np.ascontiguousarray(seq_rc, np.uint64); lp = np.ascontiguousarray(log_pmf, np.float64)
    a = np.ascontiguousarray(alphas, np.float64); e = np.ascontiguousarray(eff_in, np.float64); out = np.zeros(len(a)); models = np.zeros((4, 576))
    g = np.ascontiguousarray(gc_obs, np.float64).reshape(-1) if gc_obs is not None else None
    n = lib().orc_bias_seq_eff_lengths(oidx.h, 1 if g is not None else 0, g.ctypes.data if g is not None else None, fw.ctypes.data, rc.ctypes.data, lp.ctypes.data, len(a),
        a.ctypes.data, e.ctypes.data, out.ctypes.data, models.ctypes.data)
    return out, models, dict(num_processed=n)


def length_classes(oidx):
    q = np.zeros(5, np.uint32); cls = np.zeros(oidx.src.num_refs, np.uint8)
    n = lib().orc_length_classes(oidx.h, q.ctypes.data, cls.ctypes.data); return q[:n], cls


def _opt(a, dt):
    return None if a is None else np.ascontiguousarray(a, dt).reshape(-1)


def bias_eff_lengths(oidx, log_pmf, alphas, eff_in, gc_obs=None, seq=None, pos_obs=None, threads=8):
    """every combination that takes the per-position sweep (SPEC §B2, §P): seq = (fw, rc) counts or None, pos_obs [2][5][20] or None"""
    g = _opt(gc_obs, np.float64); fw = _opt(seq[0], np.uint64) if seq is not None else None; rc = _opt(seq[1], np.uint64) if seq is not None else None
    po = _opt(pos_obs, np.float64); lp = np.ascontiguousarray(log_pmf, np.float64)
    a = np.ascontiguousarray(alphas, np.float64); e = np.ascontiguousarray(eff_in, np.float64); out = np.zeros(len(a)); pm = np.zeros((4, 100))
    d = lambda x: x.ctypes.data if x is not None else None
    n = lib().orc_bias_eff_lengths(oidx.h, 1 if g is not None else 0, d(g), d(fw), d(rc), d(po), threads, lp.ctypes.data, len(a), a.ctypes.data, e.ctypes.data, out.ctypes.data, pm.ctypes.data)
    return out, pm, dict(num_processed=n)


def em_optimize_bias_pos(eq, eff_len, projected, oidx, log_pmf, gc_obs=None, seq=None, pos_obs=None, threads=8, opts=None):
    o = opts or api.em_opts(); t = eq.table(); txp = api.make_txp_in(eff_len, projected)
    g = _opt(gc_obs, np.float64); fw = _opt(seq[0], np.uint64) if seq is not None else None; rc = _opt(seq[1], np.uint64) if seq is not None else None
    po = _opt(pos_obs, np.float64); lp = np.ascontiguousarray(log_pmf, np.float64)
    d = lambda x: x.ctypes.data if x is not None else None
    out = np.zeros(txp.num_txp); eff = np.zeros(txp.num_txp); rep = capi.EmReport()
    rc_ = lib().orc_em_optimize_bias_pos(C.byref(t), C.byref(txp), C.byref(o), oidx.h, d(g), d(fw), d(rc), d(po), threads, lp.ctypes.data, out.ctypes.data, eff.ctypes.data, C.byref(rep))
    return out, eff, dict(iters=rep.iters, converged=bool(rep.converged), rc=rc_, num_degenerate=rep.num_degenerate)


def em_optimize_bias(eq, eff_len, projected, oidx, seq_fw, seq_rc, log_pmf, gc_obs=None, opts=None):
    o = opts or api.em_opts(); t = eq.table(); txp = api.make_txp_in(eff_len, projected)
    fw = np.ascontiguousarray(seq_fw, np.uint64); rc = np.ascontiguousarray(seq_rc, np.uint64); lp = np.ascontiguousarray(log_pmf, np.float64)
    g = np.ascontiguousarray(gc_obs, np.float64).reshape(-1) if gc_obs is not None else None
    out = np.zeros(txp.num_txp); eff = np.zeros(txp.num_txp); rep = capi.EmReport()
    rc_ = lib().orc_em_optimize_bias(C.byref(t), C.byref(txp), C.byref(o), oidx.h, g.ctypes.data if g is not None else None, fw.ctypes.data, rc.ctypes.data, lp.ctypes.data,
        out.ctypes.data, eff.ctypes.data, C.byref(rep))
    return out, eff, dict(iters=rep.iters, converged=bool(rep.converged), rc=rc_, num_degenerate=rep.num_degenerate)


def em_optimize_gc(eq, eff_len, projected, oidx, gc_obs, log_pmf, opts=None):
    o = opts or api.em_opts(); t = eq.table(); txp = api.make_txp_in(eff_len, projected)
    g = np.ascontiguousarray(gc_obs, np.float64).reshape(-1); lp = np.ascontiguousarray(log_pmf, np.float64)
    out = np.zeros(txp.num_txp); eff = np.zeros(txp.num_txp); rep = capi.EmReport()
    rc = lib().orc_em_optimize_gc(C.byref(t), C.byref(txp), C.byref(o), oidx.h, g.ctypes.data, lp.ctypes.data, out.ctypes.data, eff.ctypes.data, C.byref(rep))
    return out, eff, dict(iters=rep.iters, converged=bool(rep.converged), rc=rc, num_degenerate=rep.num_degenerate)
