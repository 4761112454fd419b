"""Host-side mirror of the reference's seams over the C ABI (numpy in / numpy out).

Names follow the reference: SalmonIndex (include/salmon/internal/index/SalmonIndex.hpp),
processReads -> QuantContext.map_batch (src/quant/SalmonQuantify.cpp:1026-1874),
EquivalenceClassBuilder.finish -> QuantContext.eq_finish, CollapsedEMOptimizer.optimize ->
em_optimize (src/inference/CollapsedEMOptimizer.cpp:732-1035).  All compute happens in
libsalmon_hip.so on a gfx950 device; nothing here computes.
"""
import ctypes as C
import numpy as np
from . import capi
from .capi import check, lib

ALN_DTYPE = np.dtype([("tid", "<u4"), ("pos", "<i4"), ("mate_pos", "<i4"), ("score", "<i4"), ("mate_score", "<i4"),
                      ("frag_len", "<u4"), ("read_len", "<u2"), ("mate_len", "<u2"), ("fwd", "u1"), ("mate_fwd", "u1"),
                      ("mate_status", "u1"), ("format_id", "u1"), ("est_aln_prob", "<f8")], align=True)
UNIMEM_DTYPE = np.dtype([("end", "<u4"), ("qpos", "<u2"), ("len", "<u2"), ("unitig", "<u8"), ("uoff", "<u4"), ("fw", "u1")], align=True)
MEM_DTYPE = np.dtype([("end", "<u4"), ("tid", "<u4"), ("rpos", "<i4"), ("qpos", "<u2"), ("len", "<u2"), ("fw", "u1")], align=True)
CHAIN_DTYPE = np.dtype([("end", "<u4"), ("tid", "<u4"), ("pos", "<i4"), ("last_end", "<i4"), ("fw", "u1"), ("n_mems", "<u4"), ("score", "<f8")],
    align=True)
CAND_DTYPE = np.dtype([("frag", "<u4"), ("tid", "<u4"), ("lpos", "<i4"), ("rpos", "<i4"), ("lfw", "u1"), ("rfw", "u1"),
                       ("mate_status", "u1"), ("valid", "u1"), ("lscore", "<i4"), ("rscore", "<i4"), ("frag_len", "<u4")], align=True)
assert ALN_DTYPE.itemsize == C.sizeof(capi.Aln)
assert UNIMEM_DTYPE.itemsize == C.sizeof(capi.UniMem) and MEM_DTYPE.itemsize == C.sizeof(capi.Mem)
assert CHAIN_DTYPE.itemsize == C.sizeof(capi.Chain) and CAND_DTYPE.itemsize == C.sizeof(capi.Cand)


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def quant_opts(**kw):
    o = capi.QuantOpts()
    lib().sq_quant_opts_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def mimic_bt2(o, strict=False):
    """--mimicBT2 / --mimicStrictBT2 on top of `o` (QuantOptionsUtils.cpp:256-294)."""
    rc = lib().sq_quant_opts_mimic_bt2(C.byref(o), 1 if strict else 0)
    if rc: raise ValueError("sq_quant_opts_mimic_bt2: %d" % rc)
    return o


LIBTYPES = {  # src/util/LibraryTypeUtils.cpp:22-46 -> (type, orientation, strandedness)
    "IU": (1, 2, 4), "ISF": (1, 2, 0), "ISR": (1, 2, 1), "OU": (1, 1, 4), "OSF": (1, 1, 0), "OSR": (1, 1, 1),
    "MU": (1, 0, 4), "MSF": (1, 0, 2), "MSR": (1, 0, 3), "U": (0, 3, 4), "SF": (0, 3, 2), "SR": (0, 3, 3)}


def set_libtype(o, name):
    t, orient, s = LIBTYPES[name.upper()]
    o.lib_type, o.lib_orientation, o.lib_strand = t, orient, s
    return o


def em_opts(**kw):
    o = capi.EmOpts()
    lib().sq_em_opts_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class SalmonIndex:
    """B0: `salmon index` + index handle."""

    def __init__(self, handle):
        self.h = C.c_void_p(handle)

    @staticmethod
    def build(fasta, outdir, decoys=None, k=31, m=0, threads=0, keep_duplicates=False, no_clip=False, gencode=False):
        o = capi.IndexOpts(k, m, int(keep_duplicates), int(no_clip), threads, int(gencode))
        check(lib().sq_index_build(C.byref(o), fasta.encode(), decoys.encode() if decoys else None, outdir.encode()), "sq_index_build")

    @staticmethod
    def build_mem(names, seqs, k=31, m=0, threads=0, first_decoy=None, outdir=None, keep_duplicates=False, no_clip=False):
        n = len(names)
        o = capi.IndexOpts(k, m, int(keep_duplicates), int(no_clip), threads, 0)
        nm = (C.c_char_p * n)(*[x.encode() if isinstance(x, str) else x for x in names])
        sb = [x.encode() if isinstance(x, str) else bytes(x) for x in seqs]
        sq = (C.c_char_p * n)(*sb)
        ln = (C.c_uint32 * n)(*[len(x) for x in sb])
        out = C.c_void_p()
        check(lib().sq_index_build_mem(C.byref(o), n, nm, sq, ln, n if first_decoy is None else first_decoy,
                                       outdir.encode() if outdir else None, C.byref(out)), "sq_index_build_mem")
        return SalmonIndex(out.value)

    @staticmethod
    def build_mem_raw(n, names_p, seqs_p, lens_p, k=31, m=0, threads=0, first_decoy=None, outdir=None):
        o = capi.IndexOpts(k, m, 0, 0, threads, 0)
        out = C.c_void_p()
        check(lib().sq_index_build_mem(C.byref(o), n, names_p, seqs_p, lens_p, n if first_decoy is None else first_decoy,
                                       outdir.encode() if outdir else None, C.byref(out)), "sq_index_build_mem")
        return SalmonIndex(out.value)

    @staticmethod
    def load(dirname, device=-1):
        out = C.c_void_p()
        check(lib().sq_index_load(dirname.encode(), device, C.byref(out)), "sq_index_load")
        return SalmonIndex(out.value)

    def to_device(self, device=0):
        check(lib().sq_index_to_device(self.h, device), "sq_index_to_device")
        return self

    def free(self):
        if self.h:
            lib().sq_index_free(self.h)
            self.h = None

    k = property(lambda s: lib().sq_index_k(s.h))
    m = property(lambda s: lib().sq_index_m(s.h))
    num_refs = property(lambda s: lib().sq_index_num_refs(s.h))
    first_decoy = property(lambda s: lib().sq_index_first_decoy(s.h))
    num_unitigs = property(lambda s: lib().sq_index_num_unitigs(s.h))
    num_kmers = property(lambda s: lib().sq_index_num_kmers(s.h))
    device_bytes = property(lambda s: lib().sq_index_device_bytes(s.h))

    def ref_names(self):
        return [lib().sq_index_ref_name(self.h, i).decode() for i in range(self.num_refs)]

    def ref_lens(self):
        return np.array([lib().sq_index_ref_len(self.h, i) for i in range(self.num_refs)], dtype=np.uint32)

    def ref_complete_lens(self):
        return np.array([lib().sq_index_ref_complete_len(self.h, i) for i in range(self.num_refs)], dtype=np.uint32)

    def view(self):
        v = capi.IndexView()
        check(lib().sq_index_get_view(self.h, C.byref(v)), "sq_index_get_view")
        return v

    def lookup_host(self, kmer):
        u, off, fw = C.c_uint64(), C.c_uint32(), C.c_int()
        ok = lib().sq_index_lookup_host(self.h, int(kmer), C.byref(u), C.byref(off), C.byref(fw))
        return (u.value, off.value, bool(fw.value)) if ok else None


def make_read_batch(seq, seq_off, n, paired=True, on_device=False):
    """seq: uint8 array (or device pointer int), seq_off: uint64 array of nrec+1 (or device pointer)."""
    rb = capi.ReadBatch()
    rb.n = n
    rb.paired = int(paired)
    rb.seq = seq if isinstance(seq, int) else seq.ctypes.data
    rb.seq_off = seq_off if isinstance(seq_off, int) else seq_off.ctypes.data
    rb.on_device = int(on_device)
    rb._keep = (seq, seq_off)   # the struct holds raw pointers: the arrays must outlive it
    return rb


def eq_table_from_arrays(off, tid, w, count, wq=None, bins=None, h1=None, h2=None):
    t = capi.EqTable()
    t.num_classes = len(count)
    t.num_labels = len(tid)
    t.off = _ptr(off, C.c_uint64); t.tid = _ptr(tid, C.c_uint32); t.w = _ptr(w, C.c_double); t.count = _ptr(count, C.c_uint64)
    if wq is not None: t.wq = _ptr(wq, C.c_uint64)
    if bins is not None: t.bins = _ptr(bins, C.c_uint32)
    if h1 is not None: t.h1 = _ptr(h1, C.c_uint64)
    if h2 is not None: t.h2 = _ptr(h2, C.c_uint64)
    t._keep = (off, tid, w, count, wq, bins, h1, h2)
    return t


class EqClasses:
    """Flattened eqVec() (EquivalenceClassBuilder.hpp:165-181): CSR of (label tids, weights, count)."""

    def __init__(self, off, tid, w, count, wq=None, bins=None, h1=None, h2=None):
        self.off, self.tid, self.w, self.count, self.wq, self.bins, self.h1, self.h2 = off, tid, w, count, wq, bins, h1, h2

    def table(self):
        return eq_table_from_arrays(self.off, self.tid, self.w, self.count, self.wq, self.bins, self.h1, self.h2)

    @staticmethod
    def alloc(E, L):
        return EqClasses(np.zeros(E + 1, np.uint64), np.zeros(L, np.uint32), np.zeros(L, np.float64), np.zeros(E, np.uint64),
                         np.zeros(L, np.uint64), np.zeros(L, np.uint32), np.zeros(E, np.uint64), np.zeros(E, np.uint64))

    def collapsed(self):
        """Collapse range-factorised labels to transcript sets (GZipWriter.cpp:89-114) -> {tuple(tids): count}."""
        d = {}
        for c in range(len(self.count)):
            key = tuple(int(x) for x in self.tid[self.off[c]:self.off[c + 1]])
            d[key] = d.get(key, 0) + int(self.count[c])
        return d


class QuantContext:
    """B1/B2: per-device mapping + online model + eq-class table."""

    def __init__(self, index, opts=None, device=0, max_batch_reads=1 << 20):
        self.index = index
        self.opts = opts if opts is not None else quant_opts()
        out = C.c_void_p()
        check(lib().sq_ctx_create(index.h, C.byref(self.opts), device, max_batch_reads, C.byref(out)), "sq_ctx_create")
        self.h = out

    def free(self):
        if self.h:
            lib().sq_ctx_free(self.h)
            self.h = None

    def map_batch(self, rb, fetch=True, aln_cap=None):
        st = capi.MapStats()
        if not fetch:
            check(lib().sq_map_batch(self.h, C.byref(rb), None, C.byref(st)), "sq_map_batch")
            return None, None, None, st.as_dict()
        n = rb.n
        cap = aln_cap or max(1024, 8 * n)
        while True:
            read_off = np.zeros(n + 1, np.uint64)
            aln = np.zeros(cap, ALN_DTYPE)
            mt = np.zeros(n, np.uint8)
            ab = capi.AlnBatch(n, _ptr(read_off, C.c_uint64), aln.ctypes.data_as(C.POINTER(capi.Aln)), cap, _ptr(mt, C.c_uint8))
            rc = lib().sq_map_batch(self.h, C.byref(rb), C.byref(ab), C.byref(st))
            if rc == -6 and cap < (1 << 31):
                cap *= 4
                continue
            check(rc, "sq_map_batch")
            break
        return read_off, aln[: int(read_off[-1])], mt, st.as_dict()

    def map_submit(self, rb, aln_cap=None, fetch=False):
        """Queue a batch on the next mapping lane (sq_map_submit); pair with map_wait(). Keeps `rb` alive."""
        if not hasattr(self, "_inflight"): self._inflight = []
        ent = dict(rb=rb, ab=None)
        if fetch:
            n = rb.n; cap = aln_cap or max(1024, 16 * n)
            ent.update(read_off=np.zeros(n + 1, np.uint64), aln=np.zeros(cap, ALN_DTYPE), mt=np.zeros(n, np.uint8))
            ent["ab"] = capi.AlnBatch(n, _ptr(ent["read_off"], C.c_uint64), ent["aln"].ctypes.data_as(C.POINTER(capi.Aln)), cap, _ptr(ent["mt"],
                C.c_uint8))
        check(lib().sq_map_submit(self.h, C.byref(rb), C.byref(ent["ab"]) if ent["ab"] is not None else None), "sq_map_submit")
        self._inflight.append(ent)

    def map_wait(self):
        """Oldest submitted batch (sq_map_wait) -> (read_off, aln, map_type, stats); arrays are None unless fetch=True was submitted."""
        st = capi.MapStats()
        check(lib().sq_map_wait(self.h, None, C.byref(st)), "sq_map_wait")
        ent = self._inflight.pop(0)
        if ent["ab"] is None:
            return None, None, None, st.as_dict()
        return ent["read_off"], ent["aln"][: int(ent["read_off"][-1])], ent["mt"], st.as_dict()

    def map_fetch(self):
        """Alignments of the batch map_wait / map_batch returned last (sq_map_fetch: size query, then the copy)."""
        ab = capi.AlnBatch()
        check(lib().sq_map_fetch(self.h, C.byref(ab)), "sq_map_fetch")
        n, cap = int(ab.n), int(ab.aln_cap)
        read_off = np.zeros(n + 1, np.uint64); aln = np.zeros(max(cap, 1), ALN_DTYPE); mt = np.zeros(n, np.uint8)
        ab = capi.AlnBatch(n, _ptr(read_off, C.c_uint64), aln.ctypes.data_as(C.POINTER(capi.Aln)), max(cap, 1), _ptr(mt, C.c_uint8))
        check(lib().sq_map_fetch(self.h, C.byref(ab)), "sq_map_fetch")
        return read_off, aln[:cap], mt

    def tap(self, what, dtype):
        n = lib().sq_debug_tap(self.h, what, None, 0)
        if n < 0:
            check(int(n), "sq_debug_tap")
        buf = np.zeros(int(n), dtype)
        if n:
            lib().sq_debug_tap(self.h, what, buf.ctypes.data, int(n))
        return buf

    def reserve(self, max_classes=0, max_labels=0):
        check(lib().sq_ctx_reserve(self.h, int(max_classes), int(max_labels)), "sq_ctx_reserve")

    def reset(self):
        check(lib().sq_ctx_reset(self.h), "sq_ctx_reset")

    def set_profiling(self, on=True):
        check(lib().sq_ctx_set_profiling(self.h, int(on)), "sq_ctx_set_profiling")

    def stage_times(self, reset=False):
        n = lib().sq_ctx_num_stages()
        ms = (C.c_double * n)(); calls = (C.c_uint64 * n)()
        check(lib().sq_ctx_stage_times(self.h, ms, calls, int(reset)), "sq_ctx_stage_times")
        return {lib().sq_ctx_stage_name(i).decode(): (float(ms[i]), int(calls[i])) for i in range(n)}

    def eq_accumulate(self):
        check(lib().sq_eq_accumulate(self.h), "sq_eq_accumulate")

    def eq_finish(self):
        t = capi.EqTable()
        check(lib().sq_eq_finish(self.h, C.byref(t)), "sq_eq_finish(size)")
        eq = EqClasses.alloc(int(t.num_classes), int(t.num_labels))
        tt = eq.table()
        check(lib().sq_eq_finish(self.h, C.byref(tt)), "sq_eq_finish")
        return eq

    def em_optimize(self, eff_len, projected=None, opts=None, unique=None):
        """CollapsedEMOptimizer::optimize over the classes this context accumulated (sq_em_optimize with eq = NULL):
        the optimizer reads the canonical-order export that already sits in HBM."""
        o = opts or em_opts()
        txp = make_txp_in(eff_len, projected, unique)
        out = np.zeros(txp.num_txp)
        rep = capi.EmReport()
        check(lib().sq_em_optimize(self.h, None, C.byref(txp), C.byref(o), _ptr(out, C.c_double), C.byref(rep)), "sq_em_optimize")
        return out, dict(iters=rep.iters, converged=bool(rep.converged), max_rel_diff=rep.max_rel_diff, alpha_sum=rep.alpha_sum,
                         device_ms=rep.device_ms, ms_per_iter=rep.ms_per_iter, num_degenerate=rep.num_degenerate)

    def eq_export_device(self):
        """Device pointers of the canonical-order export (sq_eq_export_device): dict field -> (ptr, count, numpy dtype)."""
        t = capi.EqTable()
        check(lib().sq_eq_export_device(self.h, C.byref(t)), "sq_eq_export_device")
        E, L = int(t.num_classes), int(t.num_labels)
        def addr(p): return C.cast(p, C.c_void_p).value or 0
        return dict(E=E, L=L, off=(addr(t.off), E + 1 if E else 0, np.uint64), tid=(addr(t.tid), L, np.uint32), wq=(addr(t.wq), L, np.uint64),
            count=(addr(t.count), E, np.uint64),
                    bins=(addr(t.bins), L, np.uint32), h1=(addr(t.h1), E, np.uint64), h2=(addr(t.h2), E, np.uint64))

    def eq_merge_device(self, E, L, ptrs):
        """sq_eq_merge_device: ptrs = dict field -> device address (off, tid, wq, count, bins, h1, h2) on this ctx's GPU."""
        t = capi.EqTable(); t.num_classes = E; t.num_labels = L
        t.off = C.cast(ptrs["off"], C.POINTER(C.c_uint64)); t.tid = C.cast(ptrs["tid"], C.POINTER(C.c_uint32)); t.wq = C.cast(ptrs["wq"],
            C.POINTER(C.c_uint64))
        t.count = C.cast(ptrs["count"], C.POINTER(C.c_uint64)); t.bins = C.cast(ptrs["bins"], C.POINTER(C.c_uint32)); t.h1 = C.cast(ptrs["h1"],
            C.POINTER(C.c_uint64)); t.h2 = C.cast(ptrs["h2"], C.POINTER(C.c_uint64))
        check(lib().sq_eq_merge_device(self.h, C.byref(t)), "sq_eq_merge_device")

    def eq_merge(self, eq):
        t = eq.table()
        check(lib().sq_eq_merge(self.h, C.byref(t)), "sq_eq_merge")

    def summary(self):
        s = capi.ModelSummary()
        check(lib().sq_model_summary_get(self.h, C.byref(s)), "sq_model_summary_get")
        return dict(num_observed=int(s.num_observed), num_assigned=int(s.num_assigned), num_mapped_ub=int(s.num_mapped_ub),
            burned_in=bool(s.burned_in), num_compatible=int(s.num_compatible), lib_format_id=int(s.lib_format_id), lib_detected=int(s.lib_detected))

    def drop_counts(self):
        """SPEC MG, end of the shared burn-in prefix on a rank other than 0 (sq_model_drop_counts)."""
        check(lib().sq_model_drop_counts(self.h), "sq_model_drop_counts")

    def lib_counts(self):
        out = np.zeros(64, np.uint64)
        check(lib().sq_model_fetch_lib_counts(self.h, _ptr(out, C.c_uint64)), "sq_model_fetch_lib_counts")
        return out

    def model(self):
        M = self.index.num_refs
        lm, uq, tc, le = np.zeros(M), np.zeros(M, np.uint64), np.zeros(M, np.uint64), np.zeros(M)
        check(lib().sq_model_fetch(self.h, _ptr(lm, C.c_double), _ptr(uq, C.c_uint64), _ptr(tc, C.c_uint64), _ptr(le, C.c_double)), "sq_model_fetch")
        return lm, uq, tc, le

    def gc_observed(self):
        """observedGCMass (needs quant_opts(gc_bias=1)): [3, 25] linear-space masses."""
        g = np.zeros(75)
        check(lib().sq_model_fetch_gc_observed(self.h, g.ctypes.data), "sq_model_fetch_gc_observed")
        return g.reshape(3, 25)

    def seq_observed(self):
        """Observed read-start context counts (needs quant_opts(seq_bias=1)): (fw[576], rc[576], fragments sampled)."""
        fw = np.zeros(576, np.uint64); rc = np.zeros(576, np.uint64); n = C.c_uint64()
        check(lib().sq_model_fetch_seq_observed(self.h, fw.ctypes.data, rc.ctypes.data, C.byref(n)), "sq_model_fetch_seq_observed")
        return fw, rc, int(n.value)

    def pos_observed(self):
        """Observed read-start masses by length class (needs quant_opts(pos_bias=1)): [2 (5', 3'), 5, 20], linear, without the models' initial mass."""
        g = np.zeros(200)
        check(lib().sq_model_fetch_pos_observed(self.h, g.ctypes.data), "sq_model_fetch_pos_observed")
        return g.reshape(2, 5, 20)

    def em_optimize_bias(self, eff_len, projected, log_pmf, gc_obs=None, seq=None, pos_obs=None, threads=8, opts=None, eq=None):
        """CollapsedEMOptimizer::optimize with any combination of --gcBias / --seqBias / --posBias: sq_em_optimize_bias with sq_bias_eff_lengths as the callback."""
        o = opts or em_opts(); txp = make_txp_in(eff_len, projected); M = txp.num_txp
        out = np.zeros(M); eff_out = np.zeros(M); rep = capi.EmReport(); brep = capi.BiasReport()
        lp = np.ascontiguousarray(log_pmf, np.float64); bm, keep = _bias_models(gc_obs, seq, pos_obs, threads)
        idx_h = self.index.h
        def cb(alphas, eff_in, eff_o, m, user):
            return lib().sq_bias_eff_lengths(idx_h, C.byref(bm), lp.ctypes.data, m, C.cast(alphas, C.c_void_p), C.cast(eff_in, C.c_void_p), C.cast(eff_o, C.c_void_p), None, None, C.byref(brep))
        cbf = capi.EFFLEN_CB(cb)
        t = eq.table() if eq is not None else None
        check(lib().sq_em_optimize_bias(self.h, C.byref(t) if t is not None else None, C.byref(txp), C.byref(o), cbf, None, _ptr(out, C.c_double),
            _ptr(eff_out, C.c_double), C.byref(rep)), "sq_em_optimize_bias")
        return out, eff_out, dict(iters=rep.iters, converged=bool(rep.converged), num_degenerate=rep.num_degenerate, num_processed=brep.num_processed)

    def em_optimize_seq(self, eff_len, projected, seq_fw, seq_rc, log_pmf, gc_obs=None, opts=None, eq=None):
        """CollapsedEMOptimizer::optimize with --seqBias [and --gcBias]: sq_em_optimize_bias with sq_bias_seq_eff_lengths as the callback."""
        o = opts or em_opts(); txp = make_txp_in(eff_len, projected); M = txp.num_txp
        out = np.zeros(M); eff_out = np.zeros(M); rep = capi.EmReport(); brep = capi.BiasReport()
        fw = np.ascontiguousarray(seq_fw, np.uint64); rc = np.ascontiguousarray(seq_rc, np.uint64); lp = np.ascontiguousarray(log_pmf, np.float64)
        g = np.ascontiguousarray(gc_obs, np.float64).reshape(-1) if gc_obs is not None else None
        idx_h = self.index.h
        def cb(alphas, eff_in, eff_o, m, user):
            return lib().sq_bias_seq_eff_lengths(idx_h, 1 if g is not None else 0, g.ctypes.data if g is not None else None, fw.ctypes.data, rc.ctypes.data, lp.ctypes.data, m,
                C.cast(alphas, C.c_void_p), C.cast(eff_in, C.c_void_p), C.cast(eff_o, C.c_void_p), None, C.byref(brep))
        cbf = capi.EFFLEN_CB(cb)
        t = eq.table() if eq is not None else None
        check(lib().sq_em_optimize_bias(self.h, C.byref(t) if t is not None else None, C.byref(txp), C.byref(o), cbf, None, _ptr(out, C.c_double),
            _ptr(eff_out, C.c_double), C.byref(rep)), "sq_em_optimize_bias")
        return out, eff_out, dict(iters=rep.iters, converged=bool(rep.converged), num_degenerate=rep.num_degenerate, num_processed=brep.num_processed)

    def em_optimize_gc(self, eff_len, projected, gc_obs, log_pmf, opts=None, eq=None):
        """CollapsedEMOptimizer::optimize with --gcBias: the effective lengths are re-derived from the GC models at iteration 11
        (sq_em_optimize_bias with sq_bias_gc_eff_lengths as the callback).  Returns (alphas, eff_lens, report)."""
        o = opts or em_opts(); txp = make_txp_in(eff_len, projected); M = txp.num_txp
        out = np.zeros(M); eff_out = np.zeros(M); rep = capi.EmReport(); brep = capi.BiasReport()
        g = np.ascontiguousarray(gc_obs, np.float64).reshape(-1); lp = np.ascontiguousarray(log_pmf, np.float64)
        idx_h = self.index.h
        def cb(alphas, eff_in, eff_o, m, user):
            return lib().sq_bias_gc_eff_lengths(idx_h, g.ctypes.data, lp.ctypes.data, m, C.cast(alphas, C.c_void_p), C.cast(eff_in, C.c_void_p),
                C.cast(eff_o, C.c_void_p), C.byref(brep))
        cbf = capi.EFFLEN_CB(cb)
        t = eq.table() if eq is not None else None
        check(lib().sq_em_optimize_bias(self.h, C.byref(t) if t is not None else None, C.byref(txp), C.byref(o), cbf, None, _ptr(out, C.c_double),
            _ptr(eff_out, C.c_double), C.byref(rep)), "sq_em_optimize_bias")
        return out, eff_out, dict(iters=rep.iters, converged=bool(rep.converged), num_degenerate=rep.num_degenerate, num_processed=brep.num_processed,
            fld_low=brep.fld_low, fld_high=brep.fld_high, gc_bias=np.array(brep.gc_bias_row0))

    def fld(self):
        f = np.zeros(1001)
        check(lib().sq_model_fetch_fld(self.h, _ptr(f, C.c_double)), "sq_model_fetch_fld")
        return f


def make_txp_in(eff_len, projected=None, unique=None):
    t = capi.TxpIn()
    eff_len = np.ascontiguousarray(eff_len, np.float64)
    t.num_txp = len(eff_len)
    t.eff_len = _ptr(eff_len, C.c_double)
    keep = [eff_len]
    if projected is not None:
        projected = np.ascontiguousarray(projected, np.float64); t.projected_counts = _ptr(projected, C.c_double); keep.append(projected)
    if unique is not None:
        unique = np.ascontiguousarray(unique, np.uint64); t.unique_count = _ptr(unique, C.c_uint64); keep.append(unique)
    t._keep = keep
    return t


def em_optimize(eq, eff_len, projected=None, opts=None, device=0, unique=None):
    """CollapsedEMOptimizer::optimize on the GPU. Returns (alphas, report dict)."""
    o = opts or em_opts()
    t = eq.table()
    txp = make_txp_in(eff_len, projected, unique)
    out = np.zeros(txp.num_txp)
    rep = capi.EmReport()
    check(lib().sq_em_optimize_dev(device, C.byref(t), C.byref(txp), C.byref(o), _ptr(out, C.c_double), C.byref(rep)), "sq_em_optimize_dev")
    return out, dict(iters=rep.iters, converged=bool(rep.converged), max_rel_diff=rep.max_rel_diff, alpha_sum=rep.alpha_sum,
                     device_ms=rep.device_ms, ms_per_iter=rep.ms_per_iter, num_degenerate=rep.num_degenerate)


def em_steps(eq, eff_len, alpha_in, iters, opts=None, device=0):
    o = opts or em_opts()
    t = eq.table()
    txp = make_txp_in(eff_len)
    a = np.ascontiguousarray(alpha_in, np.float64)
    out = np.zeros(txp.num_txp)
    rep = capi.EmReport()
    check(lib().sq_em_steps_dev(device, C.byref(t), C.byref(txp), C.byref(o), _ptr(a, C.c_double), iters, _ptr(out, C.c_double), C.byref(rep)),
        "sq_em_steps_dev")
    return out, dict(iters=rep.iters, device_ms=rep.device_ms, ms_per_iter=rep.ms_per_iter)


def bias_gc_eff_lengths(index, gc_obs, log_pmf, alphas, eff_in):
    """salmon::utils::updateEffectiveLengths (gcBias) -> (eff_out, report dict); the index must be on a device."""
    g = np.ascontiguousarray(gc_obs, np.float64).reshape(-1); lp = np.ascontiguousarray(log_pmf, np.float64)
    a = np.ascontiguousarray(alphas, np.float64); e = np.ascontiguousarray(eff_in, np.float64); out = np.zeros(len(a)); rep = capi.BiasReport()
    check(lib().sq_bias_gc_eff_lengths(index.h, g.ctypes.data, lp.ctypes.data, len(a), a.ctypes.data, e.ctypes.data, out.ctypes.data, C.byref(rep)),
        "sq_bias_gc_eff_lengths")
    return out, dict(num_processed=rep.num_processed, fld_low=rep.fld_low, fld_high=rep.fld_high, gc_bias=np.array(rep.gc_bias_row0))


def bias_seq_eff_lengths(index, seq_fw, seq_rc, log_pmf, alphas, eff_in, gc_obs=None):
    """updateEffectiveLengths with --seqBias [+ --gcBias] -> (eff_out, models[4, 576], report); the index must be on a device."""
    fw = np.ascontiguousarray(seq_fw, np.uint64); rc = np.ascontiguousarray(seq_rc, np.uint64); lp = np.ascontiguousarray(log_pmf, np.float64)
    a = np.ascontiguousarray(alphas, np.float64); e = np.ascontiguousarray(eff_in, np.float64); out = np.zeros(len(a)); models = np.zeros((4, 576)); rep = capi.BiasReport()
    g = np.ascontiguousarray(gc_obs, np.float64).reshape(-1) if gc_obs is not None else None
    check(lib().sq_bias_seq_eff_lengths(index.h, 1 if g is not None else 0, g.ctypes.data if g is not None else None, fw.ctypes.data, rc.ctypes.data, lp.ctypes.data, len(a),
        a.ctypes.data, e.ctypes.data, out.ctypes.data, models.ctypes.data, C.byref(rep)), "sq_bias_seq_eff_lengths")
    return out, models, dict(num_processed=rep.num_processed, fld_low=rep.fld_low, fld_high=rep.fld_high, gc_bias=np.array(rep.gc_bias_row0))


def _bias_models(gc_obs, seq, pos_obs, threads):
    keep = [np.ascontiguousarray(gc_obs, np.float64).reshape(-1) if gc_obs is not None else None,
            np.ascontiguousarray(seq[0], np.uint64) if seq is not None else None, np.ascontiguousarray(seq[1], np.uint64) if seq is not None else None,
            np.ascontiguousarray(pos_obs, np.float64).reshape(-1) if pos_obs is not None else None]
    d = [k.ctypes.data if k is not None else None for k in keep]
    bm = capi.BiasModels(d[0], d[1], d[2], d[3], threads, 0); bm._keep = keep
    return bm, keep


def bias_eff_lengths(index, log_pmf, alphas, eff_in, gc_obs=None, seq=None, pos_obs=None, threads=8):
    """updateEffectiveLengths with any combination of --gcBias / --seqBias (seq = (fw, rc) counts) / --posBias (pos_obs [2, 5, 20])
    -> (eff_out, seq models [4, 576], positional models [4, 5, 20], report); the index must be on a device."""
    lp = np.ascontiguousarray(log_pmf, np.float64); a = np.ascontiguousarray(alphas, np.float64); e = np.ascontiguousarray(eff_in, np.float64)
    out = np.zeros(len(a)); sm = np.zeros((4, 576)); pm = np.zeros((4, 5, 20)); rep = capi.BiasReport(); bm, keep = _bias_models(gc_obs, seq, pos_obs, threads)
    check(lib().sq_bias_eff_lengths(index.h, C.byref(bm), lp.ctypes.data, len(a), a.ctypes.data, e.ctypes.data, out.ctypes.data, sm.ctypes.data, pm.ctypes.data, C.byref(rep)), "sq_bias_eff_lengths")
    return out, sm, pm, dict(num_processed=rep.num_processed, fld_low=rep.fld_low, fld_high=rep.fld_high, gc_bias=np.array(rep.gc_bias_row0))


def length_classes(index):
    """Transcript::lengthClassIndex: (quantiles, class of every reference)."""
    q = np.zeros(5, np.uint32); cls = np.zeros(index.num_refs, np.uint8)
    n = lib().sq_index_length_classes(index.h, q.ctypes.data, cls.ctypes.data)
    if n < 0: raise RuntimeError("sq_index_length_classes failed")
    return q[:n], cls


def normalize_alphas(eq, log_mass, uniq, total):
    """salmon::utils::normalizeAlphas (host, once per run) -> projectedCounts."""
    M = len(log_mass)
    out = np.zeros(M)
    t = eq.table()
    lm = np.ascontiguousarray(log_mass, np.float64); uq = np.ascontiguousarray(uniq, np.uint64); tc = np.ascontiguousarray(total, np.uint64)
    check(lib().sq_normalize_alphas(M, C.byref(t), _ptr(lm, C.c_double), _ptr(uq, C.c_uint64), _ptr(tc, C.c_uint64), _ptr(out, C.c_double)),
        "sq_normalize_alphas")
    return out


def write_quant_sf(path, index, eff_len, num_reads, num_mapped=0.0, sig_digits=None):
    e = np.ascontiguousarray(eff_len, np.float64); r = np.ascontiguousarray(num_reads, np.float64)
    if sig_digits is not None:   # --sigDigits
        check(lib().sq_write_quant_sf_digits(path.encode(), index.h, _ptr(e, C.c_double), _ptr(r, C.c_double), float(num_mapped), int(sig_digits)),
              "sq_write_quant_sf_digits")
        return
    check(lib().sq_write_quant_sf(path.encode(), index.h, _ptr(e, C.c_double), _ptr(r, C.c_double), float(num_mapped)), "sq_write_quant_sf")


def write_eq_classes(path, index, eq, with_weights=False):
    t = eq.table()
    check(lib().sq_write_eq_classes(path.encode(), index.h, C.byref(t), int(with_weights)), "sq_write_eq_classes")


def write_ambig_info(path, M, eq):
    t = eq.table()
    check(lib().sq_write_ambig_info(path.encode(), M, C.byref(t)), "sq_write_ambig_info")


def read_eq_classes(path):
    """salmon::utils::readEquivCounts: (names, eff_lens, EqClasses) from an eq_classes.txt[.gz] written with weights."""
    h = C.c_void_p()
    check(lib().sq_eq_file_read(path.encode(), C.byref(h)), "sq_eq_file_read")
    try:
        M = lib().sq_eq_file_num_txp(h)
        names = [lib().sq_eq_file_name(h, i).decode() for i in range(M)]
        eff = np.ctypeslib.as_array(lib().sq_eq_file_eff_lens(h), shape=(M,)).copy()
        t = capi.EqTable(); check(lib().sq_eq_file_table(h, C.byref(t)), "sq_eq_file_table")
        E, L = int(t.num_classes), int(t.num_labels)
        off = np.ctypeslib.as_array(t.off, shape=(E + 1,)).copy()
        tid = np.ctypeslib.as_array(t.tid, shape=(L,)).copy() if L else np.zeros(0, np.uint32)
        w = np.ctypeslib.as_array(t.w, shape=(L,)).copy() if L else np.zeros(0)
        cnt = np.ctypeslib.as_array(t.count, shape=(E,)).copy() if E else np.zeros(0, np.uint64)
        return names, eff, EqClasses(off, tid, w, cnt)
    finally:
        lib().sq_eq_file_free(h)


def write_bootstraps(aux_dir, names, rows):
    """aux_info/bootstrap/{names.tsv.gz, bootstraps.gz} (GZipWriter::writeBootstrap); returns the count written."""
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names]); h = C.c_void_p()
    check(lib().sq_boot_writer_open(aux_dir.encode(), len(names), arr, C.byref(h)), "sq_boot_writer_open")
    for r in rows:
        r = np.ascontiguousarray(r, np.float64)
        check(lib().sq_boot_writer_append(h, _ptr(r, C.c_double), len(r)), "sq_boot_writer_append")
    return int(lib().sq_boot_writer_close(h))


def _collect(n_txp):
    rows = []

    def cb(ptr, m, user):
        rows.append(np.ctypeslib.as_array(ptr, shape=(m,)).copy())
        return 0
    return rows, capi.REPLICATE_CB(cb)


def bootstrap(eq, eff_len, num_bootstraps, seed, num_mapped, opts=None, device=0):
    """CollapsedEMOptimizer::gatherBootstraps on the GPU -> array [B, M] (what writeBootstrap receives)."""
    o = opts or em_opts(); t = eq.table(); txp = make_txp_in(eff_len)
    rows, cb = _collect(txp.num_txp)
    check(lib().sq_bootstrap_dev(device, C.byref(t), C.byref(txp), C.byref(o), num_bootstraps, seed, num_mapped, cb, None), "sq_bootstrap_dev")
    return np.array(rows)


def bootstrap_range(eq, eff_len, num_bootstraps, first, count, seed, num_mapped, opts=None, device=0):
    """Replicates [first, first + count) of num_bootstraps (sq_bootstrap_range_dev): what a rank of a multi-GPU job computes."""
    o = opts or em_opts(); t = eq.table(); txp = make_txp_in(eff_len)
    rows, cb = _collect(txp.num_txp)
    check(lib().sq_bootstrap_range_dev(device, C.byref(t), C.byref(txp), C.byref(o), num_bootstraps, first, count, seed, num_mapped, cb, None),
        "sq_bootstrap_range_dev")
    return np.array(rows)


def gibbs_range(eq, eff_len, alpha_init, num_samples, first, count, seed, num_mapped, gopts=None, device=0, report=False):
    """Samples [first, first + count) of num_samples (sq_gibbs_range_dev); report=True: also the device time of the sampling rounds."""
    g = gopts or gibbs_opts(); t = eq.table(); txp = make_txp_in(eff_len)
    a = np.ascontiguousarray(alpha_init, np.float64)
    rows, cb = _collect(txp.num_txp)
    rep = capi.GibbsReport()
    check(lib().sq_gibbs_range_report_dev(device, C.byref(t), C.byref(txp), C.byref(g), _ptr(a, C.c_double), num_samples, first, count, seed, num_mapped,
        cb, None, C.byref(rep)), "sq_gibbs_range_dev")
    if report:
        return np.array(rows), dict(rounds=int(rep.rounds), device_ms=rep.device_ms, ms_per_round=rep.ms_per_round, draws_per_round=int(rep.draws_per_round),
                                    items=[int(x) for x in rep.items])
    return np.array(rows)


class Dist:
    """The multi-GPU seam (sq_dist_*): one process per GPU over RCCL.  `make_id()` on rank 0, hand the 128 bytes to every rank
    (torch.distributed, MPI, a file), then Dist(id, rank, world, device) collectively."""

    @staticmethod
    def make_id():
        buf = (C.c_uint8 * 128)()
        check(lib().sq_dist_make_id(buf), "sq_dist_make_id")
        return bytes(buf)

    def __init__(self, id128, rank, world, device):
        buf = (C.c_uint8 * 128).from_buffer_copy(id128)
        h = C.c_void_p()
        check(lib().sq_dist_init(buf, rank, world, device, C.byref(h)), "sq_dist_init")
        self.h = h; self.rank = rank; self.world = world

    def free(self):
        if self.h:
            lib().sq_dist_free(self.h); self.h = None

    def merge_eq(self, ctx):
        check(lib().sq_dist_merge_eq(self.h, ctx.h), "sq_dist_merge_eq")

    def merge_eq_loopback(self, ctxs):
        """sq_dist_merge_eq_loopback: the contexts stand for the ranks of a len(ctxs)-rank job on this one device."""
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        check(lib().sq_dist_merge_eq_loopback(self.h, arr, len(ctxs)), "sq_dist_merge_eq_loopback")

    def allgather(self, a):
        """sq_dist_allgather of a host array: [world, len(a)]."""
        a = np.ascontiguousarray(a); out = np.zeros((self.world,) + a.shape, a.dtype)
        check(lib().sq_dist_allgather(self.h, a.ctypes.data, a.nbytes, out.ctypes.data), "sq_dist_allgather")
        return out

    def bcast(self, a, root=0):
        a = np.ascontiguousarray(a).copy()
        check(lib().sq_dist_bcast(self.h, a.ctypes.data, a.nbytes, root), "sq_dist_bcast")
        return a

    def reduce_model(self, log_mass, uniq, total, log_eff_len):
        lm = np.ascontiguousarray(log_mass, np.float64).copy(); uq = np.ascontiguousarray(uniq, np.uint64).copy()
        tc = np.ascontiguousarray(total, np.uint64).copy(); le = np.ascontiguousarray(log_eff_len, np.float64).copy()
        check(lib().sq_dist_reduce_model(self.h, len(lm), _ptr(lm, C.c_double), _ptr(uq, C.c_uint64), _ptr(tc, C.c_uint64), _ptr(le, C.c_double)),
            "sq_dist_reduce_model")
        return lm, uq, tc, le

    def allreduce_u64(self, a):
        a = np.ascontiguousarray(a, np.uint64).copy()
        check(lib().sq_dist_allreduce_u64(self.h, a.ctypes.data, a.size), "sq_dist_allreduce_u64")
        return a

    def barrier(self):
        check(lib().sq_dist_barrier(self.h), "sq_dist_barrier")

    def share(self, total, unit=1):
        f, n = C.c_uint32(), C.c_uint32()
        lib().sq_dist_share(self.h, total, unit, C.byref(f), C.byref(n))
        return f.value, n.value


def gibbs_opts(thinning_factor=16, no_gamma_draw=0, use_vbem=1, per_transcript_prior=1, vb_prior=1e-2):
    return capi.GibbsOpts(thinning_factor, no_gamma_draw, use_vbem, per_transcript_prior, 0, vb_prior)


def gibbs(eq, eff_len, alpha_init, num_samples, seed, num_mapped, gopts=None, device=0):
    """CollapsedGibbsSampler::sample on the GPU -> array [S, M]."""
    g = gopts or gibbs_opts(); t = eq.table(); txp = make_txp_in(eff_len)
    a = np.ascontiguousarray(alpha_init, np.float64)
    rows, cb = _collect(txp.num_txp)
    check(lib().sq_gibbs_dev(device, C.byref(t), C.byref(txp), C.byref(g), _ptr(a, C.c_double), num_samples, seed, num_mapped, cb, None),
        "sq_gibbs_dev")
    return np.array(rows)


def debug_infix_align(queries, windows, ks, device=0):
    """Device infix aligner (orphan recovery, row a5) on (query, window) ASCII pairs -> int32[n, 4] = found, distance, start, end."""
    n = len(queries)
    qo = np.zeros(n + 1, np.uint64); wo = np.zeros(n + 1, np.uint64)
    qo[1:] = np.cumsum([len(q) for q in queries]); wo[1:] = np.cumsum([len(w) for w in windows])
    qb = np.frombuffer(b"".join(queries) + b"\0", np.uint8); wb = np.frombuffer(b"".join(windows) + b"\0", np.uint8)
    k = np.ascontiguousarray(ks, np.int32); out = np.zeros((n, 4), np.int32)
    check(lib().sq_debug_infix_align(device, n, qb.ctypes.data, qo.ctypes.data, wb.ctypes.data, wo.ctypes.data, k.ctypes.data, out.ctypes.data),
        "sq_debug_infix_align")
    return out
