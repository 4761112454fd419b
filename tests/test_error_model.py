"""[r5] The CIGAR-based alignment error model of alignment-based input (`salmon quant -a` without --noErrorModel; src/alignment/AlignmentModel.cpp,
SalmonQuantifyAlignments.cpp:513-523, :860-864).  SAM / BAM files whose records carry real CIGARs (matches, mismatches, insertions, deletions, soft clips)
and sequences are written from random alignments; the reader hands the reads on with their alignments (sq_sam_keep_reads / sq_sam_reads); the checker's
online stage weighs every alignment by the model's log-likelihood and learns the matrices during burn-in (oracle.cpp, pinned to the compiled reference by
tests/test_alnmodel_pin.py); the HIP stage (k_err_like / k_err_count / k_err_apply, hip/online.hip) gives the same class table, the same online model and the
same matrices, bit for bit."""
import ctypes as C, gzip, os, subprocess
import numpy as np
import pytest
from salmon_amd import api, capi
import orc
from test_alignment_mode import sam_to_bam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASES = "ACGT"


def _edit(rng, T, pos, L):
    """a read of L bases drawn from T at pos with substitutions, short insertions and deletions and, sometimes, soft-clipped ends: (CIGAR string, read in reference orientation)"""
    ops = []; seq = []; t = pos; left = L
    def push(op, n):
        if ops and ops[-1][0] == op: ops[-1][1] += n
        else: ops.append([op, n])
    if rng.random() < 0.15: n = int(rng.integers(1, 6)); push("S", n); seq += list(rng.integers(0, 4, n)); left -= n
    tail = int(rng.integers(1, 6)) if rng.random() < 0.15 else 0; left -= tail
    while left > 0 and t < len(T):
        u = rng.random()
        if u < 0.93 or not ops or ops[-1][0] != "M":
            n = int(min(left, len(T) - t, rng.integers(5, 40)))
            for j in range(n): seq.append(int(T[t + j]) if rng.random() > 0.02 else int(rng.integers(0, 4)))
            push("M", n); t += n; left -= n
        elif u < 0.965: n = int(min(left, rng.integers(1, 3))); push("I", n); seq += list(rng.integers(0, 4, n)); left -= n
        else: n = int(rng.integers(1, 3)); push("D", n); t += n
    if ops and ops[-1][0] == "D": ops.pop()
    tail += max(0, left)
    if tail: push("S", tail); seq += list(rng.integers(0, 4, tail))
    return "".join("%d%s" % (n, op) for op, n in ops), "".join(BASES[b] for b in seq)


def write_sam_with_reads(path, names, seqs, read_off, aln, rng, star_secondary=False, pg="test"):
    """Every alignment as SAM records with CIGAR and SEQ made from the transcript it lies on.  Returns, per alignment, the two records in FILE order as
    (pos, cigar string, sequence) or None — what the reader must hand back, sorted into its left / right slots."""
    recs = []
    with open(path, "wt") as f:
        f.write("@HD\tVN:1.6\tSO:unsorted\tGO:query\n")
        for n, s in zip(names, seqs): f.write("@SQ\tSN:%s\tLN:%d\n" % (n, len(s)))
        f.write("@PG\tID:%s\n" % pg)
        for fr in range(len(read_off) - 1):
            for k, a in enumerate(aln[int(read_off[fr]):int(read_off[fr + 1])]):
                T = seqs[int(a["tid"])]; t = names[int(a["tid"])]; tag = "\tAS:i:%d" % a["score"]; tag2 = "\tAS:i:%d" % a["mate_score"]
                star = star_secondary and k > 0
                c1, s1 = _edit(rng, T, int(a["pos"]), int(a["read_len"]))
                if a["mate_status"] == 3:
                    c2, s2 = _edit(rng, T, int(a["mate_pos"]), int(a["mate_len"]))
                    fl1 = 1 | 2 | 64 | (0 if a["fwd"] else 16) | (0 if a["mate_fwd"] else 32); fl2 = 1 | 2 | 128 | (0 if a["mate_fwd"] else 16) | (0 if a["fwd"] else 32)
                    f.write("r%d/1\t%d\t%s\t%d\t255\t%s\t=\t%d\t0\t%s\t*%s\n" % (fr, fl1, t, a["pos"] + 1, c1, a["mate_pos"] + 1, s1, tag))
                    f.write("r%d/2\t%d\t%s\t%d\t255\t%s\t=\t%d\t0\t%s\t*%s\n" % (fr, fl2, t, a["mate_pos"] + 1, c2, a["pos"] + 1, s2, tag2))
                    recs.append(((int(a["pos"]), c1, s1), (int(a["mate_pos"]), c2, s2)))
                elif a["mate_status"] in (1, 2):
                    fl = 1 | 8 | (64 if a["mate_status"] == 1 else 128) | (0 if a["fwd"] else 16)
                    f.write("r%d/%d\t%d\t%s\t%d\t255\t%s\t*\t0\t0\t%s\t*%s\n" % (fr, 1 if a["mate_status"] == 1 else 2, fl, t, a["pos"] + 1, c1, s1, tag))
                    recs.append(((int(a["pos"]), c1, s1), None))
                else:
                    f.write("r%d\t%d\t%s\t%d\t255\t%s\t*\t0\t0\t%s\t*%s\n" % (fr, 0 if a["fwd"] else 16, t, a["pos"] + 1, c1, "*" if star else s1, tag))
                    recs.append(((int(a["pos"]), c1, s1), None))
    return recs


def read_sam_with_reads(path, paired=True, max_frags=1 << 20):
    """(read_off, alignments, reads) per batch; reads = dict of numpy copies of sq_aln_reads' arrays"""
    L = capi.lib(); h = C.c_void_p(); capi.check(L.sq_sam_open(str(path).encode(), int(paired), C.byref(h)), "sq_sam_open"); capi.check(L.sq_sam_keep_reads(h, 1), "keep")
    out = []
    while True:
        ab = capi.AlnBatch(); capi.check(L.sq_sam_next(h, max_frags, 0, 1.0, C.byref(ab), None), "sq_sam_next")
        if ab.n == 0: break
        ro = np.ctypeslib.as_array(ab.read_off, shape=(ab.n + 1,)).copy(); na = int(ro[-1])
        a = np.frombuffer(C.string_at(ab.aln, na * api.ALN_DTYPE.itemsize), api.ALN_DTYPE).copy()
        rd = capi.AlnReads(); capi.check(L.sq_sam_reads(h, C.byref(rd)), "sq_sam_reads"); assert rd.num_alignments == na
        co = np.ctypeslib.as_array(rd.cig_off, shape=(2 * na + 1,)).copy(); so = np.ctypeslib.as_array(rd.seq_off, shape=(2 * na + 1,)).copy()
        r = dict(cig_off=co, seq_off=so, cigar=np.ctypeslib.as_array(rd.cigar, shape=(max(1, int(co[-1])),)).copy()[:int(co[-1])], seq=np.ctypeslib.as_array(rd.seq, shape=(max(1, int(so[-1])),)).copy()[:int(so[-1])],
                 pos=np.ctypeslib.as_array(rd.pos, shape=(2 * na,)).copy(), score=np.ctypeslib.as_array(rd.aligner_score, shape=(na,)).copy())
        out.append((ro, a, r))
    L.sq_sam_close(h)
    return out


def reads_struct(r, na):
    for k in ("cig_off", "seq_off"): r[k] = np.ascontiguousarray(r[k], np.uint64)
    r["cigar"] = np.ascontiguousarray(r["cigar"], np.uint32); r["seq"] = np.ascontiguousarray(r["seq"], np.uint8); r["pos"] = np.ascontiguousarray(r["pos"], np.int32); r["score"] = np.ascontiguousarray(r["score"], np.int32)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    return capi.AlnReads(na, p(r["cig_off"], C.c_uint64), p(r["cigar"], C.c_uint32), p(r["seq_off"], C.c_uint64), p(r["seq"], C.c_uint8), p(r["pos"], C.c_int32), p(r["score"], C.c_int32))


def _cig(s):
    out = []; num = 0
    for ch in s:
        if ch.isdigit(): num = num * 10 + int(ch)
        else: out.append((num << 4) | "MIDNSHP=X".index(ch)); num = 0
    return out


def _world(rng, n_txp=14, n_frag=600):
    names = ["tx%d" % i for i in range(n_txp)]; seqs = [np.array(rng.integers(0, 4, int(rng.integers(900, 1500))), np.uint8) for _ in range(n_txp)]
    ro = [0]; rows = []
    for f in range(n_frag):
        k = int(rng.integers(1, 4)); kind = rng.choice([3, 3, 3, 1, 2])
        for t in sorted(rng.choice(n_txp, k, replace=False)):
            a = np.zeros(1, api.ALN_DTYPE)[0]; TL = len(seqs[t])
            a["tid"] = t; a["pos"] = int(rng.integers(0, TL - 420)); a["fwd"] = int(rng.integers(0, 2)); a["read_len"] = int(rng.integers(60, 101)); a["score"] = int(rng.integers(-30, 1)); a["mate_status"] = kind
            if kind == 3: a["mate_fwd"] = 1 - a["fwd"]; a["mate_pos"] = a["pos"] + int(rng.choice([0, 0, int(rng.integers(-30, 300))])); a["mate_pos"] = max(0, int(a["mate_pos"])); a["mate_len"] = int(rng.integers(60, 101)); a["mate_score"] = int(rng.integers(-30, 1))
            rows.append(a)
        ro.append(len(rows))
    return names, seqs, np.array(ro, np.uint64), np.array(rows, api.ALN_DTYPE)


def test_reader_hands_the_reads_on_with_their_alignments(built, tmp_path):
    rng = np.random.default_rng(21); names, seqs, ro, aln = _world(rng, n_frag=250)
    recs = write_sam_with_reads(tmp_path / "e.sam", names, seqs, ro, aln, rng); sam_to_bam(tmp_path / "e.sam", tmp_path / "e.bam")
    for path in (tmp_path / "e.sam", tmp_path / "e.bam"):
        got = read_sam_with_reads(path, max_frags=97); ai = 0
        for ro_b, a_b, r in got:
            for j in range(len(a_b)):
                first, second = recs[ai]; ai += 1
                if second is None: slots = {0: first} if a_b[j]["mate_status"] != 2 else {1: first}               # left orphan -> left matrices, right orphan -> right
                else: fl = first[0] < second[0]; slots = {0: first if fl else second, 1: second if fl else first}   # the smaller position is left; on a tie the file's second record
                for k in (0, 1):
                    c = r["cigar"][int(r["cig_off"][2 * j + k]):int(r["cig_off"][2 * j + k + 1])]; s = r["seq"][int(r["seq_off"][2 * j + k]):int(r["seq_off"][2 * j + k + 1])]
                    if k not in slots: assert len(c) == 0 and len(s) == 0; continue
                    pos, cg, sq = slots[k]
                    assert int(r["pos"][2 * j + k]) == pos and list(c) == _cig(cg) and "".join(BASES[b] for b in s) == sq, (path, ai, k)
                assert r["score"][j] == 0                                                                         # not bowtie2: the update weight is 1
        assert ai == len(aln)
    # the AS tags weigh the updates only when the aligner is bowtie2; a secondary record without its sequence borrows the fragment's stored one
    se = aln.copy(); se["mate_status"] = 0; se["mate_pos"] = 0; se["mate_len"] = 0; se["mate_fwd"] = 0; se["mate_score"] = 0; se["fwd"] = 1
    recs = write_sam_with_reads(tmp_path / "s.sam", names, seqs, ro, se, rng, star_secondary=True, pg="bowtie2")
    got = read_sam_with_reads(tmp_path / "s.sam", paired=False); ai = 0
    for ro_b, a_b, r in got:
        for f in range(len(ro_b) - 1):
            a0, a1 = int(ro_b[f]), int(ro_b[f + 1]); first_seq = r["seq"][int(r["seq_off"][2 * a0]):int(r["seq_off"][2 * a0 + 1])]
            for j in range(a0, a1):
                s = r["seq"][int(r["seq_off"][2 * j]):int(r["seq_off"][2 * j + 1])]
                assert r["score"][j] == a_b[j]["score"]
                if j > a0: assert np.array_equal(s, first_seq)
                ai += 1
    assert ai == len(se)


def _checker_run(oidx, opts, batches):
    ost = orc.OrcState(oidx, opts); L = orc.lib(); L.orc_eq_accumulate_reads.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    for ro_b, a_b, r in batches:
        rs = reads_struct(r, len(a_b)); a = np.ascontiguousarray(a_b); rr = np.ascontiguousarray(ro_b, np.uint64)
        L.orc_eq_accumulate_reads(ost.h, len(rr) - 1, rr.ctypes.data, a.ctypes.data, C.byref(rs), len(rr) - 1)
    ost.finish(); return ost


def _err_model(ost, bins):
    L = orc.lib(); L.orc_state_err_model.restype = C.c_uint32; L.orc_state_err_model.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    cells = np.zeros(2 * bins * 82 * 82); rows = np.zeros(2 * bins * 82); assert L.orc_state_err_model(ost.h, cells.ctypes.data, rows.ctypes.data) == bins
    return cells, rows


def test_checker_learns_and_uses_the_error_model(built, tmp_path):
    """CPU only: with the model on the matrices leave their prior during burn-in, stay put afterwards, and the class weights differ from the run without it."""
    rng = np.random.default_rng(5); names, seqs, ro, aln = _world(rng, n_frag=1200)
    write_sam_with_reads(tmp_path / "e.sam", names, seqs, ro, aln, rng)
    idx = api.SalmonIndex.build_mem(names, ["".join(BASES[b] for b in s) for s in seqs], threads=2, keep_duplicates=True, no_clip=True)
    oidx = orc.OrcIndex(idx); batches = read_sam_with_reads(tmp_path / "e.sam", max_frags=400)
    kw = dict(mini_batch_size=100, num_pre_burnin_frags=150, num_burnin_frags=700, mini_batches_in_flight=3)
    on = _checker_run(oidx, api.quant_opts(error_model=1, num_error_bins=4, **kw), batches); off = _checker_run(oidx, api.quant_opts(**kw), [(a, b, c) for a, b, c in batches])
    cells, rows = _err_model(on, 4)
    assert (cells != 0).sum() > 100 and (rows != np.log(82.0)).sum() > 20                          # learned
    assert np.allclose(np.logaddexp.reduce(cells.reshape(-1, 82), axis=1), rows, rtol=0, atol=1e-9)   # a row sum is the sum of its cells (AtomicMatrix::increment keeps both)
    e1, e0 = on.eq_finish(), off.eq_finish()
    assert e1.count.sum() == e0.count.sum() and (len(e1.wq) != len(e0.wq) or not np.array_equal(e1.wq, e0.wq))   # the same fragments, weighed differently (other weights, other range-factorisation bins)


@pytest.mark.gpu
def test_hip_stage_equals_checker_with_the_error_model(built, tmp_path):
    rng = np.random.default_rng(8); names, seqs, ro, aln = _world(rng, n_txp=20, n_frag=6000)
    write_sam_with_reads(tmp_path / "e.sam", names, seqs, ro, aln, rng); sam_to_bam(tmp_path / "e.sam", tmp_path / "e.bam")
    idx = api.SalmonIndex.build_mem(names, ["".join(BASES[b] for b in s) for s in seqs], threads=2, keep_duplicates=True, no_clip=True)
    oidx = orc.OrcIndex(idx); idx.to_device(0); L = capi.lib()
    for path, kw in ((tmp_path / "e.bam", dict(mini_batch_size=200, num_pre_burnin_frags=300, num_burnin_frags=2500, mini_batches_in_flight=4)),
                     (tmp_path / "e.sam", dict(mini_batch_size=500, num_pre_burnin_frags=0, num_burnin_frags=100000, mini_batches_in_flight=1, num_error_bins=6))):
        opts = api.quant_opts(error_model=1, **({"num_error_bins": 3} | kw)); bins = opts.num_error_bins
        batches = read_sam_with_reads(path, max_frags=1700); ost = _checker_run(oidx, opts, batches)
        ctx = api.QuantContext(idx, opts, device=0, max_batch_reads=4096)
        for ro_b, a_b, r in batches:
            rs = reads_struct(r, len(a_b)); a = np.ascontiguousarray(a_b); rr = np.ascontiguousarray(ro_b, np.uint64)
            ab = capi.AlnBatch(len(rr) - 1, rr.ctypes.data_as(C.POINTER(C.c_uint64)), a.ctypes.data_as(C.POINTER(capi.Aln)), len(a), None)
            capi.check(L.sq_aln_inject_reads(ctx.h, C.byref(ab), C.byref(rs), len(rr) - 1), "sq_aln_inject_reads"); ctx.eq_accumulate()
        eq_g = ctx.eq_finish(); lm_g, uq_g, tc_g, le_g = ctx.model(); eq_c = ost.eq_finish(); lm_c, uq_c, tc_c, le_c, _ = ost.model()
        for f in ("off", "tid", "bins", "count", "wq"): assert np.array_equal(getattr(eq_g, f), getattr(eq_c, f)), (str(path), f)
        assert np.array_equal(lm_g, lm_c) and np.array_equal(uq_g, uq_c) and np.array_equal(tc_g, tc_c) and np.array_equal(le_g, le_c)
        cells = np.zeros(2 * bins * 82 * 82); rows = np.zeros(2 * bins * 82); b = C.c_uint32(0)
        capi.check(L.sq_model_fetch_error_model(ctx.h, cells.ctypes.data, rows.ctypes.data, C.byref(b)), "fetch"); cc, rc = _err_model(ost, bins)
        assert b.value == bins and np.array_equal(cells, cc) and np.array_equal(rows, rc) and (cells != 0).sum() > 200
        # a batch without its reads is refused while the model is on
        ro_b, a_b, r = batches[0]; a = np.ascontiguousarray(a_b); rr = np.ascontiguousarray(ro_b, np.uint64)
        ab = capi.AlnBatch(len(rr) - 1, rr.ctypes.data_as(C.POINTER(C.c_uint64)), a.ctypes.data_as(C.POINTER(capi.Aln)), len(a), None)
        capi.check(L.sq_aln_inject(ctx.h, C.byref(ab), len(rr) - 1), "inject")
        with pytest.raises(Exception, match="reads"): ctx.eq_accumulate(); ctx.eq_finish()
        ctx.free()


@pytest.mark.gpu
def test_cli_alignment_mode_runs_the_error_model_by_default(built, tmp_path):
    """`salmon-hip quant -t T -a BAM` learns and applies the error model; --noErrorModel switches it off; both account for every fragment, with different estimates.
    (12 000 fragments and -p 1: a mini-batch is 5 000 fragments and sees the model as it was before it, so the second and third see what the first taught;
    with the default eight mini-batches in flight a job this small is over before the first snapshot is taken — measured on the checker with the same options.)"""
    import json
    rng = np.random.default_rng(3); names, seqs, ro, aln = _world(rng, n_txp=16, n_frag=12000)
    write_sam_with_reads(tmp_path / "e.sam", names, seqs, ro, aln, rng); sam_to_bam(tmp_path / "e.sam", tmp_path / "e.bam")
    open(tmp_path / "t.fa", "w").write("".join(">%s\n%s\n" % (n, "".join(BASES[b] for b in s)) for n, s in zip(names, seqs)))
    exe = os.path.join(ROOT, "salmon_amd", "bin", "salmon-hip"); res = {}
    for tag, extra in (("em", ["--numErrorBins", "4"]), ("noem", ["--noErrorModel"])):
        subprocess.check_call([exe, "quant", "-t", str(tmp_path / "t.fa"), "-l", "IU", "-a", str(tmp_path / "e.bam"), "-o", str(tmp_path / tag), "-q", "-p", "1", "--numPreAuxModelSamples", "200", "--numAuxModelSamples", "20000"] + extra)
        rows = [l.split("\t") for l in open(tmp_path / tag / "quant.sf").read().splitlines()[1:]]; res[tag] = np.array([float(x[4]) for x in rows])
        meta = json.load(open(tmp_path / tag / "aux_info" / "meta_info.json")); res[tag + "_n"] = meta["num_mapped"]
    # (a fragment whose every alignment is incompatible with -l IU is not assigned: fewer than 12 000 on this random input, the same number either way)
    assert 9000 < res["em_n"] == res["noem_n"] <= 12000 and np.abs(res["em"] - res["noem"]).max() > 1e-3
    assert abs(res["em"].sum() - res["noem"].sum()) < 1e-6 * res["em"].sum() and 0.9 * res["em_n"] < res["em"].sum() <= res["em_n"] + 1
