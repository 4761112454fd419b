"""[r4] Alignment-based mode (`salmon quant -a`, row f-4): the SAM record source (host/sam_reader.cpp, after the reference's BAMQueue / ReadPair) and
sq_aln_inject, which hands a batch of such alignments to the online model / equivalence-class stage in place of a mapped batch.  The SAM files here
are written from alignment arrays, so every field the reader derives can be checked against its source; the stage that follows is held to the checker
on the very records the reader produced."""
import ctypes as C, gzip, os, subprocess
import numpy as np
import pytest
import orc
from salmon_amd import api, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_sam(path, names, lens, read_off, aln, unaligned_every=0, with_as=True):
    """The alignments of every fragment as SAM records (pairs: two records, READ1 first; orphans and single-end reads: one), name-collated."""
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "wt") as f:
        f.write("@HD\tVN:1.6\tSO:unsorted\tGO:query\n")
        for n, l in zip(names, lens): f.write("@SQ\tSN:%s\tLN:%d\n" % (n, l))
        f.write("@PG\tID:test\n")
        for fr in range(len(read_off) - 1):
            if unaligned_every and fr % unaligned_every == 0:
                f.write("u%d\t77\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII\n" % fr); f.write("u%d\t141\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII\n" % fr)
            for a in aln[int(read_off[fr]):int(read_off[fr + 1])]:
                t = names[int(a["tid"])]; tag = ("\tAS:i:%d" % a["score"]) if with_as else ""; tag2 = ("\tAS:i:%d" % a["mate_score"]) if with_as else ""
                if a["mate_status"] == 3:
                    fl1 = 1 | 2 | 64 | (0 if a["fwd"] else 16) | (0 if a["mate_fwd"] else 32); fl2 = 1 | 2 | 128 | (0 if a["mate_fwd"] else 16) | (0 if a["fwd"] else 32)
                    f.write("r%d/1\t%d\t%s\t%d\t255\t%dM\t=\t%d\t0\t*\t*%s\n" % (fr, fl1, t, a["pos"] + 1, a["read_len"], a["mate_pos"] + 1, tag))
                    f.write("r%d/2\t%d\t%s\t%d\t255\t%dM\t=\t%d\t0\t*\t*%s\n" % (fr, fl2, t, a["mate_pos"] + 1, a["mate_len"], a["pos"] + 1, tag2))
                elif a["mate_status"] in (1, 2):
                    fl = 1 | 8 | (64 if a["mate_status"] == 1 else 128) | (0 if a["fwd"] else 16)
                    f.write("r%d/%d\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t*\t*%s\n" % (fr, 1 if a["mate_status"] == 1 else 2, fl, t, a["pos"] + 1, a["read_len"], tag))
                else:
                    f.write("r%d\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t*%s\n" % (fr, 0 if a["fwd"] else 16, t, a["pos"] + 1, a["read_len"], "A" * int(a["read_len"]), tag))


def sam_to_bam(sam_path, bam_path, block=40000):
    """The same records as BAM (SAM spec 4.2), BGZF-compressed: what an aligner piped through `samtools view -b` leaves.  AS goes out in the narrowest
    integer type, as samtools writes it; other tags (a string, an array) stand in front of it."""
    import struct, zlib
    op = gzip.open if str(sam_path).endswith(".gz") else open
    text = ""; refs = []; recs = []
    with op(sam_path, "rt") as f:
        for line in f:
            if line.startswith("@"):
                text += line
                if line.startswith("@SQ"): d = dict(x.split(":", 1) for x in line.rstrip("\n").split("\t")[1:]); refs.append((d["SN"], int(d["LN"])))
            else: recs.append(line.rstrip("\n").split("\t"))
    rid = {n: i for i, (n, _) in enumerate(refs)}
    out = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs)))
    for n, l in refs: out += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    ops = "MIDNSHP=X"
    for r in recs:
        qn, flag, rn, pos, mapq, cig, rnext, pnext, tlen, seq, qual = r[:11]; tags = r[11:]
        ref = rid.get(rn, -1); mref = ref if rnext == "=" else rid.get(rnext, -1)
        c = []; num = ""
        for ch in ("" if cig == "*" else cig):
            if ch.isdigit(): num += ch
            else: c.append((int(num) << 4) | ops.index(ch)); num = ""
        l_seq = 0 if seq == "*" else len(seq)
        sq4 = bytearray((l_seq + 1) // 2)
        for i, ch in enumerate("" if seq == "*" else seq): sq4[i // 2] |= "=ACMGRSVTWYHKDBN".index(ch) << (4 if i % 2 == 0 else 0)
        aux = b"XZZhello\0" + b"XBBs" + struct.pack("<i", 3) + struct.pack("<3h", 1, -2, 3)
        for t in tags:
            tg, ty, v = t.split(":", 2)
            if ty == "i":
                v = int(v)
                aux += tg.encode() + (b"C" + struct.pack("<B", v) if 0 <= v < 256 else b"c" + struct.pack("<b", v) if -128 <= v < 0 else b"S" + struct.pack("<H", v) if 0 <= v < 65536 else b"s" + struct.pack("<h", v) if -32768 <= v < 0 else b"i" + struct.pack("<i", v))
        body = struct.pack("<iiBBHHHiiii", ref, int(pos) - 1, len(qn) + 1, int(mapq), 4680, len(c), int(flag), l_seq, mref, int(pnext) - 1, int(tlen)) + qn.encode() + b"\0" + b"".join(struct.pack("<I", x) for x in c) + bytes(sq4) + b"\xff" * l_seq + aux
        out += struct.pack("<i", len(body)) + body
    with open(bam_path, "wb") as f:
        for i in list(range(0, len(out), block)) + [None]:
            chunk = bytes(out[i:i + block]) if i is not None else b""
            co = zlib.compressobj(6, zlib.DEFLATED, -15); cd = co.compress(chunk) + co.flush()
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(cd) + 8 - 1) + cd + struct.pack("<II", zlib.crc32(chunk), len(chunk)))


def read_sam(path, paired=True, max_frags=1 << 20, use_as=True, score_exp=1.0, tid_map=None):
    L = capi.lib(); h = C.c_void_p(); capi.check(L.sq_sam_open(str(path).encode(), int(paired), C.byref(h)), "sq_sam_open")
    names = [L.sq_sam_ref_name(h, i).decode() for i in range(L.sq_sam_num_refs(h))]; lens = [L.sq_sam_ref_len(h, i) for i in range(len(names))]
    if tid_map is not None:
        tm = np.ascontiguousarray(tid_map, np.uint32); capi.check(L.sq_sam_set_tid_map(h, tm.ctypes.data, len(tm)), "sq_sam_set_tid_map")
    offs = [np.zeros(1, np.uint64)]; alns = []; cnt = capi.SamCounts(); batches = 0
    while True:
        ab = capi.AlnBatch(); capi.check(L.sq_sam_next(h, max_frags, int(use_as), float(score_exp), C.byref(ab), C.byref(cnt)), "sq_sam_next")
        if ab.n == 0: break
        batches += 1
        ro = np.ctypeslib.as_array(ab.read_off, shape=(ab.n + 1,)).copy(); na = int(ro[-1])
        a = np.frombuffer(C.string_at(ab.aln, na * api.ALN_DTYPE.itemsize), api.ALN_DTYPE).copy() if na else np.zeros(0, api.ALN_DTYPE)
        offs.append(ro[1:] + offs[-1][-1]); alns.append(a)
    L.sq_sam_close(h)
    return names, lens, np.concatenate(offs), (np.concatenate(alns) if alns else np.zeros(0, api.ALN_DTYPE)), {k: int(getattr(cnt, k)) for k, _ in capi.SamCounts._fields_}, batches


def _toy_alignments(rng, n_frag, n_txp):
    """Random fragments: pairs on one or several transcripts, orphans of either side, with scores."""
    ro = [0]; rows = []
    for f in range(n_frag):
        k = int(rng.integers(1, 5)); kind = rng.choice([3, 3, 3, 1, 2])
        for t in sorted(rng.choice(n_txp, k, replace=False)):
            a = np.zeros(1, api.ALN_DTYPE)[0]
            a["tid"] = t; a["pos"] = int(rng.integers(0, 800)); a["fwd"] = int(rng.integers(0, 2)); a["read_len"] = 100; a["score"] = int(rng.integers(150, 201)); a["mate_status"] = kind
            if kind == 3:
                a["mate_fwd"] = 1 - a["fwd"]; a["mate_pos"] = a["pos"] + int(rng.integers(-40, 300)); a["mate_len"] = int(rng.integers(80, 101)); a["mate_score"] = int(rng.integers(150, 201))
            rows.append(a)
        ro.append(len(rows))
    return np.array(ro, np.uint64), np.array(rows, api.ALN_DTYPE)


def test_sam_reader_rebuilds_the_alignment_records(built, tmp_path):
    rng = np.random.default_rng(4); names = ["t%d" % i for i in range(12)]; lens = [1000 + 10 * i for i in range(12)]
    ro, aln = _toy_alignments(rng, 300, 12)
    for ext in ("sam", "sam.gz"):
        p = tmp_path / ("a." + ext); write_sam(p, names, lens, ro, aln, unaligned_every=7)
        n2, l2, ro2, aln2, cnt, nb = read_sam(p, paired=True, max_frags=64)            # several batches: a fragment never straddles two
        assert n2 == names and l2 == lens and nb == 5 and np.array_equal(ro2, ro)
        for f in ("tid", "pos", "fwd", "read_len", "mate_status", "score"): assert np.array_equal(aln2[f], aln[f]), f
        pr = aln["mate_status"] == 3
        for f in ("mate_pos", "mate_fwd", "mate_len", "mate_score"): assert np.array_equal(aln2[f][pr], aln[f][pr]), f
        # ReadPair::fragLen: |pos1 - pos2| + the length of the rightmost read; orphans 0
        want_fl = np.where(pr, np.abs(aln["pos"] - aln["mate_pos"]) + np.where(aln["pos"] < aln["mate_pos"], aln["mate_len"], aln["read_len"]), 0)
        assert np.array_equal(aln2["frag_len"], want_fl.astype(np.uint32))
        # hitType: pairs by strand and order of the starts (SalmonUtils.cpp:531-575), orphans as single-end S / A (:638-646)
        def fid(a):
            if a["mate_status"] != 3: return 0 | (3 << 1) | ((2 if a["fwd"] else 3) << 3)
            if a["fwd"] != a["mate_fwd"]:
                if a["fwd"]: return 1 | ((2 if a["pos"] <= a["mate_pos"] else 1) << 1) | (0 << 3)
                return 1 | ((2 if a["mate_pos"] <= a["pos"] else 1) << 1) | (1 << 3)
            return 1 | (0 << 1) | ((2 if a["fwd"] else 3) << 3)
        assert np.array_equal(aln2["format_id"], np.array([fid(a) for a in aln], np.uint8))
        # --useASWithoutCIGAR: exp(-scoreExp (bestAS - AS)) with AS = the sum of the mapped ends' tags
        tot = aln["score"].astype(np.int64) + np.where(pr, aln["mate_score"], 0)
        want = np.concatenate([np.exp(-1.0 * (tot[int(a):int(b)].max() - tot[int(a):int(b)])) for a, b in zip(ro[:-1], ro[1:])])
        assert np.allclose(aln2["est_aln_prob"], want, rtol=1e-15)
        assert cnt["num_fragments"] == 300 and cnt["num_unaligned"] == len(range(0, 300, 7)) and cnt["num_alignments"] == len(aln)
    # --noErrorModel: every alignment weighs 1; single-end files; a target that is not in the index is skipped
    _, _, _, a3, _, _ = read_sam(tmp_path / "a.sam", use_as=False); assert (a3["est_aln_prob"] == 1.0).all()
    tm = np.arange(12, dtype=np.uint32); tm[5] = 0xFFFFFFFF
    _, _, ro4, a4, c4, _ = read_sam(tmp_path / "a.sam", tid_map=tm)
    assert not (a4["tid"] == 5).any() and c4["num_skipped_unknown_target"] == int((aln["tid"] == 5).sum()) and len(a4) == len(aln) - c4["num_skipped_unknown_target"]
    se = aln.copy(); se["mate_status"] = 0; write_sam(tmp_path / "se.sam", names, lens, ro, se)
    _, _, ro5, a5, _, _ = read_sam(tmp_path / "se.sam", paired=False)
    assert np.array_equal(ro5, ro) and (a5["mate_status"] == 0).all() and np.array_equal(a5["pos"], aln["pos"]) and (a5["read_len"] == 100).all()
    # BAM is refused by name, a file without @SQ lines too
    open(tmp_path / "x.bam", "wb").write(b"BAM\1" + b"\0" * 64)
    with pytest.raises(capi.SalmonHipError, match="BAM"): read_sam(tmp_path / "x.bam")
    open(tmp_path / "nohdr.sam", "w").write("r1\t4\t*\t0\t0\t*\t*\t0\t0\tA\tI\n")
    with pytest.raises(capi.SalmonHipError, match="@SQ"): read_sam(tmp_path / "nohdr.sam")


def test_bam_gives_the_same_records_as_sam(built, tmp_path):
    """[r4] BAM input: header and records decoded from the binary form (BGZF inflated through the gzip reader) — the same targets, fragments, alignments,
    AS-based probabilities and counters as the SAM text they were made from; paired and single-end libraries; records that store no sequence (length
    from the CIGAR) and records that do; damaged files are refused."""
    rng = np.random.default_rng(14); names = ["tx%d" % i for i in range(9)]; lens = [900 + 13 * i for i in range(9)]
    ro, aln = _toy_alignments(rng, 400, 9)
    aln["score"] = rng.integers(-300, 70000, len(aln)); aln["mate_score"] = np.where(aln["mate_status"] == 3, rng.integers(-40, 300, len(aln)), 0)      # every integer width of the AS tag
    sam = tmp_path / "p.sam"; bam = tmp_path / "p.bam"; write_sam(sam, names, lens, ro, aln, unaligned_every=9); sam_to_bam(sam, bam)
    a = read_sam(sam, paired=True, max_frags=90); b = read_sam(bam, paired=True, max_frags=90)
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and a[3].tobytes() == b[3].tobytes() and a[4] == b[4] and a[5] == b[5]
    assert a[4]["num_fragments"] == 400 - 0 and a[4]["num_unaligned"] > 0
    # the BGZF members went through the parallel source (bgzf_source.h); through zlib's reader the same
    os.environ["SQ_SAM_BGZF"] = "0"
    try: c = read_sam(bam, paired=True, max_frags=90)
    finally: os.environ.pop("SQ_SAM_BGZF")
    assert np.array_equal(c[2], b[2]) and c[3].tobytes() == b[3].tobytes() and c[4] == b[4]
    flipped = bytearray(open(bam, "rb").read()); flipped[len(flipped) // 2] ^= 0x41; open(tmp_path / "flip.bam", "wb").write(bytes(flipped))
    with pytest.raises(Exception): read_sam(tmp_path / "flip.bam", paired=True)
    # single-end: the records carry their sequence
    se = aln.copy(); se["mate_status"] = 0; se["mate_pos"] = 0; se["mate_len"] = 0; se["mate_fwd"] = 0; se["mate_score"] = 0
    sam2 = tmp_path / "s.sam"; bam2 = tmp_path / "s.bam"; write_sam(sam2, names, lens, ro, se); sam_to_bam(sam2, bam2)
    a = read_sam(sam2, paired=False); b = read_sam(bam2, paired=False)
    assert a[0] == b[0] and np.array_equal(a[2], b[2]) and a[3].tobytes() == b[3].tobytes() and a[4] == b[4]
    # [r5] `-l A` reads the first record's FLAG through the same reader, whatever the container (the CLI used to read BAM as text there)
    L = capi.lib(); gz = tmp_path / "p.sam.gz"; gzip.open(gz, "wb").write(open(sam, "rb").read())
    for path, want_paired in ((sam, True), (gz, True), (bam, True), (sam2, False), (bam2, False)):
        fl = C.c_int(-1); assert L.sq_sam_first_flag(str(path).encode(), C.byref(fl)) == 0 and bool(fl.value & 1) == want_paired, path
    # a gzip stream cut short is an error on the zlib path too, not a shorter file
    cut = tmp_path / "cut.sam.gz"; open(cut, "wb").write(open(gz, "rb").read()[:-40])
    with pytest.raises(capi.SalmonHipError): read_sam(cut, paired=True)
    # damage: cut inside the header, cut inside a record, a record whose lengths exceed its block
    raw = gzip.open(bam, "rb").read()
    def refused(data):
        p = tmp_path / "bad.bam"; gzip.open(p, "wb").write(data); h = C.c_void_p()
        if L.sq_sam_open(str(p).encode(), 1, C.byref(h)) != 0: return True
        try:
            while True:
                ab = capi.AlnBatch()
                if L.sq_sam_next(h, 1000, 0, 1.0, C.byref(ab), None) != 0: return True
                if ab.n == 0: return False
        finally: L.sq_sam_close(h)
    assert refused(raw[:60]) and refused(raw[: len(raw) // 2 + 3])
    hdr_end = raw.index(b"tx8\0") + 8; broken = bytearray(raw); broken[hdr_end + 4 + 8] = 250          # l_read_name of the first record: longer than its block
    assert refused(bytes(broken))
    # and 300 randomly damaged copies: refused or read, never a crash (every length in a record is checked against its block)
    for it in range(300):
        b2 = bytearray(raw)
        for _ in range(int(rng.integers(1, 5))):
            op = int(rng.integers(0, 3)); q = int(rng.integers(0, len(b2)))
            if op == 0: b2[q] = int(rng.integers(0, 256))
            elif op == 1: del b2[q:]
            else: b2[q:q] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
            if not b2: b2 = bytearray(b"B")
        refused(bytes(b2))


@pytest.mark.gpu
def test_injected_alignments_run_the_same_stage_as_mapped_ones(small_world, tmp_path):
    """Alignments the mapper produced, written as SAM and read back, are injected in batches; the online model, the class table and the VBEM result equal
    the checker's on the records the reader produced — and, with the mapper's own conditional probabilities restored, the mapping-mode run."""
    w = small_world; w["idx"].to_device(0); opts = api.quant_opts(mini_batch_size=500, num_pre_burnin_frags=300, num_burnin_frags=2500)
    ctx = api.QuantContext(w["idx"], opts, device=0, max_batch_reads=8192)
    rb = api.make_read_batch(w["seq"], w["off"], w["n"], paired=True)
    ro, aln, mt, st = ctx.map_batch(rb); ctx.eq_accumulate(); eq_map = ctx.eq_finish()
    names = w["idx"].ref_names(); lens = w["idx"].ref_lens()
    keep = np.array([ro[f + 1] > ro[f] for f in range(w["n"])]); ro_k = np.concatenate([[0], np.cumsum((ro[1:] - ro[:-1])[keep])]).astype(np.uint64)
    write_sam(tmp_path / "m.sam", names, lens, ro_k, aln)
    _, _, ro_s, aln_s, cnt, _ = read_sam(tmp_path / "m.sam", use_as=True, score_exp=opts.score_exp)
    assert np.array_equal(ro_s, ro_k) and np.array_equal(aln_s["tid"], aln["tid"]) and np.array_equal(aln_s["pos"], aln["pos"])
    # the alignment-mode records through the HIP stage, 1000 fragments at a time, against the checker on the same records
    ctx.reset(); L = capi.lib(); nf = len(ro_s) - 1
    for lo in range(0, nf, 1000):
        hi = min(nf, lo + 1000); r = (ro_s[lo:hi + 1] - ro_s[lo]).astype(np.uint64); a = np.ascontiguousarray(aln_s[int(ro_s[lo]):int(ro_s[hi])])
        ab = capi.AlnBatch(hi - lo, r.ctypes.data_as(C.POINTER(C.c_uint64)), a.ctypes.data_as(C.POINTER(capi.Aln)), len(a), None)
        capi.check(L.sq_aln_inject(ctx.h, C.byref(ab), hi - lo), "sq_aln_inject"); ctx.eq_accumulate()
    eq_g = ctx.eq_finish(); lm_g, uq_g, tc_g, le_g = ctx.model()
    oidx = orc.OrcIndex(w["idx"]); ost = orc.OrcState(oidx, opts)
    for lo in range(0, nf, 1000):
        hi = min(nf, lo + 1000); ost.eq_accumulate((ro_s[lo:hi + 1] - ro_s[lo]).astype(np.uint64), np.ascontiguousarray(aln_s[int(ro_s[lo]):int(ro_s[hi])]), hi - lo)
    ost.finish(); eq_c = ost.eq_finish(); lm_c, uq_c, tc_c, le_c, _ = ost.model()
    for f in ("off", "tid", "bins", "count", "wq"): assert np.array_equal(getattr(eq_g, f), getattr(eq_c, f)), f
    assert np.array_equal(lm_g, lm_c) and np.array_equal(uq_g, uq_c) and np.array_equal(tc_g, tc_c) and np.array_equal(le_g, le_c)
    p_g = api.normalize_alphas(eq_g, lm_g, uq_g, tc_g); a_g, rep_g = ctx.em_optimize(np.exp(le_g), p_g, api.em_opts())
    a_c, rep_c = orc.em_optimize(eq_c, np.exp(le_c), orc.normalize_alphas(len(lm_c), eq_c, lm_c, uq_c, tc_c), api.em_opts())
    assert rep_g["iters"] == rep_c["iters"] and np.array_equal(a_g, a_c)
    assert int(eq_g.count.sum()) == int(eq_map.count.sum())                      # every fragment the mapper assigned is assigned here too
    ctx.free()


@pytest.mark.gpu
def test_cli_alignment_mode_quantifies_from_a_sam_file(small_world, tmp_path):
    w = small_world; w["idx"].to_device(0); exe = os.path.join(ROOT, "salmon_amd", "bin", "salmon-hip")
    ctx = api.QuantContext(w["idx"], api.quant_opts(), device=0, max_batch_reads=8192)
    ro, aln, mt, st = ctx.map_batch(api.make_read_batch(w["seq"], w["off"], w["n"], paired=True)); ctx.free()
    names = w["idx"].ref_names(); lens = w["idx"].ref_lens()
    keep = np.array([ro[f + 1] > ro[f] for f in range(w["n"])]); ro_k = np.concatenate([[0], np.cumsum((ro[1:] - ro[:-1])[keep])]).astype(np.uint64)
    write_sam(tmp_path / "m.sam.gz", names, lens, ro_k, aln, unaligned_every=50)
    w["tx"].write_fasta(str(tmp_path / "t.fa"))
    # [r5] without a flag the CIGAR-based error model runs (the reference's default); these records carry no sequences, so it has nothing to say and the job goes through
    subprocess.check_call([exe, "quant", "-t", str(tmp_path / "t.fa"), "-l", "IU", "-a", str(tmp_path / "m.sam.gz"), "-o", str(tmp_path / "out_em"), "-q"])
    subprocess.check_call([exe, "quant", "-t", str(tmp_path / "t.fa"), "-l", "IU", "-a", str(tmp_path / "m.sam.gz"), "-o", str(tmp_path / "out"), "--useASWithoutCIGAR", "-q"])
    import json
    meta = json.load(open(tmp_path / "out" / "aux_info" / "meta_info.json"))
    assert meta["mapping_type"] == "alignment" and meta["num_mapped"] == int(keep.sum()) and meta["num_processed"] == int(keep.sum()) + len(range(0, int(keep.sum()), 50))
    rows = [l.split("\t") for l in open(tmp_path / "out" / "quant.sf").read().splitlines()[1:]]
    assert abs(sum(float(x[4]) for x in rows) - keep.sum()) < 1e-3 * keep.sum() and abs(sum(float(x[3]) for x in rows) - 1e6) < 1.0
    # [r5] BAM input with the default `-l A`: paired-ness comes from the first record's FLAG (read through sq_sam, not as text), for a paired and a single-end file
    write_sam(tmp_path / "m.sam", names, lens, ro_k, aln, unaligned_every=50); sam_to_bam(tmp_path / "m.sam", tmp_path / "m.bam")
    subprocess.check_call([exe, "quant", "-t", str(tmp_path / "t.fa"), "-a", str(tmp_path / "m.bam"), "-o", str(tmp_path / "out_bam"), "--useASWithoutCIGAR", "-q"])
    mb = json.load(open(tmp_path / "out_bam" / "aux_info" / "meta_info.json")); lf = json.load(open(tmp_path / "out_bam" / "lib_format_counts.json"))
    assert mb["num_mapped"] == meta["num_mapped"] and mb["num_processed"] == meta["num_processed"] and lf["expected_format"] in ("IU", "ISF", "ISR"), lf["expected_format"]
    se = aln.copy(); se["mate_status"] = 0; se["mate_pos"] = 0; se["mate_len"] = 0; se["mate_fwd"] = 0; se["mate_score"] = 0
    write_sam(tmp_path / "s.sam", names, lens, ro_k, se); sam_to_bam(tmp_path / "s.sam", tmp_path / "s.bam")
    subprocess.check_call([exe, "quant", "-t", str(tmp_path / "t.fa"), "-a", str(tmp_path / "s.bam"), "-o", str(tmp_path / "out_se"), "--useASWithoutCIGAR", "-q"])
    ms = json.load(open(tmp_path / "out_se" / "aux_info" / "meta_info.json")); ls = json.load(open(tmp_path / "out_se" / "lib_format_counts.json"))
    assert ms["num_mapped"] == int(keep.sum()) and ls["expected_format"] in ("U", "SF", "SR"), ls["expected_format"]
