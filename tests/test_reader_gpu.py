"""[r4] FASTQ record splitting on the device (hip/fastq_dev.hip) against the host reader (host/reader.cpp, SQ_READER_DEVICE=0) on the same files:
the same batches, byte for byte — ragged read lengths, CR LF line ends, a last line without a newline, several files per mate, single-end input,
batches that end inside a file — and the same complaints about damaged input."""
import ctypes as C, os
import numpy as np
import pytest
from salmon_amd import api, capi

pytestmark = pytest.mark.gpu
HIP = None


def _d2h(ptr, nbytes):
    global HIP
    if HIP is None: HIP = C.CDLL("libamdhip64.so"); HIP.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    buf = np.zeros(max(nbytes, 1), np.uint8); assert HIP.hipMemcpy(buf.ctypes.data, ptr, nbytes, 2) == 0; return buf[:nbytes]


def _drain(files1, files2, batch, device):
    """All batches of a reader as (seq bytes, offsets) pairs; device batches are copied back to the host."""
    os.environ["SQ_READER_DEVICE"] = "1" if device else "0"
    L = capi.lib(); a1 = (C.c_char_p * len(files1))(*[f.encode() for f in files1]); h = C.c_void_p()
    a2 = (C.c_char_p * len(files2))(*[f.encode() for f in files2]) if files2 else None
    capi.check(L.sq_reader_open(a1, len(files1), a2, len(files2) if files2 else 0, batch, 3, C.byref(h)), "sq_reader_open")
    out = []
    try:
        while True:
            rb = capi.ReadBatch(); s = C.c_int(-1)
            capi.check(L.sq_reader_next(h, C.byref(rb), C.byref(s)), "sq_reader_next")
            if rb.n == 0: break
            nrec = rb.n * (2 if rb.paired else 1)
            if rb.on_device:
                off = _d2h(rb.seq_off, (nrec + 1) * 8).view(np.uint64).copy(); seq = _d2h(rb.seq, int(off[-1])).copy()
            else:
                off = np.ctypeslib.as_array(C.cast(rb.seq_off, C.POINTER(C.c_uint64)), shape=(nrec + 1,)).copy()
                seq = np.ctypeslib.as_array(C.cast(rb.seq, C.POINTER(C.c_uint8)), shape=(int(off[-1]),)).copy()
            out.append((int(rb.n), bool(rb.on_device), seq, off)); L.sq_reader_release(h, s.value)
    finally:
        L.sq_reader_close(h); os.environ.pop("SQ_READER_DEVICE", None)
    return out


def _write(path, recs, eol="\n", last_newline=True):
    body = eol.join("@%s%s%s%s+%s%s" % (n, eol, s, eol, eol, q) for n, s, q in recs) + (eol if last_newline else "")
    open(path, "wb").write(body.encode())


def _recs(rng, n, tag, lo=30, hi=151):
    out = []
    for i in range(n):
        L = int(rng.integers(lo, hi)); s = "".join(rng.choice(list("ACGTN"), L, p=[0.24, 0.24, 0.24, 0.24, 0.04])); q = "".join(chr(int(x)) for x in rng.integers(33, 74, L))
        out.append(("%s.%d some comment@+" % (tag, i), s, q))      # '@' and '+' inside the header line must not confuse the splitter
    return out


def test_device_batches_equal_host_batches(built, tmp_path):
    rng = np.random.default_rng(8)
    r1 = _recs(rng, 23456, "a"); r2 = _recs(rng, 23456, "b")
    cases = []
    f1, f2 = str(tmp_path / "p_1.fq"), str(tmp_path / "p_2.fq"); _write(f1, r1); _write(f2, r2); cases.append(("paired", [f1], [f2], 5000))
    g1, g2 = str(tmp_path / "c_1.fq"), str(tmp_path / "c_2.fq"); _write(g1, r1, eol="\r\n"); _write(g2, r2, eol="\r\n", last_newline=False); cases.append(("CR LF, no last newline", [g1], [g2], 7777))
    # several files per mate: the first of each lacks its last newline; batches straddle the file boundary
    h = [str(tmp_path / ("m%d_%d.fq" % (k, m))) for k in range(3) for m in (1, 2)]
    cuts = [0, 9000, 9001, 23456]
    for k in range(3): _write(h[2 * k], r1[cuts[k]:cuts[k + 1]], last_newline=(k != 0)); _write(h[2 * k + 1], r2[cuts[k]:cuts[k + 1]], last_newline=(k != 0))
    cases.append(("three files per mate", h[0::2], h[1::2], 4096))
    cases.append(("single-end", [f1], None, 6000))
    cases.append(("one batch holds everything", [f1], [f2], 30000))
    # [r5] blank lines at the very end: behind mate 2 only; behind a file whose records fill the batches exactly (the round that ends the input is
    # then nothing but the blank line); several of them, and with CR LF
    b2 = str(tmp_path / "bl_2.fq"); open(b2, "wb").write(open(f2, "rb").read() + b"\n"); cases.append(("mate 2 ends with a blank line", [f1], [b2], 5000))
    e1 = str(tmp_path / "ex_1.fq"); _write(e1, r1[:20000]); open(e1, "ab").write(b"\n"); cases.append(("single-end, k * batch records + a blank line", [e1], None, 5000))
    e2 = str(tmp_path / "ex_2.fq"); _write(e2, r2[:20000]); open(e2, "ab").write(b"\n\n\n"); cases.append(("paired, k * batch records + blank lines", [e1], [e2], 10000))
    c1 = str(tmp_path / "ec_1.fq"); _write(c1, r1[:20000], eol="\r\n"); open(c1, "ab").write(b"\r\n\r\n"); cases.append(("CR LF, k * batch records + blank lines", [c1], None, 4000))
    # [r5] gzip / BGZF files take the device path too (inflated by the reader's pool into the ring, split on the device): a plain gzip stream big enough for
    # the parallel inflater, a multi-member BGZF file, several files per mate, a last line without its newline
    import gzip
    from test_reader import _bgzf_write
    z1, z2 = str(tmp_path / "z_1.fq.gz"), str(tmp_path / "z_2.fq.gz"); open(z1, "wb").write(gzip.compress(open(f1, "rb").read(), 6)); open(z2, "wb").write(gzip.compress(open(g2, "rb").read().replace(b"\r\n", b"\n"), 1))
    cases.append(("gzip", [z1], [z2], 5000))
    y1, y2 = str(tmp_path / "y_1.fq.gz"), str(tmp_path / "y_2.fq.gz"); _bgzf_write(y1, open(f1, "rb").read()); _bgzf_write(y2, open(f2, "rb").read(), block=30011)
    cases.append(("BGZF", [y1], [y2], 7001)); cases.append(("BGZF single-end, one batch", [y1], None, 30000))
    hz = []
    for k in range(3):
        for m in (1, 2):
            zp = str(tmp_path / ("mz%d_%d.fq.gz" % (k, m))); raw = open(h[2 * k + (m - 1)], "rb").read()
            if k == 1: _bgzf_write(zp, raw)
            else: open(zp, "wb").write(gzip.compress(raw, 4))
            hz.append(zp)
    cases.append(("three compressed files per mate (gzip, BGZF, gzip)", hz[0::2], hz[1::2], 4096))
    for name, a, b, batch in cases:
        dev = _drain(a, b, batch, True); host = _drain(a, b, batch, False)
        assert all(d[1] for d in dev) and not any(x[1] for x in host), name          # the device path really ran, the host path really did not
        assert [d[0] for d in dev] == [x[0] for x in host], name
        for d, x in zip(dev, host):
            assert np.array_equal(d[3], x[3]) and d[2].tobytes() == x[2].tobytes(), name
    # and the batches are what the files say
    dev = _drain([f1], [f2], 5000, True); i = 0
    for n, _, seq, off in dev:
        for r in range(n):
            assert seq[int(off[2 * r]):int(off[2 * r + 1])].tobytes().decode() == r1[i][1] and seq[int(off[2 * r + 1]):int(off[2 * r + 2])].tobytes().decode() == r2[i][1]; i += 1
    assert i == 23456


def test_device_reader_feeds_the_mapper_and_reports_damage(small_world, tmp_path):
    w = small_world; w["idx"].to_device(0)
    recs = w["seq"].reshape(2 * w["n"], 100)
    def fq(path, rows):
        with open(path, "wb") as f:
            for i, r in enumerate(rows): f.write(b"@r%d\n" % i + r.tobytes() + b"\n+\n" + b"I" * 100 + b"\n")
    f1, f2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq"); fq(f1, recs[0::2]); fq(f2, recs[1::2])
    ctx = api.QuantContext(w["idx"], api.quant_opts(), device=0, max_batch_reads=4096)
    ro_ref, aln_ref, _, _ = ctx.map_batch(api.make_read_batch(w["seq"], w["off"], w["n"], paired=True))
    os.environ["SQ_READER_DEVICE"] = "1"
    L = capi.lib(); a1 = (C.c_char_p * 1)(f1.encode()); a2 = (C.c_char_p * 1)(f2.encode()); h = C.c_void_p()
    capi.check(L.sq_reader_open(a1, 1, a2, 1, 1500, 3, C.byref(h)), "sq_reader_open")
    got = []; tot = 0
    while True:
        rb = capi.ReadBatch(); s = C.c_int(-1); capi.check(L.sq_reader_next(h, C.byref(rb), C.byref(s)), "sq_reader_next")
        if rb.n == 0: break
        assert rb.on_device == 1
        ro, aln, _, _ = ctx.map_batch(rb); got.append(aln); tot += rb.n; L.sq_reader_release(h, s.value)
    assert L.sq_reader_total(h) == w["n"] == tot
    L.sq_reader_close(h)
    assert np.concatenate(got).tobytes() == aln_ref.tobytes()                      # batches cut from device-split text map like the in-memory reads
    # damage: a record without its '+' line, a truncated last record, mate files of different lengths
    bad = str(tmp_path / "bad.fq"); txt = open(f1, "rb").read().split(b"\n"); txt[4 * 700 + 2] = b"-"; open(bad, "wb").write(b"\n".join(txt))
    for files, msg in (((bad, f2), "no '\\+' line"), ((f1, None), None)):
        pass
    def first_error(p1, p2):
        a1 = (C.c_char_p * 1)(p1.encode()); a2 = (C.c_char_p * 1)(p2.encode()); h = C.c_void_p(); capi.check(L.sq_reader_open(a1, 1, a2, 1, 1500, 3, C.byref(h)), "open")
        try:
            while True:
                rb = capi.ReadBatch(); s = C.c_int(-1); rc = L.sq_reader_next(h, C.byref(rb), C.byref(s))
                if rc != 0: return L.sq_last_error().decode()
                if rb.n == 0: return None
                L.sq_reader_release(h, s.value)
        finally: L.sq_reader_close(h)
    e = first_error(bad, f2); assert e and "record 701" in e and "'+'" in e, e
    trunc = str(tmp_path / "trunc.fq"); open(trunc, "wb").write(open(f1, "rb").read()[:-150])
    e = first_error(trunc, f2); assert e and ("middle of a record" in e or "different numbers" in e or "quality" in e), e
    short = str(tmp_path / "short.fq"); fq(short, recs[0::2][:3000])
    e = first_error(short, f2); assert e and "different numbers of records" in e, e
    os.environ.pop("SQ_READER_DEVICE", None); ctx.free()


def test_bgzf_members_inflated_on_the_device(built, tmp_path):
    """[r5] All-BGZF input: the members are inflated by hip/inflate_dev.hip into chunk buffers and the batches cut out of those.  Against the host reader:
    several files per mate (the first without its last newline: the line end it gets is a pseudo-member), blank lines at the very end, CR LF, k * batch
    records exactly, chunks of 40 members (batches span chunks, the ring of chunk buffers goes round many times) and of thousands; the same files through
    the host's inflating threads (SQ_READER_BGZF_DEVICE=0); a damaged member is named."""
    from test_reader import _bgzf_write
    rng = np.random.default_rng(18); r1 = _recs(rng, 23456, "a"); r2 = _recs(rng, 23456, "b")
    def bz(name, recs, block=0xff00, **kw):
        raw = str(tmp_path / (name + ".fq")); _write(raw, recs, **kw); z = str(tmp_path / (name + ".fq.gz")); _bgzf_write(z, open(raw, "rb").read(), block=block); return z
    cases = [("paired", [bz("a1", r1)], [bz("a2", r2, block=30011)], 5000)]
    cuts = [0, 9000, 9001, 23456]
    m1 = [bz("m%d_1" % k, r1[cuts[k]:cuts[k + 1]], last_newline=(k != 0), block=20000 + 7 * k) for k in range(3)]
    m2 = [bz("m%d_2" % k, r2[cuts[k]:cuts[k + 1]], last_newline=(k != 1)) for k in range(3)]
    cases.append(("three files per mate", m1, m2, 4096))
    cases.append(("single-end, one batch", [cases[0][1][0]], None, 30000))
    e1 = str(tmp_path / "e1.fq"); _write(e1, r1[:20000]); open(e1, "ab").write(b"\n\n"); z1 = str(tmp_path / "e1.fq.gz"); _bgzf_write(z1, open(e1, "rb").read())
    e2 = str(tmp_path / "e2.fq"); _write(e2, r2[:20000], eol="\r\n"); open(e2, "ab").write(b"\r\n"); z2 = str(tmp_path / "e2.fq.gz"); _bgzf_write(z2, open(e2, "rb").read(), block=4099)
    cases.append(("k * batch records + blank lines, CR LF in mate 2", [z1], [z2], 5000))
    for members in ("40", None):          # 40 members of 4 to 64 KB: batches of 1000 records fit the ring of six such chunks, not by much
        if members: os.environ["SQ_READER_BGZF_MEMBERS"] = members
        try:
            for name, a, b, batch in cases:
                if members: batch = min(batch, 1000)
                dev = _drain(a, b, batch, True); host = _drain(a, b, batch, False)
                assert all(d[1] for d in dev) and [d[0] for d in dev] == [x[0] for x in host], (name, members)
                for d, x in zip(dev, host): assert np.array_equal(d[3], x[3]) and d[2].tobytes() == x[2].tobytes(), (name, members)
        finally: os.environ.pop("SQ_READER_BGZF_MEMBERS", None)
    os.environ["SQ_READER_BGZF_DEVICE"] = "0"
    try:
        name, a, b, batch = cases[1]; dev = _drain(a, b, batch, True); host = _drain(a, b, batch, False)
        assert all(d[1] for d in dev) and [d[0] for d in dev] == [x[0] for x in host]
        for d, x in zip(dev, host): assert np.array_equal(d[3], x[3]) and d[2].tobytes() == x[2].tobytes()
    finally: os.environ.pop("SQ_READER_BGZF_DEVICE", None)
    # different numbers of records; a member with a flipped bit; a cut file
    with pytest.raises(Exception, match="different numbers"): _drain([cases[0][1][0]], [bz("short2", r2[:23000])], 5000, True)
    good = open(cases[0][1][0], "rb").read(); bad = bytearray(good); bad[len(bad) // 2] ^= 0x20; zb = str(tmp_path / "bad.fq.gz"); open(zb, "wb").write(bytes(bad))
    with pytest.raises(Exception, match="BGZF|member|record"): _drain([zb], None, 5000, True)
    zc = str(tmp_path / "cut.fq.gz"); open(zc, "wb").write(good[: len(good) * 2 // 3])
    with pytest.raises(Exception, match="truncated|BGZF|middle of a record"): _drain([zc], None, 5000, True)
    # [r6] a member whose XLEN leaves no room for a deflate stream in front of the trailer (header + 8 >= its size) although its trailer claims text: refused by
    # the scan on both paths — the stream length handed to the device would wrap otherwise — whether XLEN swallows exactly the stream or reaches far past the member
    ms0 = (good[16] | (good[17] << 8)) + 1; ms1 = (good[ms0 + 16] | (good[ms0 + 17] << 8)) + 1
    for xlen in (ms1 - 20, ms1 - 12, ms1 + 3000):
        x = bytearray(good); x[ms0 + 10] = xlen & 0xFF; x[ms0 + 11] = xlen >> 8; zx = str(tmp_path / ("xlen%d.fq.gz" % xlen)); open(zx, "wb").write(bytes(x))
        for device in (True, False):
            with pytest.raises(Exception, match="truncated BGZF member"): _drain([zx], None, 5000, device)


def test_ordinary_gzip_files_inflated_on_the_device(built, tmp_path):
    """[r6] All-gzip input (no BGZF member table): every file is inflated by hip/gzip_dev.hip — block starts found on the device, a wave per span, windows resolved,
    CRC-32 and length of every member checked — and the batches are cut out of that text.  Against the host reader: several files per mate (the first without its last
    newline), a file of three members (one of them empty), CR LF, levels 1 and 9, k * batch records exactly; the same files through the host's inflating threads
    (SQ_READER_GZIP_DEVICE=0); a flipped bit, a wrong checksum and a cut file are refused with the file's name."""
    import gzip, zlib
    rng = np.random.default_rng(28); r1 = _recs(rng, 30000, "a"); r2 = _recs(rng, 30000, "b")
    def gz(name, recs, level=6, members=None, **kw):
        raw = str(tmp_path / (name + ".fq")); _write(raw, recs, **kw); data = open(raw, "rb").read(); z = str(tmp_path / (name + ".fq.gz"))
        if members: cuts = [0] + [len(data) * k // members for k in range(1, members)] + [len(data)]; blob = b"".join(gzip.compress(data[cuts[k]:cuts[k + 1]], level) for k in range(members)) + gzip.compress(b"", level)
        else: blob = gzip.compress(data, level)
        open(z, "wb").write(blob); return z
    cases = [("paired", [gz("a1", r1)], [gz("a2", r2, level=1)], 5000)]
    cuts = [0, 11000, 11001, 30000]
    m1 = [gz("m%d_1" % k, r1[cuts[k]:cuts[k + 1]], last_newline=(k != 0), level=(9 if k == 2 else 6)) for k in range(3)]
    m2 = [gz("m%d_2" % k, r2[cuts[k]:cuts[k + 1]], last_newline=(k != 1)) for k in range(3)]
    cases.append(("three files per mate", m1, m2, 4096))
    cases.append(("three members + an empty one, CR LF in mate 2", [gz("c1", r1, members=3)], [gz("c2", r2, eol="\r\n")], 7000))
    cases.append(("single-end, k * batch records", [gz("s1", r1)], None, 10000))
    for name, a, b, batch in cases:
        dev = _drain(a, b, batch, True); host = _drain(a, b, batch, False)
        assert all(d[1] for d in dev) and [d[0] for d in dev] == [x[0] for x in host], name
        for d, x in zip(dev, host): assert np.array_equal(d[3], x[3]) and d[2].tobytes() == x[2].tobytes(), name
    os.environ["SQ_READER_GZIP_DEVICE"] = "0"
    try:
        name, a, b, batch = cases[1]; dev = _drain(a, b, batch, True); host = _drain(a, b, batch, False)
        assert all(d[1] for d in dev) and [d[0] for d in dev] == [x[0] for x in host]
        for d, x in zip(dev, host): assert np.array_equal(d[3], x[3]) and d[2].tobytes() == x[2].tobytes()
    finally: os.environ.pop("SQ_READER_GZIP_DEVICE", None)
    good = open(cases[0][1][0], "rb").read()
    bad = bytearray(good); bad[len(bad) // 2] ^= 0x20; zb = str(tmp_path / "bad.fq.gz"); open(zb, "wb").write(bytes(bad))
    with pytest.raises(Exception, match="bad.fq.gz"): _drain([zb], None, 5000, True)
    bad = bytearray(good); bad[-7] ^= 0x01; zs = str(tmp_path / "sum.fq.gz"); open(zs, "wb").write(bytes(bad))
    with pytest.raises(Exception, match="sum.fq.gz.*checksum"): _drain([zs], None, 5000, True)
    zc = str(tmp_path / "cut.fq.gz"); open(zc, "wb").write(good[: len(good) * 2 // 3])
    with pytest.raises(Exception, match="cut.fq.gz"): _drain([zc], None, 5000, True)
