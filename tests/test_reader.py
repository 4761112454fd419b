"""Host read pipeline (sq_reader): FASTQ/FASTA, gzip or plain, several files per mate,
CRLF, batches that do not divide the input, error paths.  No GPU needed."""
import ctypes as C, gzip, os
import numpy as np
import pytest
from salmon_amd import capi


def _open(f1, f2, batch, slots=3):
    L = capi.lib()
    a1 = (C.c_char_p * len(f1))(*[x.encode() for x in f1]); a2 = (C.c_char_p * len(f2))(*[x.encode() for x in f2]) if f2 else None
    h = C.c_void_p(); rc = L.sq_reader_open(a1, len(f1), a2, len(f2) if f2 else 0, batch, slots, C.byref(h))
    assert rc == 0, L.sq_last_error()
    return h


def _drain(h):
    L = capi.lib(); out = []
    while True:
        rb = capi.ReadBatch(); slot = C.c_int(-1)
        rc = L.sq_reader_next(h, C.byref(rb), C.byref(slot))
        if rc != 0:
            return out, L.sq_last_error().decode()
        if rb.n == 0:
            return out, None
        nrec = rb.n * (2 if rb.paired else 1)
        off = np.ctypeslib.as_array(C.cast(rb.seq_off, C.POINTER(C.c_uint64)), shape=(nrec + 1,)).copy()
        seq = C.string_at(rb.seq, int(off[-1]))
        out.append([seq[int(off[i]):int(off[i + 1])] for i in range(nrec)])
        L.sq_reader_release(h, slot.value)


def _fq(path, recs, gz=False, crlf=False, fasta=False, trailing_newline=True):
    nl = "\r\n" if crlf else "\n"
    txt = "".join((">r%d%s%s%s" % (i, nl, s, nl)) if fasta else ("@r%d%s%s%s+%s%s%s" % (i, nl, s, nl, nl, "I" * len(s), nl)) for i,
        s in enumerate(recs))
    if not trailing_newline: txt = txt.rstrip("\r\n")
    (gzip.open(path, "wt", newline="") if gz else open(path, "w", newline="")).write(txt)


def test_reader_paired_multi_file_formats(built, tmp_path):
    rng = np.random.default_rng(0)
    def reads(n): return ["".join(rng.choice(list("ACGTN"), size=int(rng.integers(30, 151)))) for _ in range(n)]
    a1, a2, b1, b2 = reads(700), reads(700), reads(333), reads(333)
    _fq(tmp_path / "a_1.fq.gz", a1, gz=True); _fq(tmp_path / "a_2.fq.gz", a2, gz=True, crlf=True)
    _fq(tmp_path / "b_1.fq", b1, trailing_newline=False); _fq(tmp_path / "b_2.fa", b2, fasta=True)
    h = _open([str(tmp_path / "a_1.fq.gz"), str(tmp_path / "b_1.fq")], [str(tmp_path / "a_2.fq.gz"), str(tmp_path / "b_2.fa")], batch=256)
    got, err = _drain(h)
    assert err is None and capi.lib().sq_reader_total(h) == 1033
    capi.lib().sq_reader_close(h)
    assert [len(b) // 2 for b in got] == [256, 256, 256, 256, 9]
    flat = [r for b in got for r in b]
    want = [x.encode() for pair in zip(a1 + b1, a2 + b2) for x in pair]
    assert flat == want


def test_reader_single_end_and_errors(built, tmp_path):
    recs = ["ACGT" * 10, "", "TTTTGGGG"]
    _fq(tmp_path / "s.fq", recs)
    h = _open([str(tmp_path / "s.fq")], None, batch=2); got, err = _drain(h); capi.lib().sq_reader_close(h)
    assert err is None and [r for b in got for r in b] == [r.encode() for r in recs]
    # mate files of different length
    _fq(tmp_path / "m1.fq", ["ACGT"] * 5); _fq(tmp_path / "m2.fq", ["ACGT"] * 4)
    h = _open([str(tmp_path / "m1.fq")], [str(tmp_path / "m2.fq")], batch=100); got, err = _drain(h); capi.lib().sq_reader_close(h)
    assert err is not None and "different numbers of records" in err
    # truncated record
    open(tmp_path / "t.fq", "w").write("@r0\nACGT\n+\nIIII\n@r1\nACG")
    h = _open([str(tmp_path / "t.fq")], None, batch=100); got, err = _drain(h); capi.lib().sq_reader_close(h)
    assert err is not None and "truncated" in err
    # not FASTQ/FASTA
    open(tmp_path / "x.fq", "w").write("hello\nworld\n")
    h = _open([str(tmp_path / "x.fq")], None, batch=100); got, err = _drain(h); capi.lib().sq_reader_close(h)
    assert err is not None and "does not start" in err
    # missing file
    h = _open([str(tmp_path / "nope.fq")], None, batch=100); got, err = _drain(h); capi.lib().sq_reader_close(h)
    assert err is not None and "cannot open" in err


def test_reader_refuses_when_all_slots_busy(built, tmp_path):
    _fq(tmp_path / "s.fq", ["ACGT"] * 10)
    L = capi.lib(); h = _open([str(tmp_path / "s.fq")], None, batch=2, slots=2)
    rb = capi.ReadBatch(); s = C.c_int()
    assert L.sq_reader_next(h, C.byref(rb), C.byref(s)) == 0 and L.sq_reader_next(h, C.byref(rb), C.byref(s)) == 0
    assert L.sq_reader_next(h, C.byref(rb), C.byref(s)) != 0 and b"in use" in L.sq_last_error()
    L.sq_reader_release(h, 0)
    assert L.sq_reader_next(h, C.byref(rb), C.byref(s)) == 0 and rb.n == 2
    L.sq_reader_close(h)


def test_reader_multi_line_records(built, tmp_path):
    # kseq-style input: sequences / qualities wrapped over several lines, a quality line that begins with '@',
    # an empty record, blank lines, a FASTA file that ends without a newline
    rng = np.random.default_rng(3)
    recs = ["".join(rng.choice(list("ACGT"), size=int(n))) for n in (150, 61, 60, 1, 0, 250, 120)]
    def wrap(s, w): return [s[i:i + w] for i in range(0, len(s), w)] or [""]
    with open(tmp_path / "w.fa", "w") as f:
        for i, s in enumerate(recs):
            f.write(">r%d some description\n" % i + "\n".join(wrap(s, 60)) + ("\n\n" if i % 2 else "\n"))
    with open(tmp_path / "w.fq", "w") as f:
        for i, s in enumerate(recs):
            q = ("@" + "I" * (len(s) - 1)) if s else ""
            f.write("@r%d\n%s\n+r%d\n%s\n" % (i, "\n".join(wrap(s, 50)), i, "\n".join(wrap(q, 50))))
    open(tmp_path / "e.fa", "w").write(">a\nACGT\nAC\n>b\nGG\nT")
    for name in ("w.fa", "w.fq"):
        h = _open([str(tmp_path / name)], None, batch=3); got, err = _drain(h); capi.lib().sq_reader_close(h)
        assert err is None, err
        assert [r for b in got for r in b] == [r.encode() for r in recs], name
    h = _open([str(tmp_path / "e.fa")], None, batch=8); got, err = _drain(h); capi.lib().sq_reader_close(h)
    assert err is None and got == [[b"ACGTAC", b"GGT"]]
    # quality longer than the sequence / sequence without quality
    open(tmp_path / "q.fq", "w").write("@r0\nACGT\n+\nIIIII\n")
    h = _open([str(tmp_path / "q.fq")], None, batch=8); got, err = _drain(h); capi.lib().sq_reader_close(h)
    assert err is not None and "quality string longer" in err
    open(tmp_path / "u.fq", "w").write("@r0\nACGT\n+\nII\n")
    h = _open([str(tmp_path / "u.fq")], None, batch=8); got, err = _drain(h); capi.lib().sq_reader_close(h)
    assert err is not None and "truncated" in err


@pytest.mark.parametrize("gz", [False, True])
def test_reader_fast_path_across_chunks(built, tmp_path, gz, monkeypatch):
    # the parallel path (4-line FASTQ): files larger than several 8 MB chunks, ragged read lengths, quality lines that begin with '@',
    # CRLF in one mate, two files per mate, a batch size that divides nothing; the same bytes must come out as from the kseq-rules path
    rng = np.random.default_rng(11)
    def write(path, n, crlf, seed):
        r = np.random.default_rng(seed); nl = b"\r\n" if crlf else b"\n"
        lens = r.integers(80, 152, n); bases = r.choice(np.frombuffer(b"ACGTN", np.uint8), size=int(lens.sum()), p=[.24, .24, .24, .24, .04]).tobytes()
        out = []; p = 0; recs = []
        for i, l in enumerate(lens):
            s = bases[p:p + l]; p += l; recs.append(s)
            q = (b"@" if i % 7 == 0 else b"I") + b"F" * (l - 1)
            out.append(b"@read%d/1 extra" % i + nl + s + nl + b"+" + nl + q + nl)
        data = b"".join(out)
        (gzip.open(path, "wb", compresslevel=1) if gz else open(path, "wb")).write(data)
        return recs
    ext = ".fq.gz" if gz else ".fq"
    n1, n2 = 90000, 30011
    a1 = write(tmp_path / ("a_1" + ext), n1, False, 1); a2 = write(tmp_path / ("a_2" + ext), n1, True, 2)
    b1 = write(tmp_path / ("b_1" + ext), n2, False, 3); b2 = write(tmp_path / ("b_2" + ext), n2, False, 4)
    assert os.path.getsize(tmp_path / ("a_1" + ext)) > (3 << 20)
    f1 = [str(tmp_path / ("a_1" + ext)), str(tmp_path / ("b_1" + ext))]; f2 = [str(tmp_path / ("a_2" + ext)), str(tmp_path / ("b_2" + ext))]
    want = [x for pair in zip(a1 + b1, a2 + b2) for x in pair]
    for threads in ("1", "5"):
        monkeypatch.setenv("SQ_READER_THREADS", threads)
        h = _open(f1, f2, batch=33333); got, err = _drain(h)
        assert err is None, err
        assert capi.lib().sq_reader_total(h) == n1 + n2
        capi.lib().sq_reader_close(h)
        assert [len(b) // 2 for b in got][:3] == [33333] * 3 and [r for b in got for r in b] == want
    monkeypatch.setenv("SQ_READER_SAFE", "1")
    h = _open(f1, f2, batch=50000); got, err = _drain(h); capi.lib().sq_reader_close(h)
    assert err is None and [r for b in got for r in b] == want


@pytest.mark.parametrize("safe", [False, True])
def test_reader_keeps_read_names_on_request(built, tmp_path, safe, monkeypatch):
    # SQ_READER_KEEP_NAMES: name = header up to the first blank (kseq's name field); both parser paths; without the flag the call is refused
    if safe: monkeypatch.setenv("SQ_READER_SAFE", "1")
    L = capi.lib(); rng = np.random.default_rng(5); n = 5000
    hdr = ["frag%d/1 extra words" % i if i % 3 == 0 else ("frag%d\tx" % i if i % 3 == 1 else "frag%d" % i) for i in range(n)]
    seqs = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(40, 120)))) for _ in range(n)]
    for m in (1, 2):
        with open(tmp_path / ("n_%d.fq" % m), "w") as f:
            for i in range(n): f.write("@%s\n%s\n+\n%s\n" % (hdr[i].replace("/1", "/%d" % m), seqs[i], "I" * len(seqs[i])))
    f1 = [str(tmp_path / "n_1.fq")]; f2 = [str(tmp_path / "n_2.fq")]
    a1 = (C.c_char_p * 1)(f1[0].encode()); a2 = (C.c_char_p * 1)(f2[0].encode())
    h = C.c_void_p(); assert L.sq_reader_open_ex(a1, 1, a2, 1, 1500, 3, 1, C.byref(h)) == 0
    names = []
    while True:
        rb = capi.ReadBatch(); slot = C.c_int(-1)
        assert L.sq_reader_next(h, C.byref(rb), C.byref(slot)) == 0
        if rb.n == 0: break
        nm = C.c_void_p(); no = C.POINTER(C.c_uint64)()
        assert L.sq_reader_names(h, slot.value, C.byref(nm), C.byref(no)) == 0
        off = [no[i] for i in range(rb.n + 1)]; raw = C.string_at(nm, off[-1])
        names += [raw[off[i]:off[i + 1]].decode() for i in range(rb.n)]
        L.sq_reader_release(h, slot.value)
    L.sq_reader_close(h)
    assert names == [("frag%d/1" % i if i % 3 == 0 else "frag%d" % i) for i in range(n)]
    h = _open(f1, f2, batch=100); rb = capi.ReadBatch(); slot = C.c_int(-1); L.sq_reader_next(h, C.byref(rb), C.byref(slot))
    nm = C.c_void_p(); no = C.POINTER(C.c_uint64)()
    assert L.sq_reader_names(h, slot.value, C.byref(nm), C.byref(no)) != 0 and b"KEEP_NAMES" in L.sq_last_error()
    L.sq_reader_close(h)


def _bgzf_write(path, data, block=60000):
    """bgzip-style file: gzip members of at most 64 KB, each with the 'BC' extra field naming its compressed size"""
    import struct, zlib
    with open(path, "wb") as f:
        for i in list(range(0, len(data), block)) + [None]:
            chunk = data[i:i + block] if i is not None else b""          # the last, empty member is bgzip's end-of-file marker
            co = zlib.compressobj(6, zlib.DEFLATED, -15); cd = co.compress(chunk) + co.flush()
            bsize = 18 + len(cd) + 8
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize - 1) + cd + struct.pack("<II", zlib.crc32(chunk), len(chunk)))


def test_reader_bgzf_members_are_inflated_in_parallel_and_in_order(built, tmp_path):
    # a BGZF file gives the same batches as the plain file; a damaged member is reported, not skipped
    rng = np.random.default_rng(11); n = 20000
    seqs1 = ["".join(rng.choice(list("ACGTN"), size=int(rng.integers(50, 151)))) for _ in range(n)]
    seqs2 = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(50, 151)))) for _ in range(n)]
    def text(seqs, m): return "".join("@q%d/%d\n%s\n+\n%s\n" % (i, m, s if i % 7 else (s[:5] * 40)[:len(s)], "F" * len(s)) for i, s in enumerate(seqs)).encode()   # constant qualities and, every seventh read, a tandem repeat of period 5: long matches at distances 1 and 5 (the BGZF inflater's strided copy)
    t1, t2 = text(seqs1, 1), text(seqs2, 2)
    open(tmp_path / "p_1.fq", "wb").write(t1); open(tmp_path / "p_2.fq", "wb").write(t2)
    _bgzf_write(tmp_path / "b_1.fq.gz", t1); _bgzf_write(tmp_path / "b_2.fq.gz", t2, block=30011)
    assert gzip.open(tmp_path / "b_1.fq.gz", "rb").read() == t1            # a valid multi-member gzip file for everybody else
    h = _open([str(tmp_path / "p_1.fq")], [str(tmp_path / "p_2.fq")], batch=3000); want, err = _drain(h); capi.lib().sq_reader_close(h); assert err is None
    h = _open([str(tmp_path / "b_1.fq.gz")], [str(tmp_path / "b_2.fq.gz")], batch=3000); got, err = _drain(h); capi.lib().sq_reader_close(h)
    assert err is None and got == want and sum(len(b) for b in got) == 2 * n
    raw = bytearray(open(tmp_path / "b_1.fq.gz", "rb").read()); raw[len(raw) // 2] ^= 0x5A
    open(tmp_path / "bad_1.fq.gz", "wb").write(bytes(raw))
    h = _open([str(tmp_path / "bad_1.fq.gz")], None, batch=3000); got, err = _drain(h); capi.lib().sq_reader_close(h)
    assert err is not None and "bad_1.fq.gz" in err and ("BGZF" in err or "record" in err)
    # [r4] (the members are inflated in groups of ~4 MB of text, each straight into its place) a file cut inside a member, and one cut between two members
    whole = open(tmp_path / "b_1.fq.gz", "rb").read()
    open(tmp_path / "cut_1.fq.gz", "wb").write(whole[: len(whole) * 2 // 3 + 7])
    h = _open([str(tmp_path / "cut_1.fq.gz")], None, batch=3000); got, err = _drain(h); capi.lib().sq_reader_close(h)
    assert err is not None and "cut_1.fq.gz" in err
    offs = []; q = 0
    while q < len(whole): ms = (whole[q + 16] | (whole[q + 17] << 8)) + 1; offs.append(q); q += ms          # 'BC' is the only extra field _bgzf_write makes: BSIZE sits at 16
    open(tmp_path / "half_1.fq.gz", "wb").write(whole[: offs[len(offs) // 2]])
    h = _open([str(tmp_path / "half_1.fq.gz")], None, batch=3000); got, err = _drain(h); capi.lib().sq_reader_close(h)
    assert err is None or "record" in err                                         # whole members: either every record is whole too, or the last one is reported as cut
    assert 0 < sum(len(b) for b in got) < n


def test_reader_streams_from_fifos_and_dev_fd(built, tmp_path):
    """Non-seekable inputs — named FIFOs and /dev/fd/N (process substitution, the documented way to feed salmon from a decompressor) — are
    read once, front to back, by the streaming path: nothing is probed, reopened or mapped."""
    import threading
    rng = np.random.default_rng(3)
    recs1 = ["".join(rng.choice(list("ACGT"), size=100)) for _ in range(1000)]; recs2 = ["".join(rng.choice(list("ACGT"), size=100)) for _ in range(1000)]
    _fq(tmp_path / "a_1.fq", recs1); _fq(tmp_path / "a_2.fq", recs2)
    # named FIFOs, one writer thread each
    f1, f2 = str(tmp_path / "p1.fq"), str(tmp_path / "p2.fq"); os.mkfifo(f1); os.mkfifo(f2)
    def feed(src, dst):
        with open(dst, "wb") as o: o.write(open(src, "rb").read())
    th = [threading.Thread(target=feed, args=(str(tmp_path / "a_1.fq"), f1)), threading.Thread(target=feed, args=(str(tmp_path / "a_2.fq"), f2))]
    for t in th: t.start()
    h = _open([f1], [f2], batch=300); got, err = _drain(h); capi.lib().sq_reader_close(h)
    for t in th: t.join(timeout=30)
    assert err is None
    assert [r for b in got for r in b] == [x.encode() for pair in zip(recs1, recs2) for x in pair]
    # /dev/fd/N of a pipe (what `<(zcat reads.fq.gz)` hands over)
    r, w = os.pipe()
    t = threading.Thread(target=lambda: (os.write(w, b""), [os.write(w, c) for c in _chunks(open(tmp_path / "a_1.fq", "rb").read())], os.close(w)))
    t.start()
    h = _open(["/dev/fd/%d" % r], None, batch=256); got, err = _drain(h); capi.lib().sq_reader_close(h)
    t.join(timeout=30); os.close(r)
    assert err is None and [x for b in got for x in b] == [x.encode() for x in recs1]


def _chunks(b, n=65536):
    return [b[i:i + n] for i in range(0, len(b), n)]


def test_reader_plain_gzip_is_inflated_in_pieces_by_the_pool(built, tmp_path, monkeypatch):
    # [r3] a gzip file of some size (not BGZF) is cut into pieces that find their own block starts (host/pgzip.cpp): same batches as the
    # plain files and as the one-thread zlib path, whatever the compression level, for several members in one file; damage is reported
    rng = np.random.default_rng(12); n = 120000
    def text(m):
        L = rng.integers(60, 151, n); b = rng.integers(0, 4, int(L.sum())).astype(np.uint8); seq = np.frombuffer(b"ACGT", np.uint8)[b].tobytes()
        q = (rng.integers(0, 40, int(L.sum())) + 33).astype(np.uint8).tobytes(); o = np.concatenate([[0], np.cumsum(L)])
        return b"".join(b"@r%d/%d some text\n%s\n+\n%s\n" % (i, m, seq[o[i]:o[i + 1]], q[o[i]:o[i + 1]]) for i in range(n))
    t1, t2 = text(1), text(2)
    open(tmp_path / "p_1.fq", "wb").write(t1); open(tmp_path / "p_2.fq", "wb").write(t2)
    open(tmp_path / "g_1.fq.gz", "wb").write(gzip.compress(t1, 6))                                                      # one member, default level
    cut = [0, 1000, 1000, len(t2) // 3, len(t2)]                                                                        # four members, one of them empty
    open(tmp_path / "g_2.fq.gz", "wb").write(b"".join(gzip.compress(t2[a:b], lvl) for a, b, lvl in zip(cut[:-1], cut[1:], (1, 6, 9, 4))))
    assert os.path.getsize(tmp_path / "g_1.fq.gz") > (8 << 20) and os.path.getsize(tmp_path / "g_2.fq.gz") > (8 << 20)   # above the size where the pieces start
    L = capi.lib()
    def run(f1, f2):
        h = _open([str(f1)], [str(f2)] if f2 else None, batch=50000); got, err = _drain(h); L.sq_reader_close(h); return got, err
    want, err = run(tmp_path / "p_1.fq", tmp_path / "p_2.fq"); assert err is None
    got, err = run(tmp_path / "g_1.fq.gz", tmp_path / "g_2.fq.gz")
    assert err is None and got == want and sum(len(b) for b in got) == 2 * n
    monkeypatch.setenv("SQ_READER_PGZ_MIN", str(1 << 40))          # the same files through one zlib stream each (what files below 8 MB take)
    got0, err = run(tmp_path / "g_1.fq.gz", tmp_path / "g_2.fq.gz"); assert err is None and got0 == want
    monkeypatch.delenv("SQ_READER_PGZ_MIN")
    raw = bytearray(open(tmp_path / "g_1.fq.gz", "rb").read()); raw[len(raw) // 2] ^= 0x5A
    open(tmp_path / "bad_1.fq.gz", "wb").write(bytes(raw))
    got, err = run(tmp_path / "bad_1.fq.gz", None)
    assert err is not None and "bad_1.fq.gz" in err
    open(tmp_path / "cut_1.fq.gz", "wb").write(bytes(raw[: len(raw) // 3]))
    got, err = run(tmp_path / "cut_1.fq.gz", None)
    assert err is not None and "cut_1.fq.gz" in err
