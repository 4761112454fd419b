#!/usr/bin/env python
"""Throughput of the host read pipeline alone (sq_reader): synthetic 2x100 bp FASTQ written to /dev/shm, drained batch by batch.
   python tools/reader_bench.py [pairs] [threads]"""
import ctypes as C, gzip, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from salmon_amd import capi

def write(path, n, seed, gz):
    r = np.random.default_rng(seed)
    bases = r.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, 100))
    recs = np.empty((n, 100 + 1 + 2 + 100 + 1), np.uint8)
    recs[:, :100] = bases; recs[:, 100] = 10; recs[:, 101] = ord("+"); recs[:, 102] = 10; recs[:, 103:203] = ord("I"); recs[:, 203] = 10
    body = recs.tobytes()
    heads = [b"@SRR000000.%d %d/1\n" % (i, i) for i in range(n)]
    out = bytearray()
    for i in range(n): out += heads[i]; out += body[i * 204:(i + 1) * 204]
    (gzip.open(path, "wb", compresslevel=4) if gz else open(path, "wb")).write(bytes(out))

def bgzf_write(path, data, block=60000):
    """bgzip-style file: gzip members of at most 64 KB with the 'BC' extra field (what sq_reader inflates in parallel)"""
    import struct, zlib
    with open(path, "wb") as f:
        for i in list(range(0, len(data), block)) + [None]:
            chunk = data[i:i + block] if i is not None else b""
            co = zlib.compressobj(4, zlib.DEFLATED, -15); cd = co.compress(chunk) + co.flush()
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(cd) + 8 - 1) + cd + struct.pack("<II", zlib.crc32(chunk), len(chunk)))


def drain(f1, f2, batch):
    L = capi.lib(); a1 = (C.c_char_p * 1)(f1.encode()); a2 = (C.c_char_p * 1)(f2.encode()); h = C.c_void_p()
    assert L.sq_reader_open(a1, 1, a2, 1, batch, 3, C.byref(h)) == 0, L.sq_last_error()
    t0 = time.perf_counter(); n = 0
    while True:
        rb = capi.ReadBatch(); s = C.c_int(-1)
        assert L.sq_reader_next(h, C.byref(rb), C.byref(s)) == 0, L.sq_last_error()
        if rb.n == 0: break
        n += rb.n; L.sq_reader_release(h, s.value)
    dt = time.perf_counter() - t0; L.sq_reader_close(h)
    return n, dt

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
    if len(sys.argv) > 2: os.environ["SQ_READER_THREADS"] = sys.argv[2]
    d = "/dev/shm/sq_reader_bench"; os.makedirs(d, exist_ok=True)
    for gz in (False, True):
        ext = ".fq.gz" if gz else ".fq"; f1, f2 = d + "/r_1" + ext, d + "/r_2" + ext
        if not os.path.exists(f1): write(f1, N, 1, gz); write(f2, N, 2, gz)
        for mode in ("fast", "safe") + (("zlib",) if gz else ()):   # gzip "fast" = pieces inflated by the pool (host/pgzip.cpp); "zlib" = one stream per file
            os.environ.pop("SQ_READER_SAFE", None); os.environ.pop("SQ_READER_PGZ_MIN", None)
            if mode == "safe": os.environ["SQ_READER_SAFE"] = "1"
            if mode == "zlib": os.environ["SQ_READER_PGZ_MIN"] = str(1 << 40)
            n, dt = drain(f1, f2, 1000000)
            print("%-5s %-4s %d pairs in %.3f s = %.2f M pairs/s (%d host threads, SQ_READER_THREADS=%s)" % ("gzip" if gz else "plain", mode, n, dt, n / dt / 1e6,
                os.cpu_count(), os.environ.get("SQ_READER_THREADS", "default")))
    os.environ.pop("SQ_READER_SAFE", None); os.environ.pop("SQ_READER_PGZ_MIN", None)
    b1, b2 = d + "/b_1.fq.gz", d + "/b_2.fq.gz"
    if not os.path.exists(b1): bgzf_write(b1, open(d + "/r_1.fq", "rb").read()); bgzf_write(b2, open(d + "/r_2.fq", "rb").read())
    n, dt = drain(b1, b2, 1000000); n, dt = drain(b1, b2, 1000000)
    print("bgzf  fast %d pairs in %.3f s = %.2f M pairs/s (members inflated on the worker pool)" % (n, dt, n / dt / 1e6))
