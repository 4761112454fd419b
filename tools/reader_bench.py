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
        for mode in ("fast", "safe"):
            if mode == "safe": os.environ["SQ_READER_SAFE"] = "1"
            else: os.environ.pop("SQ_READER_SAFE", None)
            n, dt = drain(f1, f2, 1000000)
            print("%-5s %-4s %d pairs in %.3f s = %.2f M pairs/s (%d host threads, SQ_READER_THREADS=%s)" % ("gzip" if gz else "plain", mode, n, dt, n / dt / 1e6,
                os.cpu_count(), os.environ.get("SQ_READER_THREADS", "default")))
