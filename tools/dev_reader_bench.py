#!/usr/bin/env python
"""Throughput of the device FASTQ reader alone (hip/fastq_dev.hip behind sq_reader): two plain 2x100 bp files in /dev/shm, drained batch by batch
(no mapping), under several settings of its knobs.  SQ_READER_STATS=1 makes the reader print its own stage times at close.
   python tools/dev_reader_bench.py [pairs] [header bytes]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from salmon_amd import capi


def write(path, n, seed, hdr):
    r = np.random.default_rng(seed); L = 100
    row = np.empty((n, hdr + 1 + L + 3 + L + 1), np.uint8)
    row[:, 0] = ord("@"); row[:, 1:hdr] = ord("r"); row[:, hdr] = 10
    row[:, hdr + 1:hdr + 1 + L] = r.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, L))
    row[:, hdr + 1 + L:hdr + 4 + L] = np.frombuffer(b"\n+\n", np.uint8); row[:, hdr + 4 + L:hdr + 4 + 2 * L] = ord("I"); row[:, -1] = 10
    row.tofile(path); return row.shape[1]


def drain(f1, f2, batch, slots):
    L = capi.lib(); a1 = (C.c_char_p * 1)(f1.encode()); a2 = (C.c_char_p * 1)(f2.encode()); h = C.c_void_p()
    t0 = time.perf_counter()
    capi.check(L.sq_reader_open(a1, 1, a2, 1, batch, slots, C.byref(h)), "sq_reader_open")
    n = 0; dev = True
    while True:
        rb = capi.ReadBatch(); s = C.c_int(-1)
        capi.check(L.sq_reader_next(h, C.byref(rb), C.byref(s)), "sq_reader_next")
        if rb.n == 0: break
        n += rb.n; dev = dev and bool(rb.on_device); L.sq_reader_release(h, s.value)
    dt = time.perf_counter() - t0
    L.sq_reader_close(h)
    return n, dt, dev


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000000
    hdr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    d = "/dev/shm/sq_dev_reader_bench"; os.makedirs(d, exist_ok=True)
    f1, f2 = d + "/r_1.fq", d + "/r_2.fq"
    rb = write(f1, N, 1, hdr); write(f2, N, 2, hdr)
    os.environ["SQ_READER_STATS"] = "1"
    print("%d pairs, %d bytes of text per pair" % (N, 2 * rb), flush=True)
    for batch, slots, thr in ((1000000, 4, 16), (1000000, 4, 16), (1000000, 4, 8), (1000000, 4, 32), (1000000, 3, 16), (2000000, 4, 16), (250000, 4, 16), (5000000, 3, 16)):
        os.environ["SQ_READER_THREADS"] = str(thr)
        n, dt, dev = drain(f1, f2, batch, slots)
        print("batch %7d slots %d threads %2d: %d pairs in %.3f s = %6.1f M pairs/s, %5.1f GB/s of text (device path: %s)" % (batch, slots, thr, n, dt, n / dt / 1e6, n * 2 * rb / dt / 1e9, dev), flush=True)
    for f in (f1, f2): os.remove(f)
