"""Experiment (GPU box): the device gzip decoder alone on FASTQ text — sq_debug_gzip_inflate on a file image in host memory, SQ_READER_STATS phases on stderr."""
import ctypes as C, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from salmon_amd import capi
os.environ["SQ_READER_STATS"] = "1"
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 400; constq = len(sys.argv) > 2 and sys.argv[2] == 'const'; rng = np.random.default_rng(1); L = capi.lib()
n = mb * 1000000 // 207; b = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, 100))]; q = (rng.integers(0, 41, (n, 100)) + 33).astype(np.uint8)
rec = np.zeros((n, 207), np.uint8); rec[:, 0] = ord("@"); rec[:, 1:4] = np.frombuffer(b"r1\n", np.uint8); rec[:, 4:104] = b; rec[:, 104] = 10; rec[:, 105] = ord("+"); rec[:, 106] = 10; rec[:, 107:207] = (ord('F') if constq else q); rec[:, 206] = 10
text = rec.tobytes(); t0 = time.time(); co = zlib.compressobj(6, zlib.DEFLATED, 31); gz = co.compress(text) + co.flush(); print("text %.1f MB, gzip -6 %.1f MB (%.1f s to compress)" % (len(text) / 1e6, len(gz) / 1e6, time.time() - t0), flush=True)
buf = np.frombuffer(gz, np.uint8).copy(); out = np.zeros(len(text) + 1024, np.uint8); nn = C.c_uint64(); ctr = (C.c_uint64 * 4)()
for rep in range(3):
    t0 = time.time(); rc = L.sq_debug_gzip_inflate(0, buf.ctypes.data, len(gz), 0, out.ctypes.data, len(out), C.byref(nn), ctr); dt = time.time() - t0
    print("rc %d  %.3f s  %.2f GB/s of text  (segments %d spans %d) %s" % (rc, dt, nn.value / dt / 1e9, ctr[0], ctr[1], "" if out[:nn.value].tobytes() == text else "TEXT DIFFERS"), flush=True)
