"""Generates tests/golden/edlib_infix_vectors.json.gz from the REFERENCE's own infix aligner
(/root/reference/src/edlib.cpp compiled by `make -C oracle ref` into oracle/_ref/libedlib_ref.so):
query / window / k  ->  (found, edit distance, startLocations[0], endLocations[0]) of
edlibAlign(q, t, {k, EDLIB_MODE_HW, EDLIB_TASK_LOC}) — the call behind --recoverOrphans.
Run in the build container (needs /root/reference); the JSON travels, the reference does not."""
import ctypes as C, json, os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libedlib_ref.so"))
ip = C.POINTER(C.c_int)
ref.ref_edlib_infix.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, ip, ip, ip, ip]
ALPHA = np.frombuffer(b"ACGTN", np.uint8)


def cases(seed=20260924, count=600):
    rng = np.random.default_rng(seed)
    for it in range(count):
        n = int(rng.integers(1, 257)); m = int(rng.integers(1, 1001)); kind = it % 6
        t = rng.integers(0, 4, m).astype(np.uint8)
        if kind == 0 or m < n:
            q = rng.integers(0, 4, n).astype(np.uint8)                      # unrelated: mostly "not found"
        else:
            s = int(rng.integers(0, m - n + 1)); q = t[s:s + n].copy()         # a copy of the window with edits
            for _ in range(int(rng.integers(0, max(1, n // 3)))):
                op = int(rng.integers(0, 3)); p = int(rng.integers(0, len(q)))
                if op == 0: q[p] = (q[p] + 1 + rng.integers(0, 3)) % 4
                elif op == 1 and len(q) > 1: q = np.delete(q, p)
                else: q = np.insert(q, p, rng.integers(0, 4))
            q = q[:256]
            if kind == 3 and len(q) > 2: q[int(rng.integers(0, len(q)))] = 4   # an N in the read matches nothing
            if kind == 4:                                                      # short tandem repeat: many optimal locations
                u = rng.integers(0, 4, 3).astype(np.uint8); t = np.tile(u, m // 3 + 1)[:m]
                q = np.tile(u, len(q) // 3 + 1)[:len(q)].copy(); q[len(q) // 2] = (q[len(q) // 2] + 1) % 4
            if kind == 5 and len(q) <= m: q = t[:len(q)].copy() if it % 2 else t[m - len(q):].copy()   # flush with a window edge
        yield q, t, len(q) // 4


out = []
for q, t, k in cases():
    v = [C.c_int() for _ in range(4)]
    qs, ts = ALPHA[q].tobytes(), ALPHA[t].tobytes()
    ok = ref.ref_edlib_infix(qs, len(q), ts, len(t), k, *[C.byref(x) for x in v])
    out.append(dict(q=qs.decode(), t=ts.decode(), k=k, found=int(ok), ed=v[0].value if ok else -1, start=v[1].value if ok else -1, end=v[2].value if ok else -1))
import gzip
path = os.path.join(ROOT, "tests", "golden", "edlib_infix_vectors.json.gz")
with gzip.GzipFile(path, "wb", mtime=0) as f:
    f.write(json.dumps(out, separators=(",", ":")).encode())
print(len(out), "vectors,", sum(c["found"] for c in out), "found ->", path, os.path.getsize(path), "bytes")
