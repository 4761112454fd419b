"""How long the device takes to inflate BGZF members of FASTQ text (hip/inflate_dev.hip, a wave per member): N members of 65280 bytes through
sq_debug_bgzf_inflate, meant to be run under `rocprofv3 --kernel-trace --stats` (the kernel's duration is the figure; the wall time here includes the copies).
Two kinds of text: the bench's (constant qualities, 100 bp) and one with random qualities; zlib levels 1 and 6."""
import ctypes as C, os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salmon_amd import capi


def fastq(rng, n, L, const_q):
    b = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, L))]; q = (rng.integers(0, 41, (n, L)) + 33).astype(np.uint8)
    return b"".join(b"@r\n%s\n+\n%s\n" % (b[i].tobytes(), b"I" * L if const_q else q[i].tobytes()) for i in range(n))


def main():
    L = capi.lib(); rng = np.random.default_rng(1); distinct, reps = 512, int(os.environ.get("INFLATE_REPS", "16"))
    only = sys.argv[1:] and tuple(int(x) for x in sys.argv[1].split(","))      # "0,1": random qualities, level 1 — for counter passes
    for const_q in (True, False):
        for level in (1, 6):
            if only and (int(const_q), level) != only: continue
            text = fastq(rng, distinct * 65280 // 207 + 400, 100, const_q)
            members = []; comp = b""
            for k in range(distinct):
                chunk = text[k * 65280:(k + 1) * 65280]; co = zlib.compressobj(level, zlib.DEFLATED, -15); body = co.compress(chunk) + co.flush()
                members.append((len(comp), len(body), zlib.crc32(chunk) & 0xFFFFFFFF)); comp += body
            desc = np.zeros(distinct * reps, np.dtype([("coff", "<u8"), ("voff", "<u8"), ("csize", "<u4"), ("isize", "<u4"), ("crc", "<u4"), ("pad", "<u4")]))
            for r in range(reps):
                for k, (co_, cs, crc) in enumerate(members):
                    j = r * distinct + k; desc[j] = (co_, j * 65280, cs, 65280, crc, 0)
            c = np.frombuffer(comp, np.uint8).copy(); nb = len(desc) * 65280; out = np.zeros(nb + 64, np.uint8); st = np.zeros(2, np.uint32)
            for _ in range(3):
                t0 = time.perf_counter()
                capi.check(L.sq_debug_bgzf_inflate(0, c.ctypes.data, len(c), desc.ctypes.data, len(desc), out.ctypes.data, nb, st.ctypes.data), "inflate")
                dt = time.perf_counter() - t0
            assert st[0] == 0xFFFFFFFF and out[:distinct * 65280].tobytes() == text[:distinct * 65280]
            print("const_q=%d level=%d: %d members, %.1f MB text, ratio %.3f, call %.1f ms (copies included)" % (const_q, level, len(desc), nb / 1e6, len(comp) * reps / nb, dt * 1e3), flush=True)


if __name__ == "__main__":
    main()
