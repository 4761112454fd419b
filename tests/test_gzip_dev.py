"""[r6] An ordinary gzip file inflated on the device (hip/gzip_dev.hip; the decoder's source is hip/inflate_core.h, one source for host and device).
Without a GPU: the host build of the two new pieces — the search for a block start and the decoding of a span into 16-bit symbols — is held to zlib: every span ends
exactly where the next was found to start, and the symbols, resolved with the 32 KB in front of them, are the text.  With a GPU: whole files through the device path
(search, spans, window chain, translation, CRC) equal zlib's text — compression levels, constant and random qualities, several members, an empty member, stored blocks,
segments of 1 MB (spans, windows and members cross segment borders) — and damaged, truncated and wrongly-summed files are refused."""
import ctypes as C, gzip, struct, zlib
import numpy as np
import pytest
from salmon_amd import capi

NONE = 2**64 - 1


def _fastq(rng, n, L=100, const_q=False):
    b = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, L))]; q = (rng.integers(0, 41, (n, L)) + 33).astype(np.uint8)
    return b"".join(b"@read.%d some/1\n%s\n+\n%s\n" % (i, b[i].tobytes(), b"F" * L if const_q else q[i].tobytes()) for i in range(n))


def _gz(data, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, 31); return co.compress(data) + co.flush()


def test_spans_and_block_search_on_the_host(built):
    L = capi.lib(); rng = np.random.default_rng(3); text = _fastq(rng, 3000); gz = _gz(text); n = len(gz)
    buf = np.frombuffer(gz + b"\0" * 64, np.uint8).copy()
    starts = [80]; lo = 81
    while True:
        f = L.sq_debug_find_block_start_host(buf.ctypes.data, n, lo, n * 8)
        if f == NONE: break
        starts.append(int(f)); lo = int(f) + 1
    assert len(starts) >= 8                                        # ~16 KB of compressed bytes per block
    win = np.zeros(32768, np.uint8); out = bytearray(); marked = 0
    for i, s in enumerate(starts):
        stop = starts[i + 1] if i + 1 < len(starts) else NONE
        sym = np.zeros(1 << 20, np.uint16); ns = C.c_uint32(); eb = C.c_uint64(); fin = C.c_uint32()
        rc = L.sq_debug_inflate_span_host(buf.ctypes.data, n, s, stop, sym.ctypes.data, len(sym), C.byref(ns), C.byref(eb), C.byref(fin))
        assert rc == 0 and ((stop == NONE and fin.value == 1) or eb.value == stop), (i, rc, eb.value, stop)
        sy = sym[:ns.value]; m = (sy & 0x8000) != 0; marked += int(m.sum())
        res = np.where(m, win[sy & 0x7FFF], (sy & 0xFF).astype(np.uint8)).astype(np.uint8); out += res.tobytes(); win = np.concatenate([win, res])[-32768:]
    assert bytes(out) == text and marked > 1000                     # copies out of the unknown window did occur and were resolved
    # a start that is no block boundary: the span in front of it must say so
    sym = np.zeros(1 << 20, np.uint16); ns = C.c_uint32(); eb = C.c_uint64(); fin = C.c_uint32()
    assert L.sq_debug_inflate_span_host(buf.ctypes.data, n, starts[1], starts[2] + 3, sym.ctypes.data, len(sym), C.byref(ns), C.byref(eb), C.byref(fin)) == 9
    # a span whose buffer is too small
    assert L.sq_debug_inflate_span_host(buf.ctypes.data, n, starts[1], starts[2], sym.ctypes.data, 1000, C.byref(ns), C.byref(eb), C.byref(fin)) == 7


def _dev(gz, cap, seg=0):
    L = capi.lib(); buf = np.frombuffer(gz, np.uint8).copy(); out = np.zeros(cap + 64, np.uint8); n = C.c_uint64(); ctr = (C.c_uint64 * 4)()
    rc = L.sq_debug_gzip_inflate(0, buf.ctypes.data, len(gz), seg, out.ctypes.data, cap, C.byref(n), ctr)
    return rc, out[:n.value].tobytes(), list(ctr), (L.sq_last_error().decode() if rc else "")


@pytest.mark.gpu
def test_whole_files_through_the_device_decoder(built):
    rng = np.random.default_rng(9)
    fq = _fastq(rng, 60000); fqc = _fastq(rng, 60000, const_q=True)      # ~13 MB of text each
    cases = [("level 6, random qualities", fq, _gz(fq, 6)), ("level 1", fq, _gz(fq, 1)), ("level 9", fq, _gz(fq, 9)), ("constant qualities", fqc, _gz(fqc, 6)),
             ("three members", fq, _gz(fq[:4000000]) + _gz(fq[4000000:4000000]) + _gz(fq[4000000:9000000]) + _gz(fq[9000000:])),
             ("stored blocks between dynamic ones", fq[:3000000], _gz(fq[:1000000], 6) + _gz(fq[1000000:2000000], 0) + _gz(fq[2000000:3000000], 6)),
             ("tiny file", fq[:300], _gz(fq[:300])), ("python gzip module (file name in the header)", fq[:2000000], gzip.compress(fq[:2000000], 5))]
    for name, text, gz in cases:
        for seg in (0, 1 << 20):
            rc, got, ctr, err = _dev(gz, len(text) + 1024, seg)
            assert rc == 0, (name, seg, err)
            assert got == text, (name, seg, len(got), len(text))
            if seg and len(gz) > (3 << 20): assert ctr[0] >= 3 and ctr[1] > 50, (name, ctr)        # several segments, many spans
    # damage: a flipped bit in the middle, a wrong checksum, a wrong length, a cut file
    gz = bytearray(cases[0][2])
    bad = bytearray(gz); bad[len(bad) // 2] ^= 0x10; rc, _, _, err = _dev(bytes(bad), len(fq) + 1024); assert rc != 0 and err, err
    bad = bytearray(gz); bad[-8] ^= 1; rc, _, _, err = _dev(bytes(bad), len(fq) + 1024); assert rc != 0 and "checksum" in err, err
    bad = bytearray(gz); bad[-4] ^= 1; rc, _, _, err = _dev(bytes(bad), len(fq) + 1024); assert rc != 0 and "length" in err, err
    rc, _, _, err = _dev(bytes(gz[: len(gz) * 2 // 3]), len(fq) + 1024); assert rc != 0 and ("truncated" in err or "ends" in err), err
    rc, _, _, err = _dev(b"@r1\nACGT\n+\nFFFF\n" * 100, 10000); assert rc != 0 and "gzip member" in err
