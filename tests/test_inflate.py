"""[r5] The device BGZF inflater (hip/inflate_core.h, hip/inflate_dev.hip).  The decoder is one source for host and device: here (no GPU) its host build is
held to zlib on every kind of deflate stream a BGZF member can be — compression levels, stored blocks, fixed-Huffman blocks, dynamic blocks, several blocks per
stream, long matches at short distances, binary data, the empty stream — and must refuse damaged and truncated streams; with a GPU the kernel inflates whole
files' worth of members (a wave each) and its text and CRC verdicts equal zlib's."""
import ctypes as C, os, struct, zlib
import numpy as np
import pytest
from salmon_amd import capi


def _raw(data, level, strategy=zlib.Z_DEFAULT_STRATEGY, cuts=()):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy); out = b""; prev = 0
    for c in cuts: out += co.compress(data[prev:c]) + co.flush(zlib.Z_FULL_FLUSH); prev = c
    return out + co.compress(data[prev:]) + co.flush()


def _fastq(rng, n, L=100):
    b = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, L))]; q = (rng.integers(0, 41, (n, L)) + 33).astype(np.uint8)
    return b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b[i].tobytes(), q[i].tobytes() if i % 3 else b"I" * L) for i in range(n))


def _cases(rng):
    fq = _fastq(rng, 300); out = []
    for lvl in (1, 6, 9): out.append(("level %d" % lvl, fq[:65000], _raw(fq[:65000], lvl)))
    out.append(("stored", fq[:40000], _raw(fq[:40000], 0)))
    out.append(("fixed Huffman", fq[:3000], _raw(fq[:3000], 6, zlib.Z_FIXED)))
    out.append(("several blocks", fq[:60000], _raw(fq[:60000], 4, cuts=(100, 5000, 5001, 33000))))
    runs = b"".join(bytes(rng.integers(65, 91, int(p), dtype=np.uint8)) * int(rng.integers(1, 60)) for p in rng.integers(1, 40, 1500))[:65000]
    out.append(("tandem repeats", runs, _raw(runs, 6)))
    binary = rng.integers(0, 256, 65000, dtype=np.uint8).tobytes(); out.append(("binary", binary, _raw(binary, 6)))
    out.append(("empty", b"", _raw(b"", 6))); out.append(("one byte", b"A", _raw(b"A", 1)))
    return out


def test_decoder_core_equals_zlib_on_the_host(built):
    L = capi.lib(); rng = np.random.default_rng(3)
    for name, data, comp in _cases(rng):
        out = np.zeros(max(1, len(data)) + 16, np.uint8); crc = C.c_uint32(0); c = np.frombuffer(comp, np.uint8).copy()
        for shift in (0, 1, 2, 3):          # the bit reader takes aligned words: a stream may start at any byte
            buf = np.zeros(len(c) + 8, np.uint8); buf[shift:shift + len(c)] = c; out[:] = 0
            rc = L.sq_debug_inflate_core_host(buf.ctypes.data + shift, len(c), out.ctypes.data, len(data), C.byref(crc))
            assert rc == 0 and out[:len(data)].tobytes() == data and crc.value == (zlib.crc32(data) & 0xFFFFFFFF), (name, shift, rc)
        # a wrong size, a cut stream, a flipped bit: refused (or, for a flip, at least never a wrong text taken for right: the CRC differs)
        assert L.sq_debug_inflate_core_host(c.ctypes.data, len(c), out.ctypes.data, len(data) + 1, C.byref(crc)) != 0, name
        if len(c) > 8: assert L.sq_debug_inflate_core_host(c.ctypes.data, len(c) // 2, out.ctypes.data, len(data), C.byref(crc)) != 0, name
        if len(data) > 1000:
            for _ in range(20):
                d = c.copy(); d[int(rng.integers(0, len(d)))] ^= 1 << int(rng.integers(0, 8)); crc2 = C.c_uint32(0)
                rc = L.sq_debug_inflate_core_host(d.ctypes.data, len(d), out.ctypes.data, len(data), C.byref(crc2))
                assert rc != 0 or crc2.value != (zlib.crc32(data) & 0xFFFFFFFF) or out[:len(data)].tobytes() == data, name


def test_decoder_core_is_clean_under_asan_and_ubsan_on_damaged_streams(built):
    """tools/inflate_fuzz.cpp: 4000 streams (a third sound, the rest with flipped bits, cut, or with a wrong text size) through the shared decoder in exact-size
    heap buffers, AddressSanitizer + UBSan on: nothing read or written out of place, every sound stream decoded to its text."""
    import shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("make") is None: pytest.skip("needs hipcc and make")
    r = subprocess.run(["make", "-C", os.path.join(root, "tools"), "inflate_fuzz"], capture_output=True, text=True, timeout=900)
    if r.returncode != 0 and "sanitizer" in (r.stderr + r.stdout).lower() and "cannot find" in (r.stderr + r.stdout).lower(): pytest.skip("no sanitizer runtime for this compiler")
    assert r.returncode == 0 and " wrong 0" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_device_inflates_members_a_wave_each(built):
    L = capi.lib(); rng = np.random.default_rng(5); fq = _fastq(rng, 40000)              # ~8 MB of text: ~130 members
    members = []; comp = b""; voff = 0; want = b""
    for i in range(0, len(fq), 0xff00):
        chunk = fq[i:i + 0xff00]; body = _raw(chunk, 1 if (i // 0xff00) % 3 else 6)
        members.append((len(comp), voff, len(body), len(chunk), zlib.crc32(chunk) & 0xFFFFFFFF, 0)); comp += body; voff += len(chunk); want += chunk
    for name, data, body in _cases(rng):
        members.append((len(comp), voff, len(body), len(data), zlib.crc32(data) & 0xFFFFFFFF, 0)); comp += body; voff += len(data); want += data
    mem = np.array(members, np.dtype([("coff", "<u8"), ("voff", "<u8"), ("csize", "<u4"), ("isize", "<u4"), ("crc", "<u4"), ("pad", "<u4")]))
    c = np.frombuffer(comp, np.uint8).copy(); text = np.zeros(voff + 64, np.uint8); st = np.zeros(2, np.uint32)
    capi.check(L.sq_debug_bgzf_inflate(0, c.ctypes.data, len(c), mem.ctypes.data, len(mem), text.ctypes.data, voff, st.ctypes.data), "inflate")
    assert st[0] == 0xFFFFFFFF and text[:voff].tobytes() == want
    # one damaged member among many: named, with its cause; the others' text is unaffected
    bad = c.copy(); k = 57; bad[int(mem["coff"][k]) + int(mem["csize"][k]) // 2] ^= 0x10
    capi.check(L.sq_debug_bgzf_inflate(0, bad.ctypes.data, len(bad), mem.ctypes.data, len(mem), text.ctypes.data, voff, st.ctypes.data), "inflate")
    assert st[0] == k + 1 and st[1] != 0
    lo, hi = int(mem["voff"][k]), int(mem["voff"][k]) + int(mem["isize"][k]); assert text[:lo].tobytes() == want[:lo] and text[hi:voff].tobytes() == want[hi:]
    wrong = mem.copy(); wrong["crc"][3] ^= 1
    capi.check(L.sq_debug_bgzf_inflate(0, c.ctypes.data, len(c), wrong.ctypes.data, len(wrong), text.ctypes.data, voff, st.ctypes.data), "inflate")
    assert st[0] == 4 and st[1] == 8
