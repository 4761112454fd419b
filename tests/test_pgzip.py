"""host/pgzip.cpp (a plain gzip file inflated in pieces by several threads) against zlib, byte for byte, over what gzip files look like:
compression levels, sync / full flush points (pigz output is made of them), several members, stored blocks, text that is not FASTQ, binary
data (no piece finds a block start: the first one decodes everything), small pieces (many hand-overs), both read interfaces; damaged and
truncated files must be reported.  tools/pgz_check.cpp is the driver.  No GPU."""
import gzip, os, random, subprocess, zlib
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "_build", "pgz_check")


def _fastq(rng, n, L=100):
    b = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, L))]; q = (rng.integers(0, 41, (n, L)) + 33).astype(np.uint8)
    return b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b[i].tobytes(), q[i].tobytes()) for i in range(n))


def _flushed(data, level, cuts, mode):
    co = zlib.compressobj(level, zlib.DEFLATED, 31); out = b""; prev = 0
    for c in cuts: out += co.compress(data[prev:c]) + co.flush(mode); prev = c
    return out + co.compress(data[prev:]) + co.flush()


def _run(path, threads, piece, zero_copy=False):
    return subprocess.run([EXE, str(path), str(threads), str(piece)] + (["z"] if zero_copy else []), capture_output=True, text=True, timeout=300)


def test_pieces_equal_zlib_on_every_kind_of_gzip_file(built, tmp_path):
    assert os.path.exists(EXE), "tools/_build/pgz_check is built by build.build_tools()"
    rng = np.random.default_rng(31); random.seed(9)
    fq = _fastq(rng, 30000); f = tmp_path / "x.gz"
    cases = [("level %d" % l, gzip.compress(fq, l)) for l in (1, 6, 9)]
    cases.append(("stored blocks", gzip.compress(fq[:3_000_000], 0)))
    cases.append(("sync flushes", _flushed(fq, 6, sorted(random.sample(range(1, len(fq)), 120)), zlib.Z_SYNC_FLUSH)))
    cases.append(("full flushes", _flushed(fq, 4, sorted(random.sample(range(1, len(fq)), 40)), zlib.Z_FULL_FLUSH)))
    cuts = [0, 1000, 1000, len(fq) // 3, len(fq)]
    cases.append(("four members, one empty", b"".join(gzip.compress(fq[a:b], l) for a, b, l in zip(cuts[:-1], cuts[1:], (1, 6, 9, 4)))))
    cases.append(("trailing zeros", gzip.compress(fq, 6) + b"\0" * 64))
    cases.append(("repetitive text", gzip.compress(b"the quick brown fox jumps over the lazy dog\n" * 150000, 6)))
    cases.append(("binary data", gzip.compress(rng.integers(0, 256, 2_000_000, dtype=np.uint8).tobytes(), 6)))
    # [r5] long matches at short distances (a constant quality string, homopolymers, tandem repeats of every period below the vector width): the inflaters
    # lay the period out once and store it in strides
    runs = b"".join(bytes(rng.integers(65, 91, int(per), dtype=np.uint8)) * int(rng.integers(1, 40)) + b"\n" for per in rng.integers(1, 40, 60000))
    cases.append(("tandem repeats, periods 1..39", gzip.compress(runs, 6)))
    cases.append(("constant qualities", gzip.compress(b"".join(b"@r\n%s\n+\n%s\n" % (bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 100)]), b"I" * 100) for _ in range(20000)), 1)))
    for name, z in cases:
        open(f, "wb").write(z)
        for threads, piece, zc in ((4, 65536, False), (3, 250000, True), (8, 1 << 20, False)):
            r = _run(f, threads, piece, zc)
            assert r.returncode == 0 and "equal=1" in r.stdout, (name, threads, piece, r.stdout[-300:], r.stderr[-300:])
    # several pieces really were used, and every piece of the chain started where its predecessor ended
    open(f, "wb").write(gzip.compress(fq, 6)); r = _run(f, 4, 65536)
    assert int(r.stdout.split("pieces=")[1].split()[0]) > 20


def test_damage_and_truncation_are_reported(built, tmp_path):
    rng = np.random.default_rng(32); z = bytearray(gzip.compress(_fastq(rng, 20000), 6)); f = tmp_path / "bad.gz"
    bad = bytearray(z); bad[len(bad) // 2] ^= 0x5A; open(f, "wb").write(bytes(bad))
    r = _run(f, 4, 65536); assert r.returncode != 0 and "pgz error" in r.stdout, r.stdout
    open(f, "wb").write(bytes(z[: len(z) * 2 // 3]))
    r = _run(f, 4, 65536); assert r.returncode != 0 and "pgz error" in r.stdout, r.stdout
    bad = bytearray(z); bad[-6] ^= 0xFF; open(f, "wb").write(bytes(bad))          # the stored CRC-32
    r = _run(f, 4, 65536); assert r.returncode != 0 and "checksum" in r.stdout, r.stdout


def test_truncated_literal_only_stream_fails_fast_and_small(built, tmp_path):
    """[r4, ADVICE r3] A Z_HUFFMAN_ONLY stream has no matches, so a block is literals only; past the end of a truncated file the bit reader yields
    zeros and the all-zero code may be a literal — the decoder must notice the end of the input on the literal path too, not grow its output until
    memory runs out.  Run under a 2 GB address-space limit and a short timeout."""
    import resource
    rng = np.random.default_rng(33); fq = _fastq(rng, 40000)
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_HUFFMAN_ONLY); z = co.compress(fq) + co.flush(); f = tmp_path / "h.gz"
    open(f, "wb").write(z); r = _run(f, 4, 65536)
    assert r.returncode == 0 and "equal=1" in r.stdout, r.stdout[-300:]
    def limit(): resource.setrlimit(resource.RLIMIT_AS, (2 << 30, 2 << 30))
    import time
    for frac in (0.4, 0.77, 0.999):
        open(f, "wb").write(z[: int(len(z) * frac)])
        t0 = time.time()
        r = subprocess.run([EXE, str(f), "4", "65536"], capture_output=True, text=True, timeout=60, preexec_fn=limit)
        assert r.returncode != 0 and "pgz error" in r.stdout and "memory" not in r.stdout, (frac, r.stdout[-300:], r.stderr[-300:])
        assert time.time() - t0 < 20, "a truncated file must be reported at once"
