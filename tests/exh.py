"""ctypes binding of oracle/_build/libexhaustive.so — TEST INFRASTRUCTURE: the exhaustive all-positions aligner (oracle/exhaustive.cpp)
that cross-checks the seed-chain-extend rows a1-a4.  Shares nothing with the checker (oracle.cpp) or the product."""
import ctypes as C, os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(ROOT, "oracle", "_build", "libexhaustive.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(_PATH)
        L.exh_labels.restype = C.c_int
        L.exh_labels.argtypes = [C.c_uint32, C.POINTER(C.c_char_p), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_double, C.c_uint32, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        _lib = L
    return _lib


def labels(seqs, reads, off, n_pairs, opts, threads=8):
    """seqs: list of bytes (transcripts, index order); reads/off: interleaved pairs as sq_read_batch.  -> (lab_off, lab_tid, lab_score, kind)."""
    n = len(seqs)
    arr = (C.c_char_p * n)(*seqs); lens = np.array([len(s) for s in seqs], np.uint32)
    reads = np.ascontiguousarray(reads, np.uint8); off = np.ascontiguousarray(off, np.uint64)
    cap = 64 * n_pairs + 1024
    lab_off = np.zeros(n_pairs + 1, np.uint64); lab_tid = np.zeros(cap, np.uint32); lab_score = np.zeros(cap, np.int32); kind = np.zeros(n_pairs, np.uint8)
    rc = lib().exh_labels(n, arr, lens.ctypes.data, n_pairs, reads.ctypes.data, off.ctypes.data, opts.match_score, opts.mismatch_penalty, opts.gap_open,
                          opts.gap_extend, opts.min_score_fraction, opts.frag_len_max, int(opts.allow_orphans), int(opts.allow_dovetail), opts.score_exp,
                          opts.min_aln_prob, threads, lab_off.ctypes.data, lab_tid.ctypes.data, lab_score.ctypes.data, cap, kind.ctypes.data)
    if rc != 0: raise RuntimeError("exh_labels: label buffer too small")
    w = int(lab_off[-1])
    return lab_off, lab_tid[:w], lab_score[:w], kind


def compare(lab_off, lab_tid, read_off, aln_tid, max_examples=40):
    """Per-fragment label sets of the exhaustive aligner vs a mapper's alignment lists -> dict of counts by disagreement class."""
    n = len(lab_off) - 1
    out = dict(n=n, equal=0, both_unmapped=0, heuristic_subset=0, heuristic_superset=0, other=0, heuristic_unmapped=0, exhaustive_unmapped=0, examples=[])
    for f in range(n):
        a = set(int(x) for x in lab_tid[int(lab_off[f]):int(lab_off[f + 1])]); b = set(int(x) for x in aln_tid[int(read_off[f]):int(read_off[f + 1])])
        if a == b:
            out["equal"] += 1
            if not a: out["both_unmapped"] += 1
            continue
        if not b: k = "heuristic_unmapped"
        elif not a: k = "exhaustive_unmapped"
        elif b < a: k = "heuristic_subset"
        elif a < b: k = "heuristic_superset"
        else: k = "other"
        out[k] += 1
        if len(out["examples"]) < max_examples: out["examples"].append((f, k, sorted(a), sorted(b)))
    out["agreement"] = out["equal"] / max(1, n)
    return out
