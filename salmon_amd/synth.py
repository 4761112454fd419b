"""Binding for tools/_build/libsqsynth.so — seeded synthetic transcriptome / read generator."""
import ctypes as C
import os
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(_ROOT, "tools", "_build", "libsqsynth.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError("%s missing: run `python -m salmon_amd.build`" % _PATH)
        L = C.CDLL(_PATH)
        L.sqs_txome_generate.restype = C.c_void_p
        L.sqs_txome_generate.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.sqs_txome_free.argtypes = [C.c_void_p]
        L.sqs_txome_count.restype = C.c_uint32; L.sqs_txome_count.argtypes = [C.c_void_p]
        L.sqs_txome_name.restype = C.c_char_p; L.sqs_txome_name.argtypes = [C.c_void_p, C.c_uint32]
        L.sqs_txome_seq.restype = C.c_void_p; L.sqs_txome_seq.argtypes = [C.c_void_p, C.c_uint32]
        L.sqs_txome_len.restype = C.c_uint32; L.sqs_txome_len.argtypes = [C.c_void_p, C.c_uint32]
        L.sqs_txome_total_nt.restype = C.c_uint64; L.sqs_txome_total_nt.argtypes = [C.c_void_p]
        L.sqs_txome_tables.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32)]
        L.sqs_txome_write_fasta.restype = C.c_int; L.sqs_txome_write_fasta.argtypes = [C.c_void_p, C.c_char_p]
        L.sqs_reads_generate.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_double,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.sqs_genome_generate.restype = C.c_void_p
        L.sqs_genome_generate.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_uint32]
        L.sqs_genome_free.argtypes = [C.c_void_p]
        L.sqs_genome_count.restype = C.c_uint32; L.sqs_genome_count.argtypes = [C.c_void_p]
        L.sqs_genome_name.restype = C.c_char_p; L.sqs_genome_name.argtypes = [C.c_void_p, C.c_uint32]
        L.sqs_genome_seq.restype = C.c_void_p; L.sqs_genome_seq.argtypes = [C.c_void_p, C.c_uint32]
        L.sqs_genome_len.restype = C.c_uint64; L.sqs_genome_len.argtypes = [C.c_void_p, C.c_uint32]
        L.sqs_reads_generate_decoy.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_double,
                                               C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        _lib = L
    return _lib


class Txome:
    def __init__(self, seed=1, n_genes=200, iso_per_gene=10, threads=8):
        self.h = C.c_void_p(lib().sqs_txome_generate(seed, n_genes, iso_per_gene, threads))
        self.n = lib().sqs_txome_count(self.h)

    def free(self):
        if self.h:
            lib().sqs_txome_free(self.h); self.h = None

    def names(self):
        return [lib().sqs_txome_name(self.h, i).decode() for i in range(self.n)]

    def seqs(self):
        return [C.string_at(lib().sqs_txome_seq(self.h, i), lib().sqs_txome_len(self.h, i)) for i in range(self.n)]

    def total_nt(self):
        return int(lib().sqs_txome_total_nt(self.h))

    def tables(self):
        names = (C.c_char_p * self.n)(); seqs = (C.c_char_p * self.n)(); lens = (C.c_uint32 * self.n)()
        lib().sqs_txome_tables(self.h, names, seqs, lens)
        return names, seqs, lens

    def write_fasta(self, path):
        if lib().sqs_txome_write_fasta(self.h, path.encode()) != 0:
            raise IOError(path)

    def reads(self, n_pairs, read_len=100, seed=2, first_pair=0, sub_rate=0.005, indel_rate=0.0001, junk_frac=0.01, threads=8, truth=True):
        """Returns (seq uint8[2*n*read_len], seq_off uint64[2n+1], truth_tid, truth_pos)."""
        seq = np.empty(2 * n_pairs * read_len, np.uint8)
        tt = np.empty(n_pairs, np.uint32) if truth else None
        tp = np.empty(n_pairs, np.uint32) if truth else None
        lib().sqs_reads_generate(self.h, seed, first_pair, n_pairs, read_len, sub_rate, indel_rate, junk_frac, seq.ctypes.data,
                                 tt.ctypes.data if truth else None, tp.ctypes.data if truth else None, threads)
        off = np.arange(0, 2 * n_pairs + 1, dtype=np.uint64) * np.uint64(read_len)
        return seq, off, tt, tp


class Genome:
    """Synthetic decoy genome around a Txome (SURVEY.md 8d "G3G"): every gene's exons in order with introns, repeat families, random
    background; `n_chrom` chromosomes totalling ~total_nt."""

    def __init__(self, tx, seed=3, total_nt=30_000_000, n_chrom=3, repeat_frac=0.45, threads=8):
        self.h = C.c_void_p(lib().sqs_genome_generate(tx.h, seed, int(total_nt), n_chrom, repeat_frac, threads))
        self.n = lib().sqs_genome_count(self.h)

    def free(self):
        if self.h:
            lib().sqs_genome_free(self.h); self.h = None

    def names(self):
        return [lib().sqs_genome_name(self.h, i).decode() for i in range(self.n)]

    def seqs(self):
        return [C.string_at(lib().sqs_genome_seq(self.h, i), lib().sqs_genome_len(self.h, i)) for i in range(self.n)]

    def total_nt(self):
        return int(sum(lib().sqs_genome_len(self.h, i) for i in range(self.n)))

    def append_tables(self, tx):
        """Pointer tables for sq_index_build_mem: the transcripts followed by the chromosomes (the decoys)."""
        n = tx.n + self.n
        names = (C.c_char_p * n)(); seqs = (C.c_char_p * n)(); lens = (C.c_uint32 * n)()
        lib().sqs_txome_tables(tx.h, names, seqs, lens)
        self._keep = [lib().sqs_genome_name(self.h, i) for i in range(self.n)]
        for i in range(self.n):
            names[tx.n + i] = self._keep[i]
            seqs[tx.n + i] = C.cast(lib().sqs_genome_seq(self.h, i), C.c_char_p)
            lens[tx.n + i] = lib().sqs_genome_len(self.h, i)
        return names, seqs, lens

    def reads(self, tx, n_pairs, read_len=150, seed=2, first_pair=0, sub_rate=0.005, indel_rate=0.0001, junk_frac=0.01, genomic_frac=0.05,
              threads=8, truth=True):
        """Like Txome.reads, with `genomic_frac` of the pairs drawn from gene loci of the genome (truth_tid 0xFFFFFFFE)."""
        seq = np.empty(2 * n_pairs * read_len, np.uint8)
        tt = np.empty(n_pairs, np.uint32) if truth else None
        tp = np.empty(n_pairs, np.uint32) if truth else None
        lib().sqs_reads_generate_decoy(tx.h, self.h, seed, first_pair, n_pairs, read_len, sub_rate, indel_rate, junk_frac, genomic_frac,
                                       seq.ctypes.data, tt.ctypes.data if truth else None, tp.ctypes.data if truth else None, threads)
        off = np.arange(0, 2 * n_pairs + 1, dtype=np.uint64) * np.uint64(read_len)
        return seq, off, tt, tp
