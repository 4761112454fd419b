"""Index builder (cDBG + contig table + SSHash-style dictionary) against brute force (CPU only)."""
import numpy as np
import os, tarfile, tempfile
import pytest
from salmon_amd import api, synth, capi
import orc

CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def enc(s):
    x = 0
    for i, c in enumerate(s):
        x |= CODE[c] << (2 * i)
    return x


def test_cdbg_invariants_and_dictionary_small(small_world):
    idx = small_world["idx"]
    assert orc.check_cdbg(idx) == 0          # tiles every reference, each k-mer once, unitigs maximal
    oi = small_world["oidx"]
    assert idx.num_kmers == orc.lib().orc_index_num_kmers(oi.h)
    v = idx.view(); k = idx.k
    U = idx.num_unitigs
    useq = np.ctypeslib.as_array(v.useq, shape=((v.total_unitig_nt + 31) // 32 + 1,))
    uoff = np.ctypeslib.as_array(v.uoff, shape=(U + 1,))
    rng = np.random.default_rng(0)
    for u in rng.choice(U, size=min(U, 400), replace=False):
        b, e = int(uoff[u]), int(uoff[u + 1])
        for p in range(b, e - k + 1, 3):
            x = 0
            for i in range(k):
                x |= ((int(useq[(p + i) >> 5]) >> (((p + i) & 31) * 2)) & 3) << (2 * i)
            assert idx.lookup_host(x) == (u, p - b, True)
            assert oi.lookup(x) == (u, p - b, True)
    # random (absent) k-mers agree with brute force too
    for x in rng.integers(0, 1 << 62, 3000):
        assert idx.lookup_host(int(x)) == oi.lookup(int(x))


def test_reverse_complement_lookup(small_world):
    idx = small_world["idx"]; tx = small_world["tx"]
    s = tx.seqs()[0].decode(); k = idx.k
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for p in range(0, min(len(s) - k, 300), 11):
        km = s[p:p + k]; rc = "".join(comp[c] for c in reversed(km))
        a, b = idx.lookup_host(enc(km)), idx.lookup_host(enc(rc))
        assert a is not None and b is not None
        assert a[0] == b[0] and a[1] == b[1] and a[2] != b[2]


def test_sample_data_index_roundtrip(built):
    src = "/root/reference/sample_data.tgz"
    if not os.path.exists(src):
        pytest.skip("reference sample data not present on this box")
    with tempfile.TemporaryDirectory() as d:
        tarfile.open(src).extractall(d)
        out = os.path.join(d, "idx")
        api.SalmonIndex.build(os.path.join(d, "sample_data", "transcripts.fasta"), out, threads=2)
        for f in ("index.bin", "info.json", "versionInfo.json", "duplicate_clusters.tsv"):
            assert os.path.exists(os.path.join(out, f))
        idx = api.SalmonIndex.load(out)
        assert idx.num_refs == 15 and idx.k == 31 and idx.m == 20      # SURVEY.md §4 fixtures; m = min(20, max(4, k-4))
        assert int(idx.ref_lens().sum()) <= 28562
        assert orc.check_cdbg(idx) == 0
        idx.free()


def test_bad_arguments_and_missing_index(built, tmp_path):
    with pytest.raises(capi.SalmonHipError):
        api.SalmonIndex.load(str(tmp_path / "nope"))               # SalmonIndex.hpp:124-131 throws on missing versionInfo.json
    fa = tmp_path / "t.fa"; fa.write_text(">a\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    with pytest.raises(capi.SalmonHipError):
        api.SalmonIndex.build(str(fa), str(tmp_path / "i"), k=30)  # k must be odd (BuildSalmonIndex.cpp:204)
    with pytest.raises(capi.SalmonHipError):
        api.SalmonIndex.build(str(fa), str(tmp_path / "i"), k=33)  # k <= 31 (:207)


def test_polya_clipping_duplicates_and_short_refs(built):
    body = "ACGTTGCATGCCGATAGCTAGCTAGGATCGATCGGGATATCGCGATTAGC" * 2
    names = ["t1", "t2_dup", "t3_polyA", "t4_short"]
    seqs = [body, body, body[:69] + "C" + "A" * 40, "ACGTACGTAC"]
    idx = api.SalmonIndex.build_mem(names, seqs, threads=1)
    assert idx.ref_names() == ["t1", "t3_polyA", "t4_short"]       # duplicate dropped (keepDuplicates=false)
    assert list(idx.ref_lens()) == [100, 70, 10] and list(idx.ref_complete_lens()) == [100, 110, 10]
    assert orc.check_cdbg(idx) == 0
    idx2 = api.SalmonIndex.build_mem(names, seqs, threads=1, keep_duplicates=True, no_clip=True)
    assert idx2.num_refs == 4 and list(idx2.ref_lens()) == [100, 100, 110, 10]
    assert orc.check_cdbg(idx2) == 0


def test_a_minimizer_with_many_occurrences_goes_through_the_skew_table(built, tmp_path):
    # [r6] one 20-mer (chosen for a tiny minimizer hash: it is the minimizer of every k-mer that contains it) in 90 different contexts = a bucket of more than
    # SQ_SKEW_THRESH occurrences: its k-mers are found through the skew table.  Every k-mer of every reference against the checker's own k-mer map, absent ones too;
    # one thread and eight build the same file (the slot records are laid out in three parallel steps)
    import hashlib, json
    rng = np.random.default_rng(3); X = "CTATGTTGATCCAAAAGCAA"
    R = lambda n: "".join("ACGT"[x] for x in rng.integers(0, 4, n))
    names = ["s%d" % i for i in range(90)] + ["r%d" % i for i in range(120)]; seqs = [R(60) + X + R(60) for _ in range(90)] + [R(400) for _ in range(120)]
    idx = api.SalmonIndex.build_mem(names, seqs, threads=8, outdir=str(tmp_path / "t8"))
    info = json.load(open(os.path.join(str(tmp_path / "t8"), "info.json")))
    assert info["max_bucket"] > 32 and info["num_skew_kmers"] > 500
    assert orc.check_cdbg(idx) == 0
    oi = orc.OrcIndex(idx); k = idx.k
    for s_ in seqs[:90:7] + seqs[90::17]:
        for p in range(len(s_) - k + 1):
            x = enc(s_[p:p + k]); got = idx.lookup_host(x)
            assert got is not None and got == oi.lookup(x)
    for x in rng.integers(0, 1 << 62, 2000): assert idx.lookup_host(int(x)) == oi.lookup(int(x))
    idx.free()
    api.SalmonIndex.build_mem(names, seqs, threads=1, outdir=str(tmp_path / "t1")).free()
    sha = lambda d: hashlib.sha256(open(os.path.join(str(tmp_path / d), "index.bin"), "rb").read()).hexdigest()
    assert sha("t1") == sha("t8")


def test_partitioned_kmer_table_builds_the_same_index(built, tmp_path, monkeypatch):
    # the builder's k-mer table under a tiny memory budget (dozens of hash partitions, SQ_INDEX_TABLE_GB) writes the same index file,
    # byte for byte, as the single pass; long unitigs (a decoy "chromosome") go through the 4 M-position pieces of the minimizer phase
    import hashlib
    rng = np.random.default_rng(4)
    tx = synth.Txome(seed=8, n_genes=60, iso_per_gene=4, threads=2)
    names = [n if isinstance(n, str) else n.decode() for n in tx.names()]; seqs = [s.decode() for s in tx.seqs()]
    chrom = "".join(rng.choice(list("ACGT"), 5_000_000)); chrom = chrom[:100000] + seqs[3] + chrom[100000:]
    def build(tag):
        d = str(tmp_path / tag)
        api.SalmonIndex.build_mem(names + ["chrD"], seqs + [chrom], threads=4, first_decoy=len(seqs), outdir=d).free()
        return hashlib.sha256(open(os.path.join(d, "index.bin"), "rb").read()).hexdigest()
    one = build("one")
    monkeypatch.setenv("SQ_INDEX_TABLE_GB", "0.004")      # 400 k slots per pass: ~17 passes over 5 M k-mers
    assert build("many") == one


def test_quant_sf_writer_digits_and_decoy_rows(built, tmp_path):
    # sq_write_quant_sf(_digits): rows for the targets only (decoys are dropped), TPM normalised over them, --sigDigits decimals
    names = ["t0", "t1", "t2", "decoyA"]; rng = np.random.default_rng(2)
    seqs = ["".join(rng.choice(list("ACGT"), n)) for n in (400, 500, 600, 3000)]
    idx = api.SalmonIndex.build_mem(names, seqs, threads=1, first_decoy=3)
    eff = np.array([250.123456, 350.5, 450.0, 1.0]); reads = np.array([10.0, 30.25, 0.0, 99.0])
    api.write_quant_sf(str(tmp_path / "a.sf"), idx, eff, reads); api.write_quant_sf(str(tmp_path / "b.sf"), idx, eff, reads, sig_digits=6)
    a = [l.split("\t") for l in open(tmp_path / "a.sf").read().splitlines()]; b = [l.split("\t") for l in open(tmp_path / "b.sf").read().splitlines()]
    assert a[0] == ["Name", "Length", "EffectiveLength", "TPM", "NumReads"] and [r[0] for r in a[1:]] == ["t0", "t1", "t2"]
    assert a[1][2] == "250.123" and b[1][2] == "250.123456" and a[2][4] == "30.250" and b[2][4] == "30.250000" and a[3][4] == "0.000"
    tpm = np.array([float(r[3]) for r in a[1:]]); w = reads[:3] / eff[:3]
    assert abs(tpm.sum() - 1e6) < 1.0 and np.allclose(tpm, 1e6 * w / w.sum(), rtol=1e-6) and [r[3] for r in a[1:]] == [r[3] for r in b[1:]]
    idx.free()
