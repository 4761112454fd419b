"""[r4] Seam B4, the files downstream tools read beside quant.sf: aux_info/meta_info.json with the reference's key set (the list is parsed out of
GZipWriter::writeMeta by tests/golden/make_meta_keys.py), the index digests that identify the transcriptome (SalmonIndex.hpp:94-98), fld.gz, the
legacy bias vectors and the binary model dumps."""
import ctypes as C, gzip, hashlib, json, os, struct, subprocess
import numpy as np
import pytest
from salmon_amd import api, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "meta_info_keys.json")))


def test_committed_key_list_is_what_the_reference_source_says():
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "src", "output", "GZipWriter.cpp")):
        pytest.skip("no reference tree here (GPU box): the committed list stands")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_meta_keys", os.path.join(ROOT, "tests", "golden", "make_meta_keys.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    assert m.parse(ref) == GOLD


def _meta(**kw):
    libs = (C.c_char_p * 1)(b"IU"); lq = (C.c_uint32 * 4)(500, 1000, 2000, 4000)
    m = capi.MetaInfo(samp_type=b"none", opt_type=b"vb", num_libraries=1, library_types=libs, frag_dist_length=1001, frag_length_mean=250.5, frag_length_sd=25.25,
                      num_bias_bins=4096, mapping_type=b"mapping", keep_duplicates=0, range_factorized=1, num_valid_targets=7, num_decoy_targets=2, num_eq_classes=11,
                      num_length_classes=4, length_classes=lq, index_seq_hash=b"ab" * 32, index_name_hash=b"cd" * 32, index_seq_hash512=b"ef" * 64, index_name_hash512=b"01" * 64,
                      index_decoy_seq_hash=b"23" * 32, index_decoy_name_hash=b"45" * 32, num_processed=100, num_mapped=90, percent_mapped=90.0,
                      start_time=b"Thu Sep 24 12:00:00 2026", end_time=b"Thu Sep 24 12:00:01 2026")
    for k, v in kw.items(): setattr(m, k, v)
    m._keep = (libs, lq)
    return m


def test_meta_info_has_the_reference_keys_in_the_reference_order(built, tmp_path):
    p = str(tmp_path / "meta_info.json")
    capi.check(capi.lib().sq_write_meta_info(p.encode(), C.byref(_meta(backend=b"salmon-hip test", num_em_iterations=123))), "sq_write_meta_info")
    d = json.load(open(p))
    keys = [k for k in d if k != "salmon_hip"]
    assert keys == GOLD["keys"]                                                   # same names, same order; the extras sit under their own object, last
    assert list(d)[-1] == "salmon_hip" and d["salmon_hip"]["num_em_iterations"] == 123
    assert d["eq_class_properties"] == ["range_factorized", "gzipped"] and d["length_classes"] == [500, 1000, 2000, 4000] and d["library_types"] == ["IU"]
    assert d["quant_errors"] == [] and d["keep_duplicates"] is False and d["seq_bias_correct"] is False and d["call"] == "quant"
    assert d["index_seq_hash"] == "ab" * 32 and d["index_name_hash512"] == "01" * 64 and d["frag_length_mean"] == 250.5
    # the UNKNOWN duplicate status writes no key (GZipWriter.cpp:518-529); an error list has one entry (writeEmptyMeta)
    capi.check(capi.lib().sq_write_meta_info(p.encode(), C.byref(_meta(keep_duplicates=-1, quant_errors=b"insufficient_assigned_fragments", scalar_weights=1))), "sq_write_meta_info")
    d = json.load(open(p))
    assert "keep_duplicates" not in d and d["quant_errors"] == ["insufficient_assigned_fragments"] and d["eq_class_properties"] == ["range_factorized", "scalar_weights", "gzipped"]
    assert "salmon_hip" not in d


def test_index_digests_are_sha2_of_the_input_records(built, tmp_path):
    rng = np.random.default_rng(5)
    seqs = ["".join(rng.choice(list("ACGT"), n)) for n in (400, 650, 900, 1200)] + ["".join(rng.choice(list("acgtn"), 3000))]   # the decoy in lower case with Ns
    seqs[1] = seqs[1][:-40] + "A" * 40                                            # a poly-A tail that the builder clips: the digest is of the record as given
    names = ["t1|gene1|x", "t2|gene2|y", "t3", "t4", "chrD"]
    for gencode in (0, 1):
        o = capi.IndexOpts(31, 0, 0, 0, 2, gencode)
        nm = (C.c_char_p * 5)(*[n.encode() for n in names]); sq = (C.c_char_p * 5)(*[s.encode() for s in seqs]); ln = (C.c_uint32 * 5)(*[len(s) for s in seqs])
        out = C.c_void_p(); d = str(tmp_path / ("idx%d" % gencode))
        capi.check(capi.lib().sq_index_build_mem(C.byref(o), 5, nm, sq, ln, 4, d.encode(), C.byref(out)), "sq_index_build_mem")
        cut = (lambda n: n.split("|")[0]) if gencode else (lambda n: n)
        want = [hashlib.sha256("".join(seqs[:4]).encode()).hexdigest(), hashlib.sha256("".join(cut(n) for n in names[:4]).encode()).hexdigest(),
                hashlib.sha512("".join(seqs[:4]).encode()).hexdigest(), hashlib.sha512("".join(cut(n) for n in names[:4]).encode()).hexdigest(),
                hashlib.sha256(seqs[4].encode()).hexdigest(), hashlib.sha256(names[4].encode()).hexdigest()]
        idx = api.SalmonIndex(out.value)
        assert [capi.lib().sq_index_hash(idx.h, i).decode() for i in range(6)] == want
        info = json.load(open(os.path.join(d, "info.json")))
        assert [info[k] for k in ("SeqHash", "NameHash", "SeqHash512", "NameHash512", "DecoySeqHash", "DecoyNameHash")] == want and info["keep_duplicates"] is False
        idx2 = api.SalmonIndex.load(d)                                            # the digests come back from info.json, where the reference keeps them
        assert [capi.lib().sq_index_hash(idx2.h, i).decode() for i in range(6)] == want and capi.lib().sq_index_keeps_duplicates(idx2.h) == 0
        idx.free(); idx2.free()


def test_fld_samples_follow_samplesFromLogPMF(built, tmp_path):
    x = np.arange(1001, dtype=np.float64); lp = -0.5 * ((x - 260.0) / 30.0) ** 2; lp -= np.log(np.exp(lp).sum())
    p = str(tmp_path / "fld.gz"); mean, sd, sup = C.c_double(), C.c_double(), C.c_uint32()
    capi.check(capi.lib().sq_write_fld_samples(p.encode(), lp.ctypes.data, 120, 1000, 10000, 7, C.byref(mean), C.byref(sd), C.byref(sup)), "sq_write_fld_samples")
    h = np.frombuffer(gzip.open(p).read(), np.int32)
    assert len(h) == 1001 == sup.value and h.sum() == 10000 and h[:120].sum() == 0 and h[1000] == 0          # bins [minVal, maxVal) carry mass (DistributionUtils.cpp:76)
    # the summary, restated: renormalise over [minVal, maxVal], mean = exp(logsum log(i) + logp_i), var = sum p_i i^2 - mean^2 over i in [minVal, maxVal)
    q = lp[120:1001] - np.log(np.exp(lp[120:1001]).sum()); i = np.arange(120, 1000)
    m = np.exp(np.logaddexp.reduce(np.log(i) + q[:-1])); s = np.sqrt((np.exp(q[:-1]) * i * i).sum() - m * m)
    assert abs(mean.value - m) < 1e-9 * m and abs(sd.value - s) < 1e-7 * s and abs(mean.value - 260.0) < 0.5
    assert abs((h * np.arange(1001)).sum() / 10000.0 - 260.0) < 1.5                                          # the draws follow the distribution
    capi.check(capi.lib().sq_write_fld_samples(p.encode(), lp.ctypes.data, 120, 1000, 10000, 7, None, None, None), "again")
    assert np.array_equal(np.frombuffer(gzip.open(p).read(), np.int32), h)                                   # a function of the seed, not of the run


def test_legacy_bias_vectors_and_model_dump_layouts(built, tmp_path):
    n = C.c_uint32(); capi.check(capi.lib().sq_write_legacy_bias(str(tmp_path).encode(), C.byref(n)), "sq_write_legacy_bias")
    assert n.value == 4096
    assert np.array_equal(np.frombuffer(gzip.open(tmp_path / "observed_bias.gz").read(), np.int32), np.ones(4096, np.int32))
    assert np.array_equal(np.frombuffer(gzip.open(tmp_path / "observed_bias_3p.gz").read(), np.int32), np.ones(4096, np.int32))
    assert np.array_equal(np.frombuffer(gzip.open(tmp_path / "expected_bias.gz").read(), np.float64), np.ones(4096))
    # GCFragModel::writeBinary: int32 dtype, int64 rows, int64 cols, totals[rows], counts column-major
    cnt = np.arange(75, dtype=np.float64).reshape(3, 25); tot = cnt.sum(axis=1)
    capi.check(capi.lib().sq_write_gc_model(str(tmp_path / "obs_gc.gz").encode(), 0, 3, 25, tot.ctypes.data, cnt.ctypes.data), "sq_write_gc_model")
    raw = gzip.open(tmp_path / "obs_gc.gz").read()
    assert struct.unpack("<iqq", raw[:20]) == (0, 3, 25) and np.array_equal(np.frombuffer(raw[20:44], np.float64), tot)
    assert np.array_equal(np.frombuffer(raw[44:], np.float64).reshape(25, 3).T, cnt)
    # SBModel::writeBinary: three int32s, three int32[9] tables, the 64 x 9 matrix and the 4 x 9 marginals with their int64 shapes
    lp = np.log(np.full((9, 64), 0.25)); capi.check(capi.lib().sq_write_seq_model(str(tmp_path / "obs5_seq.gz").encode(), lp.ctypes.data), "sq_write_seq_model")
    raw = gzip.open(tmp_path / "obs5_seq.gz").read()
    assert struct.unpack("<3i", raw[:12]) == (9, 3, 5) and struct.unpack("<9i", raw[12:48]) == (0, 1, 2, 2, 2, 2, 2, 2, 2)
    assert struct.unpack("<9i", raw[48:84]) == tuple(18 - 2 * (i + 1) for i in range(9)) and struct.unpack("<9i", raw[84:120]) == (2, 4, 6, 6, 6, 6, 6, 6, 6)
    assert struct.unpack("<qq", raw[120:136]) == (64, 9) and np.array_equal(np.frombuffer(raw[136:136 + 4608], np.float64).reshape(9, 64), lp)
    assert struct.unpack("<qq", raw[4744:4760]) == (4, 9) and np.allclose(np.frombuffer(raw[4760:], np.float64), 0.25)
    # the positional models: count, bounds, then (length, masses) per model
    ms = np.arange(100, dtype=np.float64); lb = np.array([500, 1000, 2000, 4000, 2 ** 32 - 1], np.uint32)
    capi.check(capi.lib().sq_write_pos_models(str(tmp_path / "obs5_pos.gz").encode(), 5, lb.ctypes.data, 20, ms.ctypes.data), "sq_write_pos_models")
    raw = gzip.open(tmp_path / "obs5_pos.gz").read()
    assert struct.unpack("<I", raw[:4]) == (5,) and np.array_equal(np.frombuffer(raw[4:24], np.uint32), lb) and len(raw) == 24 + 5 * (4 + 160)
    assert struct.unpack("<I", raw[24:28]) == (20,) and np.array_equal(np.frombuffer(raw[28:188], np.float64), ms[:20])


@pytest.mark.gpu
def test_gpu_cli_writes_the_reference_aux_info(built, tmp_path):
    """`salmon-hip quant` end to end: meta_info.json's keys are the reference's, the digests are the index's, fld.gz and the bias vectors exist, and
    with --seqBias --gcBias --posBias the model dumps GZipWriter::writeMeta writes under those flags are there."""
    import fixtures
    exe = os.path.join(ROOT, "salmon_amd", "bin", "salmon-hip"); g = fixtures.G
    names, seqs = fixtures.load_fasta()
    fa = tmp_path / "t.fa"
    with open(fa, "w") as f:
        for n, s in zip(names, seqs): f.write(">%s\n%s\n" % (n, s))
    subprocess.check_call([exe, "index", "-t", str(fa), "-i", str(tmp_path / "idx"), "-p", "2"])
    for tag, extra in (("plain", []), ("bias", ["--seqBias", "--gcBias", "--posBias"])):
        out = tmp_path / ("out_" + tag)
        subprocess.check_call([exe, "quant", "-i", str(tmp_path / "idx"), "-l", "IU", "-1", os.path.join(g, "reads_1.fq.gz"), "-2", os.path.join(g, "reads_2.fq.gz"), "-o", str(out), "-q"] + extra)
        meta = json.load(open(out / "aux_info" / "meta_info.json"))
        assert [k for k in meta if k != "salmon_hip"] == GOLD["keys"]
        assert meta["index_seq_hash"] == hashlib.sha256("".join(seqs).encode()).hexdigest() and meta["index_name_hash"] == hashlib.sha256("".join(names).encode()).hexdigest()
        assert meta["index_seq_hash512"] == hashlib.sha512("".join(seqs).encode()).hexdigest() and len(meta["index_decoy_seq_hash"]) == 64
        assert meta["num_bias_bins"] == 4096 and meta["frag_dist_length"] == 1001 and 200 < meta["frag_length_mean"] < 300 and 5 < meta["frag_length_sd"] < 80
        assert meta["keep_duplicates"] is False and meta["eq_class_properties"] == ["range_factorized", "gzipped"] and len(meta["length_classes"]) >= 1
        assert meta["start_time"] and meta["end_time"] and meta["seq_bias_correct"] == (tag == "bias") and meta["gc_bias_correct"] == (tag == "bias")
        have = set(os.listdir(out / "aux_info"))
        assert set(GOLD["aux_files_always"]) | {"meta_info.json", "ambig_info.tsv"} <= have
        fld = np.frombuffer(gzip.open(out / "aux_info" / "fld.gz").read(), np.int32)
        assert len(fld) == 1001 and fld.sum() == 10000
        if tag == "bias":
            assert set(GOLD["aux_files_seq_bias"]) | set(GOLD["aux_files_pos_bias"]) | set(GOLD["aux_files_gc_bias"]) <= have
            raw = gzip.open(out / "aux_info" / "exp5_seq.gz").read()
            assert struct.unpack("<3i", raw[:12]) == (9, 3, 5) and len(raw) == 120 + 16 + 4608 + 16 + 288
            raw = gzip.open(out / "aux_info" / "obs5_pos.gz").read()
            nm = struct.unpack("<I", raw[:4])[0]; assert nm == len(meta["length_classes"]) and len(raw) == 4 + 4 * nm + nm * (4 + 160)
        else:
            assert not (set(GOLD["aux_files_seq_bias"]) & have)
