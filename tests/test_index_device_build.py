"""[r6] Index construction with its k-mer table built on the device (hip/index_build_dev.hip) against the host's partitioned table
(host/index_build.cpp): the index files are byte for byte the same — one partition, several partitions under a small memory cap, references
shorter than k, repeated sequence, a decoy block — and a build without a device takes the host's passes."""
import hashlib
import os
import numpy as np
import pytest
from salmon_amd import api, capi, synth


def _sha_dir(d):
    out = {}
    for f in sorted(os.listdir(d)):
        if f in ("versionInfo.json",): continue
        out[f] = hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest()
    return out


def _refs():
    rng = np.random.default_rng(11)
    tx = synth.Txome(seed=3, n_genes=300, iso_per_gene=4, threads=4)
    names = list(tx.names()); seqs = [x.decode() for x in tx.seqs()]
    # what a transcriptome does not have: a reference shorter than k, a homopolymer, a tandem repeat, a long "chromosome" with a copy of a transcript inside
    names += ["short", "polyA", "tandem"]; seqs += ["ACGTACGTAC", "A" * 200, "ACGGT" * 60]
    chrom = "".join("ACGT"[x] for x in rng.integers(0, 4, 60000)); chrom = chrom[:20000] + seqs[5] + chrom[20000:]
    names.append("chr_decoy"); seqs.append(chrom)
    return names, seqs


@pytest.mark.gpu
def test_device_and_host_tables_build_the_same_index(built, tmp_path, monkeypatch):
    names, seqs = _refs(); L = capi.lib(); nd = len(names) - 1
    try:
        capi.check(L.sq_index_build_set_device(-1), "host")
        api.SalmonIndex.build_mem(names, seqs, threads=4, first_decoy=nd, outdir=str(tmp_path / "host")).free()
        capi.check(L.sq_index_build_set_device(0), "device")
        api.SalmonIndex.build_mem(names, seqs, threads=4, first_decoy=nd, outdir=str(tmp_path / "dev")).free()
        monkeypatch.setenv("SQ_INDEX_DEVICE_GB", "0.004")      # ~4 MB of table: several partitions of the ~0.9 M positions
        api.SalmonIndex.build_mem(names, seqs, threads=4, first_decoy=nd, outdir=str(tmp_path / "dev_parts")).free()
        monkeypatch.delenv("SQ_INDEX_DEVICE_GB")
        monkeypatch.setenv("SQ_INDEX_DEVICE_MIN_POS", "1000")   # the automatic choice takes the device when the input is large enough
        capi.check(L.sq_index_build_set_device(-2), "auto")
        api.SalmonIndex.build_mem(names, seqs, threads=4, first_decoy=nd, outdir=str(tmp_path / "auto")).free()
    finally:
        L.sq_index_build_set_device(-2)
    h = _sha_dir(str(tmp_path / "host"))
    assert "index.bin" in h
    for d in ("dev", "dev_parts", "auto"): assert _sha_dir(str(tmp_path / d)) == h, d


def test_without_a_device_the_host_builds_and_an_explicit_device_is_an_error(built, tmp_path):
    import torch
    if torch.cuda.is_available(): pytest.skip("a device is present")
    names, seqs = _refs(); L = capi.lib()
    try:
        os.environ["SQ_INDEX_DEVICE_MIN_POS"] = "1000"
        capi.check(L.sq_index_build_set_device(-2), "auto")
        api.SalmonIndex.build_mem(names, seqs, threads=4, first_decoy=len(names) - 1, outdir=str(tmp_path / "auto")).free()
        capi.check(L.sq_index_build_set_device(-1), "host")
        api.SalmonIndex.build_mem(names, seqs, threads=4, first_decoy=len(names) - 1, outdir=str(tmp_path / "host")).free()
        assert _sha_dir(str(tmp_path / "auto")) == _sha_dir(str(tmp_path / "host"))
        capi.check(L.sq_index_build_set_device(0), "device")
        with pytest.raises(capi.SalmonHipError): api.SalmonIndex.build_mem(names, seqs, threads=4, first_decoy=len(names) - 1, outdir=str(tmp_path / "dev"))
        assert L.sq_index_build_set_device(-7) != 0
    finally:
        os.environ.pop("SQ_INDEX_DEVICE_MIN_POS", None); L.sq_index_build_set_device(-2)
