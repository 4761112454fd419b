import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # [r5] the first `import torch` on a fresh GPU box pages the image in and has been seen to take more than seven minutes — inside a test it ran
    # into pytest-timeout and cost the run a test that never started.  For a GPU run it happens here, before any test's clock
    m = config.getoption("-m", default="") or ""
    if "gpu" in m and "not gpu" not in m:
        try: import torch  # noqa: F401
        except Exception: pass


@pytest.fixture(scope="session")
def built():
    from salmon_amd import build
    build.build_all()
    return True


@pytest.fixture(scope="session")
def small_world(built):
    """A small synthetic transcriptome, its index, the checker's index and 4000 read pairs."""
    from salmon_amd import api, synth
    import orc
    tx = synth.Txome(seed=7, n_genes=120, iso_per_gene=6, threads=4)
    names, seqs, lens = tx.tables()
    idx = api.SalmonIndex.build_mem_raw(tx.n, names, seqs, lens, threads=4)
    oidx = orc.OrcIndex(idx)
    seq, off, tt, tp = tx.reads(4000, read_len=100, seed=11, threads=4)
    return dict(tx=tx, idx=idx, oidx=oidx, seq=seq, off=off, truth_tid=tt, truth_pos=tp, n=4000)


def random_eq_classes(M, E, seed=0, max_size=8, init_uniform_weights=False):
    """Random CSR eq-classes with sorted, duplicate-free labels (vectorised; zipf-hot transcripts)."""
    from salmon_amd import api
    rng = np.random.default_rng(seed)
    sizes = np.minimum(rng.integers(1, max_size + 1, E), M)
    cls = np.repeat(np.arange(E), sizes)
    hot = rng.zipf(1.3, size=M).astype(np.float64); hot /= hot.sum()
    tid = rng.choice(M, size=len(cls), replace=True, p=hot)
    key = np.unique(cls.astype(np.int64) * M + tid)          # sorted by (class, tid), duplicates removed
    cls2 = (key // M).astype(np.int64); tid2 = (key % M).astype(np.uint32)
    # every class keeps at least one label (cls is dense in 0..E-1 because sizes >= 1)
    cnt = np.bincount(cls2, minlength=E)
    off = np.zeros(E + 1, np.uint64); off[1:] = np.cumsum(cnt)
    x = np.ones(len(tid2)) if init_uniform_weights else rng.random(len(tid2)) + 0.05
    s = np.add.reduceat(x, off[:-1].astype(np.int64))
    w = x / np.repeat(s, cnt)
    count = rng.integers(1, 200, E).astype(np.uint64)
    return api.EqClasses(off, tid2, w, count)
