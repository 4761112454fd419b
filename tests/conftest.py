import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
    """Random CSR eq-classes with sorted, duplicate-free labels."""
    from salmon_amd import api
    rng = np.random.default_rng(seed)
    sizes = rng.integers(1, max_size + 1, E)
    sizes = np.minimum(sizes, M)
    off = np.zeros(E + 1, np.uint64); off[1:] = np.cumsum(sizes)
    L = int(off[-1])
    tid = np.zeros(L, np.uint32); w = np.zeros(L)
    hot = rng.zipf(1.3, size=M).astype(np.float64); hot /= hot.sum()
    for c in range(E):
        a, b = int(off[c]), int(off[c + 1])
        t = np.sort(rng.choice(M, size=b - a, replace=False, p=hot))
        tid[a:b] = t
        x = np.ones(b - a) if init_uniform_weights else rng.random(b - a) + 0.05
        w[a:b] = x / x.sum()
    count = rng.integers(1, 200, E).astype(np.uint64)
    return api.EqClasses(off, tid, w, count)
