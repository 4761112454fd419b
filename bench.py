#!/usr/bin/env python
"""bench.py — `salmon quant` hot path (map + eq-classes + EM) on MI355X, BASELINE.json metric.

  python bench.py --gpus N --steps K --warmup W [--workload c2|c2s|c3|c5|c4]   (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path (sq_map_batch + sq_eq_accumulate) over one batch of synthetic read pairs already
resident in HBM; a batch is handed over as `--sub` consecutive sq_map_batch calls of `--batch` pairs (the library maps at
most 2^23 pairs per call).  Workloads (BASELINE.json `configs`):

  c2  (default, configs[1]/[2])  human-transcriptome-shaped index (synthetic: 60 000 genes x ~4 isoforms, ~190 k transcripts,
        ~430 Mnt, ~150 M distinct 31-mers — SURVEY.md §8 shape C2), K x 5 M 2x100 bp pairs (the driver's --steps 20 = the 100 M-pair
        job BASELINE.json's metric names), then the job's inference tail (eq-class export, normalizeAlphas, VBEM to convergence),
        all inside the timed region.  [r4] Beside it, outside the timed steps (rank 0, N = 1): the same job on the first 10 M pairs
        (`jobs.10M` = configs[1] as stated), the same job on the c2s index (`c2s`), and `spread` (how far NumReads / TPM move with the
        mini-batches in flight, the batch size and the rank count).
  c2s the round-1/2 index (20 000 genes x ~10 isoforms: 54 M distinct k-mers, 5.3 alignments per fragment), same job.
  c3  (configs[2]) STRONG scaling: a fixed total of K x 5 M pairs (100 M at --steps 20) split over the WORLD_SIZE ranks; every rank
        maps its share, one RCCL exchange of the class tables, EM replicated.  `"scaling": "strong"`.
  c5  (configs[4]) c2's index, 50 M pairs, VBEM, then 100 Gibbs samples (4 chains x 25 samples x 16 thinning rounds) in the timed region.
  c4  (configs[3]) decoy-aware index: c2's transcriptome + a synthetic genome (`--genome-gnt`, default 1.0 Gnt; every gene's exons
        with introns, 45 % repeat families) as decoys, 2x150 bp pairs of which 5 % come from gene loci of the genome.

Weak scaling: every rank maps its own K batches; eq-class tables are all-gathered over RCCL and merged exactly.
Prints ONE JSON line on rank 0.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# [r6] before anything initialises the HIP runtime: pageable copies go through the runtime's staging buffers instead of pinning the caller's pages — cached pins of memory that
# is freed later make amdkfd evict the process's queues, and the next submission starts 6-30 ms late (libsalmon_hip.so sets the same default when it is loaded first:
# salmon_amd/csrc/hip/map.hip, sq_runtime_defaults; profiles/r06_eviction_ab.txt).  An explicit setting of the caller wins
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")
import numpy as np

WORKLOADS = {
    # genes, iso, read_len, batch, sub, genome_gnt, genomic_frac
    "c2": dict(genes=60000, iso=4, read_len=100, batch=5000000, sub=1, genome_gnt=0.0, genomic=0.0),
    "c2s": dict(genes=20000, iso=10, read_len=100, batch=5000000, sub=1, genome_gnt=0.0, genomic=0.0),
    "c3": dict(genes=60000, iso=4, read_len=100, batch=5000000, sub=1, genome_gnt=0.0, genomic=0.0),
    "c5": dict(genes=60000, iso=4, read_len=100, batch=5000000, sub=1, genome_gnt=0.0, genomic=0.0),
    "c4": dict(genes=60000, iso=4, read_len=150, batch=4000000, sub=1, genome_gnt=1.0, genomic=0.05),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 20 (c2, c2s, c3 = 100 M pairs), 10 (c5 = 50 M pairs), 6 (c4)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="read pairs per sq_map_batch call (<= 2^23)")
    ap.add_argument("--sub", type=int, default=None, help="sq_map_batch calls per step")
    ap.add_argument("--genes", type=int, default=None)
    ap.add_argument("--iso", type=int, default=None)
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--genome-gnt", type=float, default=None, help="c4: decoy genome size in 10^9 nt")
    ap.add_argument("--gibbs-samples", type=int, default=100, help="c5: posterior samples (100 = 4 chains, 1600 rounds)")
    ap.add_argument("--cpu-sample", type=int, default=5000000, help="pairs timed through the CPU checker and compared with the HIP path (0 = skip)")
    ap.add_argument("--no-chain-check", dest="chain_check", action="store_false", help="skip the parity leg that runs the burned-in mini-batch chain against the checker (2 M pairs)")
    ap.add_argument("--no-extras", action="store_true", help="c2: skip the 10 M-pair job, the c2s leg and the spread measurement")
    ap.add_argument("--spread-pairs", type=int, default=10000000, help="c2 extras: pairs of the job the spread variants run (0 = skip)")
    ap.add_argument("--index-cache", default=None, help="directory to keep the built index in between runs (experiments; the driver's run builds it)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
        help="initialise torch.distributed (RCCL) even for one rank, so the N>1 code path (the RCCL exchange + merge) runs on a 1-GPU box")
    ap.add_argument("--debug-one-device", action="store_true",
        help="functional check of the N>1 path on a 1-GPU box: every rank uses cuda:0 and the collectives run over gloo (numbers meaningless)")
    ap.add_argument("--fastq-gz-pairs", type=int, default=20000000, help="pairs of the gzip / BGZF legs of the from-FASTQ run (compressing the input is what takes the time)")
    ap.add_argument("--fastq-pairs", type=int, default=20000000,
        help="second measurement (outside the timed steps, rank 0, N=1, c2/c2s): this many pairs written as FASTQ files to /dev/shm and run through "
             "the host read pipeline (sq_reader) + the same GPU path, end to end from files (0 = skip)")
    ap.add_argument("--py-dist", action="store_true",
        help="N>1: exchange the class tables with torch.distributed collectives (salmon_amd/dist.py) instead of the library's own RCCL path (sq_dist_*, the default)")
    ap.add_argument("--inflight", type=int, default=0,
        help="mini-batches per model snapshot (sq_quant_opts.mini_batches_in_flight = the reference's -p / numThreads); 0 = the library default (8)")
    ap.add_argument("--dry-launch", action="store_true", help="print the command `--gpus N` would start its ranks with, and exit (no device needed)")
    ap.add_argument("--lanes", type=int, default=1, help="batches in flight on the mapping lanes (sq_map_submit/sq_map_wait); 1 = plain sq_map_batch")
    a = ap.parse_args()
    w = WORKLOADS[a.workload]
    for k_arg, k_w in (("batch", "batch"), ("sub", "sub"), ("genes", "genes"), ("iso", "iso"), ("read_len", "read_len"), ("genome_gnt", "genome_gnt")):
        if getattr(a, k_arg) is None: setattr(a, k_arg, w[k_w])
    a.genomic = w["genomic"] if a.genome_gnt > 0 else 0.0
    if a.steps is None: a.steps = {"c4": 6, "c5": 10}.get(a.workload, 20)
    return a


# algorithmic bytes per unit for every stage (DESIGN.md §5 states the same table); `st` = the sq_map_stats counters summed over the timed steps
def stage_bytes(st, n_pairs, read_len, paired=True):
    nrec = 2 * n_pairs if paired else n_pairs
    L = read_len
    pk = 8 * ((L + 31) // 32) + 4 * ((L + 63) // 64) * 2     # packed read + N mask actually touched
    return {
        "k_pack": nrec * (L + 8 + 64 + 32 + 2 + 1),
        # [r6] k_seed2's own algorithmic bytes — what `roofline.frac` is quoted on: the packed words of a read end the kernel loads (32 B for reads of
        # up to 128 bases, 64 B up to 256) + its length, the byte "this end has an N" (the 32-byte mask itself is read for those ends only: not charged) and the two
        # counts it leaves, the 64-byte filter block once per run of probes that share a minimizer
        # (`filter_fills`, counted by the kernel), and per uni-MEM three dependent sectors (minimizer-table bucket, string-pool word, unitig bounds)
        # + extension words, contig-table bounds and its 32-byte record
        "k_seed": nrec * ((32 if L <= 128 else 64) + 1 + 2 + 8) + st.get("filter_fills", st["num_lookups"]) * 64 + st["num_seeds"] * (3 * 64 + 16 + 16 + 32 + 16),
        # round 4's model of the same job for the kernel k_seed2 replaced (one filter sector per PROBE, four dependent sectors per hit: pilot, slot
        # record, string-pool word, unitig bounds).  k_seed2 does not move these bytes; kept as a separately named field (`roofline.round4_model`)
        # so that rounds 3-5 can still be compared, never as `frac`
        "k_seed_r4model": nrec * (64 + 32 + 2 + 8) + st["num_lookups"] * 64 + st["num_seeds"] * (4 * 64 + 16 + 16 + 32 + 16),
        "scan_mems": nrec * (4 + 8),
        # fused projection + per-end sort + chaining (mem_kernels.h): uni-MEM records and contig-table runs in, sorted MEM records and chains out
        "k_mems": st["num_seeds"] * 32 + st["num_mems"] * (8 + 8 + 16) + st["num_chains"] * 40 + nrec * (16 + 4),
        "k_join_fill": st["num_chains"] * 40 + st["num_candidates"] * 52 + n_pairs * 16,
        # [r3] SURVEY §8(d): each distinct sector once per read — the packed reads and chain heads of a fragment are charged once per
        # fragment (its candidates share them), the candidate record, its reference window and its score once per candidate
        "k_score": st["num_candidates"] * (48 + 64 * ((L + 40 + 255) // 256) + 16 + 4) + n_pairs * 2 * pk + st["num_chains"] * 16,
        "k_dp": st["num_dp_alignments"] * (48 + 96 + 64 + 2 * 40 + 2 * 4),   # + the queue read twice more and the permutation (counting sort by length)
        "k_finalize": st["num_candidates"] * (48 + 8) + n_pairs * 8,
        "k_select": st["num_candidates"] * 48 + st["num_alignments"] * 40 + n_pairs * 30,
        "compact_alns": st["num_alignments"] * 80 + n_pairs * 28,
        "eq_flags_scan": st["num_alignments"] * (40 + 32) + n_pairs * 24,
        # [r3] after burn-in the online stage is a model-independent launch per batch (eq_static: per alignment the 32-byte pre-record in, the 24-byte
        # dynamic record + fixed-point weight + bin out, two counters; per fragment offsets and the label hash) and per group of mini-batches
        # the mass terms (eq_mini_batches = k_frag_dynamic + k_apply_flagged: the dynamic record, the transcript's log-count, one mass slot)
        "eq_static": st["num_alignments"] * (32 + 24 + 8 + 4 + 2 * 8) + n_pairs * (16 + 16),
        "eq_mini_batches": st["num_alignments"] * (24 + 8 + 16 + 8) + n_pairs * 16,
        "eq_table": st["num_alignments"] * (4 + 4 + 8 + 8) + n_pairs * (16 + 4 + 32),
    }


KERNEL_OF_STAGE = {"k_pack": "k_pack8_staged", "k_seed": "k_seed2", "k_mems": "k_mems", "k_join_fill": "k_join2", "k_score": "k_score", "k_dp": "k_dp",
                   "k_select": "k_select", "k_finalize": "k_finalize", "eq_mini_batches": "k_frag_dynamic", "eq_static": "k_frag_static",
                   "eq_table": "k_eq_insert"}


def _fastq_pass(ctx, idx, files, batch, lib, api, capi, read_len):
    import ctypes as C
    ctx.reset()
    a1 = (C.c_char_p * 1)(files[0].encode()); a2 = (C.c_char_p * 1)(files[1].encode()); h = C.c_void_p()
    lanes = 2
    lib.sq_ctx_set_lanes(ctx.h, lanes)
    t0 = time.perf_counter()
    capi.check(lib.sq_reader_open(a1, 1, a2, 1, batch, lanes + 2, C.byref(h)), "sq_reader_open")   # one slot staging, one uploading, one per mapping lane
    inflight = []; n = 0; tot_mapped = 0
    def finish_one():
        nonlocal tot_mapped
        _, _, _, st = ctx.map_wait(); ctx.eq_accumulate(); tot_mapped += st["num_mapped"]
        lib.sq_reader_release(h, inflight.pop(0)[1])
    while True:
        rb = capi.ReadBatch(); slot = C.c_int(-1)
        capi.check(lib.sq_reader_next(h, C.byref(rb), C.byref(slot)), "sq_reader_next")
        if rb.n == 0: break
        if len(inflight) == lanes: finish_one()
        ctx.map_submit(rb); inflight.append((rb, slot.value)); n += rb.n
    while inflight: finish_one()
    t_read_map = time.perf_counter() - t0
    eq = ctx.eq_finish(); lm, uq, tc, le = ctx.model()
    proj = api.normalize_alphas(eq, lm, uq, tc)
    alphas, rep = ctx.em_optimize(np.exp(le), proj, api.em_opts())
    dt = time.perf_counter() - t0
    lib.sq_reader_close(h)
    return {"value": round(n / dt / 1e6, 3), "unit": "M read-pairs/s", "pairs": int(n), "seconds": round(dt, 4), "read_map_eq_s": round(t_read_map, 4),
            "mapped_frac": round(tot_mapped / max(1, n), 4), "em_iters": rep["iters"]}


def run_from_fastq(ctx, idx, tx, n_pairs, read_len, batch, threads, api, capi, gz_pairs=None):
    """`salmon quant` from FASTQ files: sq_reader (parallel record splitting into page-locked batches) -> H2D -> mapping lanes -> online model /
    eq-classes -> export -> normalizeAlphas -> VBEM.  Wall time from opening the files to the converged alphas; plain, BGZF and gzip input."""
    import shutil, tempfile, subprocess, gzip
    d = tempfile.mkdtemp(prefix="sq_bench_fq_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    # the files (2 x n x (2 L + 7) bytes, the head copies and their compressed forms: ~1.4 x) have to fit where they are written: fewer pairs if they do not
    room = shutil.disk_usage(d).free; need = lambda n: int(1.8 * 2 * n * (2 * read_len + 7)) + (256 << 20)
    while n_pairs > 1000000 and need(n_pairs) > room: n_pairs //= 2
    gz_pairs = min(n_pairs, gz_pairs or n_pairs)
    out = {"input": "2 FASTQ files of %d x %d bp in /dev/shm (page cache), batches of %d pairs; the gzip / BGZF legs read the first %d pairs of them" % (n_pairs, read_len, batch, gz_pairs), "host_threads": os.cpu_count(), "host_cpu_quota_cores": cpu_allowance()[1],
           "reader_threads": os.environ.get("SQ_READER_THREADS", "default: min(64, hw/2) inflating, min(16, hw/4) copying"),
           "what": "end to end from files through sq_reader, H2D included; plain files: text staged in page-locked memory, records split on the device "
                   "(hip/fastq_dev.hip); BGZF: the compressed members cross PCIe and are inflated on the device (hip/inflate_dev.hip, a wave per member), records split there; "
                   "gzip (one deflate stream per file, no member table): [r6] inflated on the device as well (hip/gzip_dev.hip: block starts found by trying bit offsets, a wave per span between two of them into 16-bit symbols, the 32 KB windows resolved span after span, CRC-32 and length checked), records split there"}
    try:
        seq, off, _, _ = tx.reads(n_pairs, read_len=read_len, seed=77, first_pair=0, threads=threads, truth=False)
        recs = seq.reshape(2 * n_pairs, read_len)
        L = read_len; row = np.empty((n_pairs, 3 + L + 3 + L + 1), np.uint8)
        row[:, 0:3] = np.frombuffer(b"@r\n", np.uint8); row[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", np.uint8); row[:, 6 + L:6 + 2 * L] = ord("I"); row[:, -1] = 10
        files = []
        for m in (0, 1):
            row[:, 3:3 + L] = recs[m::2]
            f = os.path.join(d, "r_%d.fq" % (m + 1)); row.tofile(f); files.append(f)
        del row, recs, seq
        lib = capi.lib(); ctx.set_profiling(False)
        res = None
        for attempt in range(2):   # the first pass sizes the work buffers of both mapping lanes and the reader's page-locked slots; the second is reported
            res = _fastq_pass(ctx, idx, files, batch, lib, api, capi, read_len)
        out["plain"] = res; out["value"] = res["value"]; out["unit"] = res["unit"]
        # gzip (one member: the reader inflates it on one thread per mate file unless it can split it) and BGZF (blocked gzip: members inflate in parallel)
        try:
            rowb = 3 + read_len + 3 + read_len + 1
            if gz_pairs < n_pairs:
                short = []
                for f in files:
                    h = f + ".head.fq"; subprocess.check_call("head -c %d %s > %s" % (gz_pairs * rowb, f, h), shell=True); short.append(h)
                files = short
            gz = []
            for f in files:
                g = f + ".gz"; _write_gzip(f, g); gz.append(g)
            out["gzip"] = _fastq_pass(ctx, idx, gz, batch, lib, api, capi, read_len)
            bg = []
            for f in files:
                g = f + ".bgz.gz"; _write_bgzf(f, g); bg.append(g)
            out["bgzf"] = _fastq_pass(ctx, idx, bg, batch, lib, api, capi, read_len)
        except Exception as e:   # the compressed legs are extras: never lose the bench line over them
            out["compressed_error"] = str(e)[:200]
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _write_gzip(src, dst, chunk=16 << 20):
    """ONE gzip member (what `gzip -1` writes: a single deflate stream), compressed by Python threads the way pigz does it: every chunk is deflated on
    its own and ends with a sync flush (byte-aligned, no final block), the last one finishes the stream; header, bodies, CRC-32 and length follow each other."""
    import zlib, struct
    from concurrent.futures import ThreadPoolExecutor
    data = np.fromfile(src, np.uint8); starts = list(range(0, len(data), chunk)) or [0]
    def part(i):
        co = zlib.compressobj(1, zlib.DEFLATED, -15); piece = data[i:i + chunk].tobytes()
        return co.compress(piece) + co.flush(zlib.Z_FINISH if i == starts[-1] else zlib.Z_FULL_FLUSH), zlib.crc32(piece), len(piece)
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex, open(dst, "wb") as f:
        f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x04\x03"); crc = 0; first = True
        for body, c, n in ex.map(part, starts):
            f.write(body); crc = c if first else _crc32_combine(crc, c, n); first = False
        f.write(struct.pack("<II", crc & 0xffffffff, len(data) & 0xffffffff))


def _crc32_combine(crc1, crc2, len2):
    """zlib's crc32_combine (not exposed by Python): the CRC-32 of A || B from CRC(A), CRC(B) and len(B), by squaring the shift operator over GF(2)."""
    def times(mat, vec):
        s = 0; i = 0
        while vec:
            if vec & 1: s ^= mat[i]
            vec >>= 1; i += 1
        return s
    def square(mat): return [times(mat, mat[n]) for n in range(32)]
    if len2 <= 0: return crc1
    odd = [0xedb88320] + [1 << n for n in range(31)]
    even = square(odd); odd = square(even)
    while True:
        even = square(odd)
        if len2 & 1: crc1 = times(even, crc1)
        len2 >>= 1
        if not len2: break
        odd = square(even)
        if len2 & 1: crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2: break
    return crc1 ^ crc2


def _write_bgzf(src, dst, block=0xff00):
    """Blocked gzip (BGZF: every member carries its compressed size in a 'BC' extra field), written with zlib from Python threads."""
    import zlib, struct
    from concurrent.futures import ThreadPoolExecutor
    data = np.fromfile(src, np.uint8)
    def member(i):
        chunk = data[i:i + block].tobytes()
        co = zlib.compressobj(1, zlib.DEFLATED, -15); body = co.compress(chunk) + co.flush()
        hdr = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(body) + 25)
        return hdr + body + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex, open(dst, "wb") as f:
        for m in ex.map(member, range(0, len(data), block)): f.write(m)
        f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00")


def cpu_stat():
    """The cgroup's CPU accounting (usage_usec, nr_periods, nr_throttled, throttled_usec), {} where there is none: read before and after the timed region, it says how many
    cores the host side used and whether the container's CPU quota stopped it meanwhile."""
    try: return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines()) if k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec")}
    except Exception: return {}


def kfd_evicted_ms():
    """Milliseconds this process's device queues have spent EVICTED so far (amdkfd's per-process counter; None where it is not exposed).  The driver takes a process's
    queues off the device while it rebuilds mappings — e.g. when host memory it had pinned for a copy is unmapped — and the next submission waits for their return."""
    import glob
    try:
        fs = glob.glob("/sys/class/kfd/kfd/proc/*/stats_*/evicted_ms")    # (the directories carry host pids, not this container's: all of them are summed, a difference over the timed region is this job's unless the node is shared)
        return sum(int(open(f).read()) for f in fs) if fs else None
    except Exception: return None


def cpu_allowance():
    """(hardware threads the process may be scheduled on, CPU quota of its cgroup in cores or None): the GPU boxes of this pool show 256 threads and give a
    container 16 cores' worth of time (cpu.max = 1600000 100000) — the quota, not the thread count, is what the host legs have."""
    hw = os.cpu_count() or 8; quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max": quota = float(q) / float(p)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0: quota = q / p
        except Exception: pass
    return hw, quota


def baseline_metric():
    """BASELINE.json's metric string, verbatim (one "read" there is one 2x100 bp pair; `unit` says so explicitly)."""
    try:
        return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")))["metric"]
    except Exception:
        return "M reads/s quantified (map+EM), 100M 2x100bp vs human txome; EM iters/s"


def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def kernel_source_sha():
    """sha of the kernel sources the PMC traffic profile was taken on: a traffic figure is attached only when it still matches (weak #10 of the round-3 review)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("map_kernels.h", "mem_kernels.h", "map.hip", "ctx.h", "online.hip"):
        h.update(open(os.path.join(ROOT, "salmon_amd", "csrc", "hip", f), "rb").read())
    return h.hexdigest()[:16]


class World:
    """A workload's index on the device + its read generator."""

    def __init__(self, a, genes, iso, genome_gnt, genomic, thr, local, api, synth):
        t0 = time.time()
        self.tx = synth.Txome(seed=1, n_genes=genes, iso_per_gene=iso, threads=min(thr, 32))
        names, seqs, lens = self.tx.tables()
        self.genome = None; first_decoy = None; n_in = self.tx.n
        if genome_gnt > 0:   # c4: the genome's chromosomes follow the transcripts as decoys (`salmon index -d decoys.txt`)
            self.genome = synth.Genome(self.tx, seed=3, total_nt=int(genome_gnt * 1e9), n_chrom=25, repeat_frac=0.45, threads=min(thr, 32))
            names, seqs, lens = self.genome.append_tables(self.tx)
            first_decoy = self.tx.n; n_in = self.tx.n + self.genome.n
        self.t_synth = time.time() - t0
        cache = os.path.join(a.index_cache, "idx_g%d_i%d_d%.3f" % (genes, iso, genome_gnt)) if a.index_cache else None
        if cache: os.makedirs(a.index_cache, exist_ok=True)
        if cache and os.path.exists(os.path.join(cache, "info.json")):      # experiments: several runs of one gpurun call share the index files
            self.idx = api.SalmonIndex.load(cache)
        else:
            self.idx = api.SalmonIndex.build_mem_raw(n_in, names, seqs, lens, threads=thr, first_decoy=first_decoy, outdir=cache)
        self.t_index = time.time() - t0 - self.t_synth
        self.idx.to_device(local)
        self.genomic = genomic; self.thr = thr

    def reads(self, n, read_len, first_pair):
        if self.genome is not None:
            return self.genome.reads(self.tx, n, read_len=read_len, seed=2, first_pair=first_pair, genomic_frac=self.genomic, threads=min(self.thr, 64), truth=False)[0]
        return self.tx.reads(n, read_len=read_len, seed=2, first_pair=first_pair, threads=min(self.thr, 64), truth=False)[0]

    def free(self):
        self.idx.free(); self.tx.free()
        if self.genome is not None: self.genome.free()


def one_job(ctx, rbs, api, idx, em_opts=None):
    """One whole job on one rank, timed like the headline: reset, map + online model + eq-classes over `rbs`, export, normalizeAlphas, VBEM to
    convergence.  Returns (seconds, alphas, eff_lens, report)."""
    import torch
    ctx.reset(); ctx.set_profiling(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0; mapped = 0
    for rb in rbs:
        _, _, _, st = ctx.map_batch(rb, fetch=False); ctx.eq_accumulate(); n += rb.n; mapped += st["num_mapped"]
    t_map = time.perf_counter() - t0
    eq = ctx.eq_finish(); lm, uq, tc, le = ctx.model()
    proj = api.normalize_alphas(eq, lm, uq, tc); eff = np.exp(le)
    alphas, rep = ctx.em_optimize(eff, proj, em_opts or api.em_opts())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rep = dict(rep); rep.update(pairs=n, seconds=dt, map_eq_s=t_map, mapped=mapped, eq_classes=len(eq.count), labels=len(eq.tid))
    return dt, alphas, eff, rep


def tpm_of(alphas, eff):
    r = alphas / np.maximum(eff, 1e-300); s = r.sum()
    return r * (1e6 / s) if s > 0 else r


def rel_spread(base, other, floor):
    """Relative difference |x - y| / max(x, y) over the entries where either side is >= floor: (max, 99.9th percentile, 99th, median, entries)."""
    m = (base >= floor) | (other >= floor)
    if not m.any(): return None
    d = np.abs(base[m] - other[m]) / np.maximum(base[m], other[m])
    return {"max": float(d.max()), "p999": float(np.quantile(d, 0.999)), "p99": float(np.quantile(d, 0.99)), "median": float(np.median(d)), "entries": int(m.sum())}


def run_spread(a, world_obj, parked, off_d, B, RL, api, capi, local):
    """[r4] What the deterministic online stage's free constants do to the result (oracle/SPEC.md §D1, §MG): the same `--spread-pairs` job with
    W = mini-batches per model snapshot in {1, 8, 32}, handed over in batches of 1 M instead of 5 M pairs, and split over two ranks (two
    contexts on this device, tables exchanged through the communicator's buffers: sq_dist_merge_eq_loopback).  For every variant the relative
    difference of NumReads and TPM against the default (W = 8, one rank, 5 M-pair batches)."""
    import torch, ctypes as C
    idx = world_obj.idx; nb = max(1, min(len(parked), a.spread_pairs // B)); N = nb * B
    def rbs_of(batch_pairs):
        out = []
        for t in parked[:nb]:
            for lo in range(0, B, batch_pairs):
                n = min(batch_pairs, B - lo)
                out.append(api.make_read_batch(int(t.data_ptr()) + 2 * RL * lo, int(off_d.data_ptr()), n, paired=True, on_device=True))
        return out
    res = {}; base = None
    def variant(name, W, batch_pairs):
        nonlocal base
        o = api.quant_opts();
        if W: o.mini_batches_in_flight = W
        c = api.QuantContext(idx, o, device=local, max_batch_reads=batch_pairs); c.reserve(1000000, 0)
        one_job(c, rbs_of(batch_pairs)[:1], api, idx)                       # sizes the work buffers
        dt, al, eff, rep = one_job(c, rbs_of(batch_pairs), api, idx)
        c.free()
        tp = tpm_of(al, eff)
        if base is None: base = (al, tp)
        res[name] = {"W": W or 8, "batch_pairs": batch_pairs, "ranks": 1, "seconds": round(dt, 4), "em_iters": rep["iters"], "eq_classes": rep["eq_classes"],
                     "num_reads_ge_0.01": rel_spread(base[0], al, 1e-2), "num_reads_ge_10": rel_spread(base[0], al, 10.0),
                     "tpm_ge_0.01": rel_spread(base[1], tp, 1e-2), "tpm_ge_1": rel_spread(base[1], tp, 1.0)}
    tables = {}
    def eq_digest(eq): return sha(np.concatenate([eq.off.view(np.uint8), eq.tid.view(np.uint8), eq.bins.view(np.uint8), eq.count.view(np.uint8), eq.wq.view(np.uint8)]))
    variant("W8_default", 0, B)
    variant("W1", 1, B); variant("W32", 32, B); variant("W64", 64, B)
    B1 = min(B, 1000000)
    # the one-rank job in 1 M-pair batches, kept: the two-rank job below uses the same batch plan
    o = api.quant_opts(); c1 = api.QuantContext(idx, o, device=local, max_batch_reads=B1); c1.reserve(1000000, 0)
    one_job(c1, rbs_of(B1)[:1], api, idx)
    dt, al1, eff1, rep1 = one_job(c1, rbs_of(B1), api, idx); tp1 = tpm_of(al1, eff1); tables["one_rank_1M"] = eq_digest(c1.eq_finish()); c1.free()
    res["batch_1M"] = {"W": 8, "batch_pairs": B1, "ranks": 1, "seconds": round(dt, 4), "em_iters": rep1["iters"], "eq_classes": rep1["eq_classes"],
                       "num_reads_ge_0.01": rel_spread(base[0], al1, 1e-2), "num_reads_ge_10": rel_spread(base[0], al1, 10.0),
                       "tpm_ge_0.01": rel_spread(base[1], tp1, 1e-2), "tpm_ge_1": rel_spread(base[1], tp1, 1.0)}
    # two ranks (two contexts on this device), SPEC MG: the shared burn-in prefix on both, rank 1 drops its counts, the rest alternates;
    # the tables go through the communicator's buffers in loop-back
    try:
        o = api.quant_opts(); ctxs = [api.QuantContext(idx, o, device=local, max_batch_reads=B1) for _ in range(2)]
        for c in ctxs: c.reserve(2000000, 0)
        rb_all = rbs_of(B1); d = api.Dist(api.Dist.make_id(), 0, 1, local)
        import torch; torch.cuda.synchronize(); t0 = time.perf_counter(); npfx = 0
        while npfx < len(rb_all) and not ctxs[0].summary()["burned_in"]:
            for c in ctxs: c.map_batch(rb_all[npfx], fetch=False); c.eq_accumulate()
            npfx += 1
        ctxs[1].drop_counts()
        for i, rb in enumerate(rb_all[npfx:]):
            c = ctxs[i % 2]; c.map_batch(rb, fetch=False); c.eq_accumulate()
        models = [c.model() for c in ctxs]
        d.merge_eq_loopback(ctxs)
        eq = ctxs[0].eq_finish()
        uq = (models[0][1] + models[1][1]).astype(np.uint64); tc = (models[0][2] + models[1][2]).astype(np.uint64)
        allm = np.ascontiguousarray(np.stack([m[0] for m in models])); lm = np.zeros(allm.shape[1])
        capi.check(capi.lib().sq_merge_log_masses(allm.shape[1], 2, allm.ctypes.data, lm.ctypes.data), "sq_merge_log_masses")
        le = models[0][3]
        proj = api.normalize_alphas(eq, lm, uq, tc); eff = np.exp(le)
        al, rep = ctxs[0].em_optimize(eff, proj, api.em_opts())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0; tp = tpm_of(al, eff)
        res["ranks_2"] = {"W": 8, "batch_pairs": B1, "ranks": 2, "shared_prefix_batches": npfx, "seconds": round(dt, 4), "em_iters": rep["iters"], "eq_classes": len(eq.count),
                          "class_table_equals_one_rank_job": eq_digest(eq) == tables["one_rank_1M"],
                          "vs_one_rank_same_batches": {"num_reads_ge_10": rel_spread(al1, al, 10.0), "tpm_ge_1": rel_spread(tp1, tp, 1.0)},
                          "num_reads_ge_0.01": rel_spread(base[0], al, 1e-2), "num_reads_ge_10": rel_spread(base[0], al, 10.0),
                          "tpm_ge_0.01": rel_spread(base[1], tp, 1e-2), "tpm_ge_1": rel_spread(base[1], tp, 1.0)}
        d.free()
        for c in ctxs: c.free()
    except Exception as e:
        res["ranks_2"] = {"error": str(e)[:300]}
    return {"pairs": N, "what": "relative difference |x - y| / max(x, y) of NumReads and TPM against the default (W = 8 mini-batches per model snapshot, one rank, "
            "%d-pair batches), over the transcripts where either side reaches the floor; VBEM to convergence in every variant" % B, "variants": res}


def launch_command(a, argv, env):
    """The command that runs this script as `--gpus N` ranks: None when the ranks exist already (a launcher set WORLD_SIZE) or one rank is asked for."""
    if "WORLD_SIZE" in env or a.gpus <= 1: return None
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.abspath(__file__)] + [x for x in argv if x != "--dry-launch"]


def main():
    a = parse()
    # ONE line on stdout: the communicator libraries print a version banner to the C-level stdout when the first communicator is made (RCCL does,
    # with one rank too) — everything but the JSON line is sent to stderr, by descriptor, for the life of the process
    cmd = launch_command(a, sys.argv[1:], os.environ)
    if a.dry_launch:
        print(json.dumps({"relaunch": cmd})); return
    if cmd is not None:        # [r6] `python bench.py --gpus N` outside a launcher: N ranks are started here, one per GPU (the driver's own N > 1 command is this one)
        sys.stdout.flush(); os.execv(cmd[0], cmd)
    sys.stdout.flush(); real_stdout = os.dup(1); os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): the line would not describe the run" % (a.gpus, world))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if a.debug_one_device:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        if a.debug_one_device: dist.init_process_group("gloo")
        else: dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from salmon_amd import api, synth, capi
    sqd = None
    if dist is not None and not a.debug_one_device and not a.py_dist:
        # the product's multi-GPU seam: an RCCL communicator owned by libsalmon_hip.so (hip/dist.hip); torch.distributed only carries the
        # 128-byte unique id to the ranks and keeps the barrier / timing contract of this script
        idt = torch.zeros(128, dtype=torch.uint8, device=torch.device("cuda", local))
        if rank == 0: idt.copy_(torch.frombuffer(bytearray(api.Dist.make_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        sqd = api.Dist(bytes(idt.cpu().numpy().tobytes()), rank, world, local)
    ncores = os.cpu_count() or 8
    thr = max(4, ncores // max(1, world))
    out = run_workload(a, a.workload, rank, world, local, dist, sqd, thr, ncores, api, synth, capi, extras=not a.no_extras)
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


def run_workload(a, wl, rank, world, local, dist, sqd, thr, ncores, api, synth, capi, extras=True, leg=False):
    """One workload: index, parked reads, warm-up, the timed job; on rank 0 the JSON object.  `leg` = a secondary leg inside another
    workload's run (c2s beside c2): no CPU sample, no FASTQ pass, no extras."""
    import torch
    w = WORKLOADS[wl]
    # a leg keeps its own workload's shape, scaled like the caller's (`--genes 600` for a test shrinks both)
    genes, iso, gnt = (a.genes, a.iso, a.genome_gnt) if not leg else (max(50, w["genes"] * a.genes // WORKLOADS[a.workload]["genes"]), w["iso"], w["genome_gnt"])
    genomic = a.genomic if not leg else w["genomic"]
    strong = wl == "c3"
    Wd = World(a, genes, iso, gnt, genomic, thr, local, api, synth)
    tx, genome, idx = Wd.tx, Wd.genome, Wd.idx
    B = a.batch; K = a.steps; W = a.warmup; RL = a.read_len if not leg else w["read_len"]; S = max(1, a.sub)
    if B > (1 << 23): raise SystemExit("--batch above 2^23 pairs per sq_map_batch call")
    total_pairs = K * S * B            # c3: the whole job's pairs, whatever the rank count
    if strong:
        # [r4] SPEC MG: the job's batch plan does not depend on the rank count.  The shared burn-in prefix — batches of <= 1 M pairs covering the
        # first ~5.5 M pairs (numAuxModelSamples = 5 M assigned fragments at >= 92 % mapped) — is mapped by EVERY rank; the rest, in batches of
        # <= 2 M pairs, is dealt out (batch j -> rank j mod R).  With one rank this is simply the job in that batch order.
        PB = min(B, 1000000); RB = min(B, 2000000)
        prefix_pairs = min(total_pairs, 6 * PB if total_pairs > 6 * PB else total_pairs)
        plan = [(i * PB, min(PB, prefix_pairs - i * PB), True) for i in range(-(-prefix_pairs // PB))]
        nrest = -(-(total_pairs - prefix_pairs) // RB)
        plan += [(prefix_pairs + j * RB, min(RB, total_pairs - prefix_pairs - j * RB), False) for j in range(nrest) if j % world == rank]
    else:
        plan = [(((rank * (K + W) + W) * S + i) * B, B, False) for i in range(K * S)]
    call_sizes = [n for _, n, _ in plan]
    opts = api.quant_opts()
    if a.inflight > 0: opts.mini_batches_in_flight = a.inflight
    ctx = api.QuantContext(idx, opts, device=local, max_batch_reads=B)
    # end-of-job buffers (eq-class export, EM workspace) sized like the reference's initial eq-class map (10^6 classes); with several
    # ranks every GPU ends up holding the union of all ranks' classes, so the class table is sized for that
    ctx.reserve(1000000 * max(1, world // 2), 0)
    # synthetic reads generated on the host, parked in HBM.  Weak scaling: rank r, step s, call u -> pairs [(((r*(K+W))+s)*S+u)*B, ...);
    # strong (c3): the timed calls cut the job's pair range [0, total) by rank, the warm-up calls lie beyond it
    dev = torch.device("cuda", local)
    off_np = (np.arange(0, 2 * B + 1, dtype=np.int64) * RL)
    off_d = torch.from_numpy(off_np).to(dev)
    batches = []; sizes = []
    host_first = None
    t_gen0 = time.time()
    def park(first, n):
        seq = Wd.reads(n, RL, first)
        batches.append(torch.from_numpy(seq).to(dev)); sizes.append(n)
        return seq
    for s in range(W * S):
        park((total_pairs + (rank * W * S + s) * B) if strong else ((rank * (K + W)) * S + s) * B, B)
    for i, (first, n, _) in enumerate(plan):
        seq = park(first, n)
        if i == 0 and rank == 0 and not leg and a.cpu_sample > 0:
            host_first = seq[: 2 * RL * min(n, a.cpu_sample)].copy()
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen0
    rbs = [api.make_read_batch(int(b.data_ptr()), int(off_d.data_ptr()), n, paired=True, on_device=True) for b, n in zip(batches, sizes)]
    # ---- warmup (sizes every work buffer; model/eq state is reset afterwards) ----
    for s in range(W * S):
        ctx.map_batch(rbs[s], fetch=False); ctx.eq_accumulate()
    if a.lanes > 1:
        for s in range(min(W * S, a.lanes)):   # size the work buffers of every lane
            ctx.map_submit(rbs[0])
        for s in range(min(W * S, a.lanes)):
            ctx.map_wait()
    if W:
        e = ctx.eq_finish()
        api.em_steps(e, idx.ref_lens().astype(np.float64), np.full(idx.num_refs, 100.0), 2, api.em_opts(), device=local)
    ctx.reset()
    ctx.set_profiling(not a.no_profile)
    ctx.stage_times(reset=True)
    capi.lib().sq_ctx_seed_filter_fills(ctx.h, 1)
    M = idx.num_refs
    # ---- timed region: exactly K steps + the job's inference tail ----
    if dist: dist.barrier()
    torch.cuda.synchronize()
    cs0 = cpu_stat(); ev0 = kfd_evicted_ms()
    t0 = time.perf_counter()
    tot = None; t_prefix = 0.0; prefix_burned = None
    depth = max(1, a.lanes)          # batches in flight on the mapping lanes (sq_map_submit / sq_map_wait)
    lo, hi = W * S, len(rbs)
    if depth > 1:
        for s in range(lo, min(hi, lo + depth)):
            ctx.map_submit(rbs[s])
    for s in range(lo, hi):
        if depth > 1:
            _, _, _, st = ctx.map_wait()
        else:
            _, _, _, st = ctx.map_batch(rbs[s], fetch=False)
        ctx.eq_accumulate()
        if depth > 1 and s + depth < hi:
            ctx.map_submit(rbs[s + depth])
        tot = st if tot is None else {k: tot[k] + v for k, v in st.items()}
        if strong and plan[s - lo][2] and (s + 1 == hi or not plan[s + 1 - lo][2]):      # the last batch of the shared prefix
            t_prefix = time.perf_counter() - t0; prefix_burned = ctx.summary()["burned_in"]
            if rank != 0: ctx.drop_counts()
    t_map = time.perf_counter() - t0
    eq = ctx.eq_finish()
    t_eqf = time.perf_counter() - t0 - t_map
    lm, uq, tc, le = ctx.model()
    t_merge = 0.0
    if dist:  # one RCCL all-gather of the packed tables; every rank merges the others' tables exactly (integer sums)
        t_a = time.perf_counter()
        from salmon_amd import dist as sqdist
        cdev = torch.device("cpu") if a.debug_one_device else dev
        if a.debug_one_device:      # gloo has no device collectives: host tables
            tables = sqdist.all_gather_tables(eq, dist, cdev)
            for r in range(world):
                if r != rank:
                    ctx.eq_merge(tables[r])
        elif sqd is not None:       # sq_dist_merge_eq: one packed all-gather over RCCL (HBM -> xGMI -> HBM), merged where it lands
            sqd.merge_eq(ctx)
        else:
            sqdist.merge_all_device(ctx, dist, dev)
        eq = ctx.eq_finish()
        if sqd is not None: lm, uq, tc, le = sqd.reduce_model(lm, uq, tc, le)      # SPEC MG
        else: lm, uq, tc, le = sqdist.reduce_model(lm, uq, tc, le, dist, cdev)
        t_merge = time.perf_counter() - t_a
    t_a = time.perf_counter()
    proj = api.normalize_alphas(eq, lm, uq, tc)
    t_norm = time.perf_counter() - t_a
    eff = np.exp(le)
    t_a = time.perf_counter()
    alphas, rep = ctx.em_optimize(eff, proj, api.em_opts())   # the ctx's own classes (incl. merged ones), read from the export resident in HBM
    t_em = time.perf_counter() - t_a
    gibbs = None
    if wl == "c5":   # configs[4]: posterior samples by collapsed Gibbs, whole chains sharded by rank (sq_dist_share)
        t_a = time.perf_counter()
        nmapped = int(eq.count.sum()); Sn = a.gibbs_samples
        first, cnt = (sqd.share(Sn, capi.lib().sq_gibbs_chain_step(Sn)) if sqd is not None else (0, Sn))
        gs, grep = api.gibbs_range(eq, eff, alphas, Sn, first, cnt, 2024, nmapped, api.gibbs_opts(), device=local, report=True)
        t_g = time.perf_counter() - t_a
        gibbs = {"samples": Sn, "samples_this_rank": int(cnt), "seconds": round(t_g, 4), "report": grep,
                 "mean_abs_dev_from_vbem": float(np.mean(np.abs(gs.mean(axis=0) - alphas))) if len(gs) else None}
    torch.cuda.synchronize()
    if dist: dist.barrier()
    t1 = time.perf_counter()
    dt = t1 - t0
    cs1 = cpu_stat(); ev1 = kfd_evicted_ms()
    host_cpu = {"cores_used": round((cs1["usage_usec"] - cs0["usage_usec"]) / (dt * 1e6), 2), "quota_periods_throttled": cs1["nr_throttled"] - cs0["nr_throttled"],
                "throttled_ms": round((cs1["throttled_usec"] - cs0["throttled_usec"]) / 1e3, 2)} if cs0 and cs1 else None
    if dist:
        tt_ = torch.tensor([dt], device=(torch.device("cpu") if a.debug_one_device else dev), dtype=torch.float64)
        dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
        dt = float(tt_.item())
    stages = ctx.stage_times()
    tot["filter_fills"] = int(capi.lib().sq_ctx_seed_filter_fills(ctx.h, 0))
    # EM iteration rate from a fixed-count run on the final table (outside the timed region)
    _, rep_it = api.em_steps(eq, eff, np.maximum(alphas, 1e-3), 200, api.em_opts(), device=local)
    if rank != 0:
        ctx.free(); Wd.free()
        return None
    E = len(eq.count); Lb = len(eq.tid)
    em_bytes = 36 * Lb + 16 * E + 64 * M
    em_gbs = em_bytes / (rep_it["ms_per_iter"] * 1e-3) / 1e9
    NP = sum(call_sizes)               # pairs this rank mapped in the timed region
    job_pairs = total_pairs if strong else world * NP
    sb = stage_bytes(tot, NP, RL)
    stage_rows = {k: {"ms_total": round(v[0], 3), "launches": v[1], "avg_ms": round(v[0] / max(1, v[1]), 4),
        "alg_GBps": round(sb.get(k, 0) / max(v[0], 1e-9) / 1e6, 1)} for k, v in stages.items() if v[1]}
    # roofline: the stage kernel with the largest total time in the timed region — every stage is a candidate, the online model's mini-batch
    # chain included (its row aggregates one launch pair per group of mini-batches)
    pm = {}; pm_note = "no PMC profile committed for this kernel on this workload"
    try:   # the counter passes were taken on the c2 workload: no traffic figure for the others; [r4] nor when the kernel sources changed since
        if wl == "c2":   # (c3 maps the same index in other batch sizes: not the launches that were counted)
            prof = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")))
            if prof.get("kernel_source_sha") == kernel_source_sha() and prof.get("pairs_per_launch") == B: pm = prof["kernels"]
            else: pm_note = "profiles/r06_pmc_traffic.json was taken on other kernel sources or another batch size (sha %s, %s pairs): no traffic figure attached" % (prof.get("kernel_source_sha"), prof.get("pairs_per_launch"))
    except Exception: pass
    cand = [k for k in stage_rows if k in KERNEL_OF_STAGE]
    roof = None; roofs = {}
    # the online chain's row is a launch PAIR per group of mini-batches (k_frag_dynamic, k_apply_flagged): its time is split between the two
    # kernels in the proportion the committed rocprofv3 summary shows (50/50 without it)
    pair_share = {"k_frag_dynamic": 0.5, "k_apply_flagged": 0.5}; pair_note = "no committed rocprofv3 summary found: the launch pair's time is split 50/50"
    try:
        ktot = {}
        for line in open(os.path.join(ROOT, "profiles", "r06_kernel_stats_c2_last.txt")):
            for kn in pair_share:
                if kn + "(" in line and "total=" in line: ktot[kn] = float(line.split("total=")[1].split("ms")[0])
        if len(ktot) == 2:
            pair_share = {kn: ktot[kn] / sum(ktot.values()) for kn in ktot}
            pair_note = "split %.2f / %.2f between k_frag_dynamic and k_apply_flagged as in profiles/r06_kernel_stats_c2_last.txt" % (pair_share["k_frag_dynamic"], pair_share["k_apply_flagged"])
    except Exception: pass
    for k in cand:
        per_launch = sb[k] / max(1, stage_rows[k]["launches"])
        parts = [(KERNEL_OF_STAGE[k], 1.0)] if k != "eq_mini_batches" else [(kn, sh) for kn, sh in pair_share.items()]
        for kn, sh in parts:
            ms_total = stage_rows[k]["ms_total"] * sh; avg_ms = stage_rows[k]["avg_ms"] * sh
            ach = per_launch * sh / (avg_ms * 1e-3) / 1e9
            kk = pm.get(kn); traffic = None
            if kk and kk.get("fetch_bytes_per_launch") is not None:
                traffic = int(kk["fetch_bytes_per_launch"] + (kk.get("write_bytes_per_launch") or 0))
            roofs[kn] = {"kernel": kn, "bound": "hbm", "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 5),
                         "traffic": traffic, "avg_launch_ms": round(avg_ms, 4), "alg_bytes_per_launch": int(per_launch * sh), "ms_total": round(ms_total, 3),
                         # second denominator: the measured rate of random 8-byte loads of 64-byte lines on this chip (tools/gather_bench.hip,
                         # profiles/r03_gather_bench.txt: 2.75 TB/s at 8 blocks per CU over a 1 GB table)
                         "frac_of_random_sector_ceiling_2750GBps": round(ach / 2750.0, 5)}
    if roofs:
        dom = max(roofs, key=lambda k: roofs[k]["ms_total"])
        roof = dict(roofs[dom])
        roof["traffic_note"] = ("FETCH_SIZE + WRITE_SIZE per launch from profiles/r06_pmc_traffic.json (rocprofv3 --pmc, separate passes, same workload and batch size, "
                                "kernel sources unchanged since: sha %s; calibrated on tools/gather_bench: no correction for this access pattern); PMC cannot be sampled inside the timed run" % kernel_source_sha()) if roof["traffic"] is not None else pm_note
        roof["alg_bytes_note"] = "per-kernel byte model = bench.py::stage_bytes (DESIGN.md section 5)"
        roof["chain_pair_note"] = pair_note
        if dom == "k_seed2" and "k_seed_r4model" in sb:   # the round-4 byte model of the kernel k_seed2 replaced, for comparison across rounds only
            r4 = sb["k_seed_r4model"] / max(1, stage_rows["k_seed"]["launches"])
            roof["round4_model"] = {"alg_bytes_per_launch": int(r4), "achieved": round(r4 / (roof["avg_launch_ms"] * 1e-3) / 1e9, 2), "frac": round(r4 / (roof["avg_launch_ms"] * 1e-3) / 1e9 / 8000.0, 5),
                                    "note": "one filter sector per probe and four dependent sectors per hit — the bytes rounds 3-5 quoted `frac` on; k_seed2 moves fewer (frac above is on its own bytes)"}
            roof["traffic_over_alg_bytes"] = round(roof["traffic"] / roof["alg_bytes_per_launch"], 3) if roof.get("traffic") else None
        if dom == "k_seed2" and wl == "c2":   # SQ counters of the same kernel on this workload (profiles/r06_pack_counters.txt, rocprofv3 --pmc, call ZN): how busy the vector issue ports are
            roof["valu_issue"] = {"vector_instructions_per_launch": 1391333484, "busy_cycles_per_shader_engine": 9044344, "simds": 768, "frac_of_issue_cycles": round(1391333484 * 4 / 768 / 9044344.0, 3),
                                  "waiting_for_memory_frac_of_wave_cycles": round(2030252060 / 10282191389.0, 3),
                                  "note": "four cycles per wave64 vector instruction on a 16-lane SIMD; from the committed counter profile (5 M pairs per launch), not sampled in this run: the kernel is instruction-issue-bound at least as much as sector-rate-bound (DESIGN.md section 6)"}
        roof["all_kernels"] = {k: {"frac": roofs[k]["frac"], "ms_total": roofs[k]["ms_total"], "avg_launch_ms": roofs[k]["avg_launch_ms"]} for k in roofs}
    if gibbs is not None:   # c5 is inference-bound: its dominant kernel is the Gibbs round
        g = gibbs["report"]; bg = 28 * Lb + 16 * E + 32 * M
        if g.get("rounds"):
            ach = bg / (g["ms_per_round"] * 1e-3) / 1e9
            gibbs["roofline"] = {"kernel": "k_gibbs_round (mu + per-class multinomials + counts)", "bound": "hbm", "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s",
                                 "frac": round(ach / 8000.0, 5), "alg_bytes_per_round": bg, "rounds_per_s": round(1e3 / g["ms_per_round"], 1),
                                 "draws_per_s": round(g.get("draws_per_round", 0) / (g["ms_per_round"] * 1e-3), 1),
                                 "note": "B_gibbs = 28 L + 16 E + 32 M per round (SURVEY 8d); the working set is cache-resident and a round's time is its N categorical draws"}
    cpu = None; parity = None
    if a.cpu_sample > 0 and world == 1 and host_first is not None:
        try:   # the checker's leg: its failure must not cost the bench line (cpu_baseline / parity_check then say what went wrong)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import orc
            Sn = min(sizes[W * S], a.cpu_sample)
            c_a = time.perf_counter()
            oidx = orc.OrcIndex(idx)
            t_oidx = time.perf_counter() - c_a
            soff = (np.arange(0, 2 * Sn + 1, dtype=np.uint64) * np.uint64(RL))
            rb = api.make_read_batch(host_first, soff, Sn, paired=True)
            c0 = time.perf_counter()
            ro, aln, mt, stc = orc.map_batch(oidx, opts, rb, threads=ncores)
            c1 = time.perf_counter()
            ost = orc.OrcState(oidx, opts); ost.eq_accumulate(ro, aln, stc["num_with_joint_hits"]); ost.finish(); eqc = ost.eq_finish()
            lmc, uqc, tcc, lec, _ = ost.model()
            pc = orc.normalize_alphas(M, eqc, lmc, uqc, tcc)
            c2 = time.perf_counter()
            a_c, repc = orc.em_optimize(eqc, np.exp(lec), pc, api.em_opts())
            c3 = time.perf_counter()
            # the checker's EM loop is single-threaded (order-defined sums); its multi-threaded iteration (same arithmetic,
            # transcripts split over threads) is timed separately and used for the composite so the CPU gets its cores
            em_single_s = c3 - c2
            em_try = {t: orc.em_time_iters(eqc, np.exp(lec), 20, t) / 20.0 * repc["iters"] for t in sorted({min(ncores, 8), min(ncores, 32), ncores})}
            em_thr_n, em_thr_s = min(em_try.items(), key=lambda kv: kv[1])
            if em_single_s < em_thr_s: em_thr_n, em_thr_s = 1, em_single_s       # the CPU side gets its best configuration
            em_cpu_s = orc.em_time_iters(eq, eff, 20, ncores) / 20.0
            t_cpu = (c1 - c0) + (c2 - c1) + em_thr_s
            # parity at bench scale (outside the timed region): the same pairs through the HIP path on a reset context, compared with
            # what the checker just produced — alignment records, per-read offsets, mapping types, counters, the class table (labels,
            # bins, counts, fixed-point weight sums), the online model, projected counts and the VBEM result
            ctx.set_profiling(False); ctx.reset()
            ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
            ctx.eq_accumulate(); eq_g = ctx.eq_finish(); lm_g, uq_g, tc_g, le_g = ctx.model()
            p_g = api.normalize_alphas(eq_g, lm_g, uq_g, tc_g)
            a_g, rep_g = ctx.em_optimize(np.exp(le_g), p_g, api.em_opts())
            checks = {"alignments": sha(aln_g) == sha(aln), "read_offsets": sha(ro_g) == sha(ro), "map_types": sha(mt_g) == sha(mt), "counters": st_g == stc,
                "eq_classes": all(sha(getattr(eq_g, f)) == sha(getattr(eqc, f)) for f in ("off", "tid", "bins", "count", "wq", "h1", "h2")),
                "online_model": sha(lm_g) == sha(lmc) and sha(uq_g) == sha(uqc) and sha(tc_g) == sha(tcc) and sha(le_g) == sha(lec),
                "projected_counts": sha(p_g) == sha(pc), "vbem": rep_g["iters"] == repc["iters"] and sha(a_g) == sha(a_c)}
            parity = {"pairs": Sn, "transcripts": int(M), "alignments": int(len(aln)), "eq_classes": int(len(eqc.count)), "equal": all(checks.values()),
                "checks": checks, "alignments_sha256": sha(aln_g), "decoy_fragments": int(st_g["num_decoy_fragments"]),
                "what": "first %d pairs of timed step 0: HIP path vs CPU checker, sha256 of every output array" % Sn}
            if Sn >= 2000000 and a.chain_check:
                # [r6] the burned-in mini-batch chain (k_chain: one launch per mapped batch, its workgroups behind a counter barrier) at this index's size and beside
                # running mapping kernels: the sample's first 2 M pairs as four batches with the burn-in moved into the second, HIP path vs checker, the model bit for bit
                nb_, bp_ = 4, 500000; o2 = api.quant_opts(num_burnin_frags=600000)
                cx2 = api.QuantContext(idx, o2, device=local, max_batch_reads=bp_); cx2.reserve(1000000, 0); os2 = orc.OrcState(oidx, o2)
                for i_ in range(nb_):
                    rb_ = api.make_read_batch(host_first[i_ * bp_ * 2 * RL:(i_ + 1) * bp_ * 2 * RL], soff[:2 * bp_ + 1], bp_, paired=True)
                    cx2.map_batch(rb_); cx2.eq_accumulate()
                    ro_, aln_, mt_, st_ = orc.map_batch(oidx, o2, rb_, threads=ncores); os2.eq_accumulate(ro_, aln_, st_["num_with_joint_hits"])
                os2.finish(); m_g = cx2.model(); m_c = os2.model(); e_g = cx2.eq_finish(); e_c = os2.eq_finish()
                parity["checks"]["burned_in_chain_4x500k"] = bool(cx2.summary()["burned_in"]) and all(sha(x) == sha(y) for x, y in zip(m_g, m_c[:4])) and \
                    all(sha(getattr(e_g, f)) == sha(getattr(e_c, f)) for f in ("off", "tid", "bins", "count", "wq"))
                parity["equal"] = all(parity["checks"].values()); cx2.free(); os2.free(); del cx2, os2
            if gibbs is not None and Sn >= 1000:   # c5: the Gibbs sampler on the sample's classes, every sample byte-equal to the checker's
                g_g = api.gibbs(eq_g, np.exp(le_g), a_g, 8, 7, int(eq_g.count.sum()), api.gibbs_opts(), device=local)
                g_c = orc.gibbs(eqc, np.exp(lec), a_c, 8, 7, int(eqc.count.sum()), api.gibbs_opts())
                parity["checks"]["gibbs_8_samples"] = sha(g_g) == sha(g_c); parity["equal"] = all(parity["checks"].values())
            hw_thr, quota = cpu_allowance()
            cpu = {"value": round(Sn / t_cpu / 1e6, 4), "unit": "M read-pairs/s", "cores": int(round(min(ncores, quota))) if quota else ncores, "kind": "port",
                   "threads_started": ncores, "cpu_quota_cores": quota,
                   "sample": "%d of the %d pairs of step 0 through the CPU checker (oracle/): map %.2fs (%d threads) + online model / eq-classes %.2fs (1 thread: the mini-batch chain is sequential) + VBEM %d iters %.2fs (best of 1/8/32/%d threads: %d); %.1fs of CPU work in all; checker index built in %.1fs (not counted)" % (Sn,
                       sizes[W * S], c1 - c0, ncores, c2 - c1, repc["iters"], em_thr_s, ncores, em_thr_n, t_cpu, t_oidx),
                   "map_only_M_pairs_per_s": round(Sn / (c1 - c0) / 1e6, 4), "em_iters_per_s_full_table_%dthr" % ncores: round(1.0 / em_cpu_s, 2),
                   # what the sample's rates would mean for the whole timed job (a model, not a measurement): per-pair costs scale with the pairs,
                   # the EM runs once over the full table for as many iterations as the GPU job needed
                   "extrapolated_full_job_M_pairs_per_s": round(NP / (NP * ((c1 - c0) + (c2 - c1)) / Sn + rep["iters"] * em_cpu_s) / 1e6, 4)}
            del oidx, ost
        except Exception as e:
            cpu = cpu or {"error": str(e)[:300]}; parity = parity or {"error": str(e)[:300]}
    jobs = {}; spread = None; c2s = None
    if K * S * B == 100000000 and world == 1: jobs["100M"] = {"value": round(NP / dt / 1e6, 4), "pairs": NP, "seconds": round(dt, 4), "what": "the timed region of this line"}
    if extras and not leg and world == 1 and wl == "c2":
        # configs[1] as stated: the SAME job (map + online model + eq-classes + export + normalizeAlphas + VBEM to convergence) on the first 10 M pairs
        n10 = max(1, min(K * S, 10000000 // B))
        try:   # an extra: never lose the bench line over it
            d10, al10, eff10, rep10 = one_job(ctx, rbs[W * S: W * S + n10], api, idx)
            jobs["10M" if n10 * B == 10000000 else "%d" % (n10 * B)] = {"value": round(rep10["pairs"] / d10 / 1e6, 4), "pairs": rep10["pairs"], "seconds": round(d10, 4), "map_eq_s": round(rep10["map_eq_s"], 4),
                        "em_iters": rep10["iters"], "eq_classes": rep10["eq_classes"], "what": "configs[1] as stated: the whole job on the first %d pairs, timed the same way (outside the timed steps of this line)" % rep10["pairs"]}
        except Exception as e: jobs["10M"] = {"error": str(e)[:300]}
        if a.spread_pairs > 0:
            try: spread = run_spread(a, Wd, batches[W * S:], off_d, B, RL, api, capi, local)
            except Exception as e: spread = {"error": str(e)[:300]}
    fq = None
    if a.fastq_pairs > 0 and world == 1 and wl in ("c2", "c2s") and not leg:
        try:   # an extra (and one that needs room for its files): never lose the bench line over it
            fq = run_from_fastq(ctx, idx, tx, a.fastq_pairs, RL, min(B, 1000000), min(thr, 64), api, capi, gz_pairs=a.fastq_gz_pairs)
        except Exception as e: fq = {"error": str(e)[:300]}
    cfg_name = {"c2": "configs[1]: human-transcriptome-shaped synthetic index (60k genes x ~4 isoforms; SURVEY C2 shape)",
                "c3": "configs[2]: human-transcriptome-shaped synthetic index (as c2), a fixed total of %d pairs split over %d rank(s)" % (total_pairs, world),
                "c2s": "configs[1], round-1/2 index (T200k: 20k genes x ~10 isoforms, 54 M distinct k-mers)",
                "c5": "configs[4]: human-shaped index, 50 M pairs, VBEM + %d Gibbs samples" % a.gibbs_samples,
                "c4": "configs[3]: decoy-aware index (human-shaped txome + %.2f Gnt synthetic genome as decoys), 2x%d bp, %.0f %% genomic pairs" % (gnt, RL, 100 * genomic)}[wl]
    out = {
        "metric": baseline_metric(), "value": round(job_pairs / dt / 1e6, 4), "unit": "M read-pairs/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 3),
        "rccl_ranks": int(capi.lib().sq_dist_world(sqd.h)) if sqd is not None else None,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "u64/i32 (2-bit k-mers, integer scores) + f64 (log-space model, EM)", "data": "synthetic",
        "config": {"workload": "%s (k=31, m=20), %s, -l IU defaults, VBEM" % (cfg_name, ("%d synthetic 2x%dbp pairs in all; every rank maps the shared burn-in prefix (%d pairs), the rest is dealt out: %d pairs on this rank in %d calls" % (total_pairs, RL,
                           sum(n for _, n, sh in plan if sh), NP, len(call_sizes))) if strong
                       else ("%d steps x %d x %d = %d synthetic 2x%dbp pairs per GPU" % (K, S, B, NP, RL))),
                   "workload_id": wl,
                   "transcripts": int(tx.n), "refs": int(M), "txome_nt": int(tx.total_nt()), "decoy_nt": int(genome.total_nt()) if genome is not None else 0,
                   "distinct_kmers": int(idx.num_kmers), "unitigs": int(idx.num_unitigs), "index_hbm_bytes": int(idx.device_bytes),
                   "pairs_per_step": S * B, "pairs_per_map_call": B, "job_pairs": int(job_pairs),
                   "parallelism": "reads sharded over %d GPU(s); eq-class tables all-gathered + merged exactly (%s); EM replicated" % (world,
                       "sq_dist_* over RCCL" if sqd is not None else ("torch.distributed" if dist is not None else "single rank"))},
        "breakdown": {"map_eq_s": round(t_map, 4), "shared_prefix_s": round(t_prefix, 4) if strong else None, "shared_prefix_pairs": int(sum(n for _, n, sh in plan if sh)) if strong else None,
            "model_burned_in_at_prefix_end": prefix_burned, "tail_s(eq_export+merge+normalize+EM%s)" % ("+Gibbs" if gibbs else ""): round(dt - t_map, 4), "eq_finish_s": round(t_eqf, 4),
            "dist_merge_s": round(t_merge, 4), "normalize_alphas_s": round(t_norm, 4), "em_call_s": round(t_em, 4), "em_iters": rep["iters"], "em_converged": rep["converged"],
            "em_device_ms": round(rep["device_ms"], 2), "host_cpu_in_timed_region": host_cpu, "kfd_queues_evicted_ms_in_timed_region": (ev1 - ev0) if ev0 is not None and ev1 is not None else None, "synth_s": round(Wd.t_synth, 1), "read_gen_and_park_s": round(t_gen, 1),
                      "index_build_s": round(Wd.t_index, 1), "mapped_frac": round(tot["num_mapped"] / tot["num_reads"], 4),
                      "decoy_frac": round(tot["num_decoy_fragments"] / tot["num_reads"], 4),
                      "hits_per_frag": round(tot["num_alignments"] / max(1, tot["num_mapped"]), 3),
                      "eq_classes": E, "label_entries": Lb, "stats": tot},
        "em": {"iters_per_s": round(1e3 / rep_it["ms_per_iter"], 1), "ms_per_iter": round(rep_it["ms_per_iter"], 4), "alg_bytes_per_iter": em_bytes,
            "alg_GBps": round(em_gbs, 1), "frac_of_8TBps": round(em_gbs / 8000.0, 4)},
        "gibbs": gibbs,
        "stages": stage_rows, "roofline": (gibbs or {}).get("roofline") if wl == "c5" and gibbs and gibbs.get("roofline") else roof,
        "mapping_roofline": roof if wl == "c5" else None,
        "cpu_baseline": cpu, "parity_check": parity, "from_fastq": fq, "jobs": jobs or None, "spread": spread,
    }
    # DESIGN.md section 8's arithmetic on THIS run's own components, for every N a SCALE run takes: weak scaling keeps the mapping time and pays one exchange of the class
    # tables (a table of N shards' classes also lengthens the replicated EM: not modelled — the figure is an upper bound); strong scaling (--workload c3) divides what
    # lies behind the shared burn-in prefix.  A measured line at N ranks is to be read against "N" here.
    try:
        t_tail = dt - t_map; merge = {1: 0.0, 2: 0.003, 4: 0.004, 8: 0.005}
        if strong:
            t_rest1 = (t_map - t_prefix) * world   # what lies behind the prefix, summed over the ranks
            pred = {str(n): round(total_pairs / 1e6 / (t_prefix + t_rest1 / n + merge[n] + (t_tail - t_merge)), 1) for n in (1, 2, 4, 8)}
        else:
            pred = {str(n): round(n * NP / 1e6 / (t_map + merge[n] + (t_tail - t_merge)), 1) for n in (1, 2, 4, 8)}
        out["scaling_prediction"] = {"unit": "M read-pairs/s", "by_n_gpus": pred, "from": "this run's prefix / mapping / tail times (DESIGN.md section 8); merge 3-5 ms assumed; the replicated EM on the merged table of N shards is not modelled (weak: upper bound)"}
    except Exception as e:
        out["scaling_prediction"] = {"error": str(e)[:200]}
    ctx.free(); del batches, rbs; Wd.free(); torch.cuda.empty_cache()
    if extras and not leg and world == 1 and wl == "c2":
        # the same job on the round-1/2 index (T200k: 20 000 genes x ~10 isoforms, 5.3 alignments per fragment) — the workload the headline
        # was quoted on before round 3, carried beside it so the two can be compared at equal job size
        try:
            leg_out = run_workload(a, "c2s", rank, world, local, None, None, thr, ncores, api, synth, capi, extras=False, leg=True)
            out["c2s"] = {k: leg_out[k] for k in ("value", "unit", "steps", "ms_per_step")}
            out["c2s"].update(workload=leg_out["config"]["workload"], distinct_kmers=leg_out["config"]["distinct_kmers"], hits_per_frag=leg_out["breakdown"]["hits_per_frag"],
                              map_eq_s=leg_out["breakdown"]["map_eq_s"], em_iters=leg_out["breakdown"]["em_iters"], em_ms_per_iter=leg_out["em"]["ms_per_iter"],
                              k_seed_frac=(leg_out["roofline"] or {}).get("all_kernels", {}).get("k_seed", {}).get("frac"))
        except Exception as e:
            out["c2s"] = {"error": str(e)[:300]}
    return out


if __name__ == "__main__":
    main()
