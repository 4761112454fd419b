#!/usr/bin/env python
"""bench.py — `salmon quant` hot path (map + eq-classes + EM) on MI355X, BASELINE.json metric.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path (sq_map_batch + sq_eq_accumulate) over one batch of synthetic
read pairs already resident in HBM.  The default N=1 workload is BASELINE.json configs[1]: a
human-transcriptome-shaped index (synthetic T200k: 20 000 genes x ~10 isoforms, k=31) and
K x batch = 10 x 4 000 000 = 40 M synthetic 2x100 bp pairs, followed by the job's inference tail
(eq-class export, normalizeAlphas, VBEM to convergence), all inside the timed region.  Weak scaling:
every rank maps its own K batches; eq-class tables are all-gathered over RCCL and merged exactly.
Prints ONE JSON line on rank 0.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4000000,
        help="read pairs per step (one sq_map_batch call); 4 x 10^6 amortises the tails of the persistent kernels: 65 vs 52 M pairs/s at 10^6 on MI355X")
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--iso", type=int, default=10)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--cpu-sample", type=int, default=200000, help="pairs timed through the CPU checker (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
        help="initialise torch.distributed (RCCL) even for one rank, so the N>1 code path (device-resident all_gather + merge) runs on a 1-GPU box")
    ap.add_argument("--debug-one-device", action="store_true",
        help="functional check of the N>1 path on a 1-GPU box: every rank uses cuda:0 and the collectives run over gloo (numbers meaningless)")
    ap.add_argument("--fastq-pairs", type=int, default=4000000,
        help="second measurement (outside the timed steps, rank 0, N=1): this many pairs written as plain FASTQ files to /dev/shm and run through "
             "the host read pipeline (sq_reader) + the same GPU path, end to end from files (0 = skip)")
    ap.add_argument("--py-dist", action="store_true",
        help="N>1: exchange the class tables with torch.distributed collectives (salmon_amd/dist.py) instead of the library's own RCCL path (sq_dist_*, the default)")
    ap.add_argument("--inflight", type=int, default=0,
        help="mini-batches per model snapshot (sq_quant_opts.mini_batches_in_flight = the reference's -p / numThreads); 0 = the library default (8)")
    ap.add_argument("--lanes", type=int, default=1,
        help="batches in flight on the mapping lanes (sq_map_submit/sq_map_wait); 1 = plain sq_map_batch. Measured on MI355X: 2 lanes shorten mapping (18.4 -> 17.4 ms per step) but the ordered online/eq chain (14.6 ms per step on its CU partition) then lags and the job does not finish sooner")
    return ap.parse_args()


# algorithmic bytes per unit for every stage (DESIGN.md §5 states the same table)
def stage_bytes(st, n_pairs, read_len, paired=True):
    nrec = 2 * n_pairs if paired else n_pairs
    L = read_len
    return {
        "k_pack": nrec * (L + 8 + 64 + 32 + 2),
        # every probe reads one word (a 64 B sector) of the k-mer membership filter; a probe that passes (every hit; < 1 % of the misses)
        # then walks 4 dependent sectors (pilot, slot record, string-pool word, unitig bounds) and a uni-MEM adds extension words,
        # contig-table bounds and its record
        "k_seed": nrec * (64 + 32 + 2 + 8) + st["num_lookups"] * 64 + st["num_seeds"] * (4 * 64 + 16 + 16 + 32 + 16),
        "scan_mems": nrec * (4 + 8),
        # fused projection + per-end sort + chaining (mem_kernels.h): uni-MEM records and contig-table runs in, sorted MEM records and chains out
        # (the uni-MEM record carries its contig-table run since round 2: 32 B each, no bounds gathers; 16 B list entry per end)
        "k_mems": st["num_seeds"] * 32 + st["num_mems"] * (8 + 8 + 16) + st["num_chains"] * 40 + nrec * (16 + 4),
        "k_join_fill": st["num_chains"] * 40 + st["num_candidates"] * 52 + n_pairs * 16,
        "k_score": st["num_candidates"] * (48 * 2 + 2 * 40 + 2 * (96 + 64) + 4) + st["num_mems"] * 0,
        "k_dp": st["num_dp_alignments"] * (48 + 96 + 64 + 2 * 40 + 2 * 4),   # + the queue read twice more and the permutation (counting sort by length)
        "k_select": st["num_candidates"] * (48 * 2) + st["num_alignments"] * 40 + n_pairs * 30,
        "compact_alns": st["num_alignments"] * 80 + n_pairs * 28,
        "eq_flags_scan": st["num_alignments"] * 40 + n_pairs * 24,
        "eq_mini_batches": st["num_alignments"] * (40 + 12 + 3 * 16) + n_pairs * 32,
        "eq_table": st["num_alignments"] * (40 + 12 + 8) + n_pairs * (16 + 4 + 32),
    }


def _fastq_pass(ctx, idx, files, batch, lib, api, capi, read_len):
    import ctypes as C
    ctx.reset()
    a1 = (C.c_char_p * 1)(files[0].encode()); a2 = (C.c_char_p * 1)(files[1].encode()); h = C.c_void_p()
    lanes = 2
    lib.sq_ctx_set_lanes(ctx.h, lanes)
    t0 = time.perf_counter()
    capi.check(lib.sq_reader_open(a1, 1, a2, 1, batch, lanes + 1, C.byref(h)), "sq_reader_open")
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
            "mapped_frac": round(tot_mapped / max(1, n), 4), "em_iters": rep["iters"], "input": "2 plain FASTQ files of %d x %d bp in /dev/shm (page cache), batches of %d pairs" % (n,
                read_len, batch), "host_threads": os.cpu_count(), "reader_threads": os.environ.get("SQ_READER_THREADS", "default: min(32, hw/2)"),
            "what": "end to end from files through sq_reader (mmap + parallel record split + page-locked batch assembly), H2D included; gzip input is bound by one inflate thread per mate file (~1.4 M pairs/s per file pair on this class of host)"}


def run_from_fastq(ctx, idx, tx, n_pairs, read_len, batch, threads, api, capi):
    """`salmon quant` from FASTQ files: sq_reader (parallel record splitting into page-locked batches) -> H2D -> mapping lanes -> online model /
    eq-classes -> export -> normalizeAlphas -> VBEM.  Wall time from opening the files to the converged alphas."""
    import ctypes as C, shutil, tempfile
    d = tempfile.mkdtemp(prefix="sq_bench_fq_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
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
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


def baseline_metric():
    """BASELINE.json's metric string, verbatim (one "read" there is one 2x100 bp pair; `unit` says so explicitly)."""
    try:
        return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")))["metric"]
    except Exception:
        return "M reads/s quantified (map+EM), 100M 2x100bp vs human txome; EM iters/s"


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
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
    t_setup = time.time()
    tx = synth.Txome(seed=1, n_genes=a.genes, iso_per_gene=a.iso, threads=min(thr, 32))
    names, seqs, lens = tx.tables()
    idx = api.SalmonIndex.build_mem_raw(tx.n, names, seqs, lens, threads=thr)
    t_index = time.time() - t_setup
    idx.to_device(local)
    B = a.batch; K = a.steps; W = a.warmup; RL = a.read_len
    opts = api.quant_opts()
    if a.inflight > 0: opts.mini_batches_in_flight = a.inflight
    ctx = api.QuantContext(idx, opts, device=local, max_batch_reads=B)
    # end-of-job buffers (eq-class export, EM workspace) sized like the reference's initial eq-class map (10^6 classes); with several
    # ranks every GPU ends up holding the union of all ranks' classes, so the class table is sized for that
    ctx.reserve(1000000 * max(1, world // 2), 0)
    # synthetic reads: rank r, step s -> pairs [((r*(K+W))+s)*B, ...), generated on the host, parked in HBM
    dev = torch.device("cuda", local)
    off_np = (np.arange(0, 2 * B + 1, dtype=np.int64) * RL)
    off_d = torch.from_numpy(off_np).to(dev)
    batches = []
    host_first = None
    for s in range(W + K):
        seq, off, tt, tp = tx.reads(B, read_len=RL, seed=2, first_pair=(rank * (K + W) + s) * B, threads=min(thr, 64), truth=False)
        if s == W and rank == 0:
            host_first = seq[: 2 * RL * min(B, a.cpu_sample)].copy() if a.cpu_sample > 0 else None
        batches.append(torch.from_numpy(seq).to(dev))
    torch.cuda.synchronize()
    rbs = [api.make_read_batch(int(b.data_ptr()), int(off_d.data_ptr()), B, paired=True, on_device=True) for b in batches]
    # ---- warmup (sizes every work buffer; model/eq state is reset afterwards) ----
    for s in range(W):
        ctx.map_batch(rbs[s], fetch=False); ctx.eq_accumulate()
    if a.lanes > 1:
        for s in range(min(W, a.lanes)):   # size the work buffers of every lane
            ctx.map_submit(rbs[0])
        for s in range(min(W, a.lanes)):
            ctx.map_wait()
    if W:
        e = ctx.eq_finish()
        api.em_steps(e, idx.ref_lens().astype(np.float64), np.full(idx.num_refs, 100.0), 2, api.em_opts(), device=local)
    ctx.reset()
    ctx.set_profiling(not a.no_profile)
    ctx.stage_times(reset=True)
    eff_ref = idx.ref_lens().astype(np.float64)
    # ---- timed region: exactly K steps + the job's inference tail ----
    if dist: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot = None
    depth = max(1, a.lanes)          # batches in flight on the mapping lanes (sq_map_submit / sq_map_wait)
    if depth > 1:
        for s in range(W, min(W + K, W + depth)):
            ctx.map_submit(rbs[s])
    for s in range(W, W + K):
        if depth > 1:
            _, _, _, st = ctx.map_wait()
        else:
            _, _, _, st = ctx.map_batch(rbs[s], fetch=False)
        ctx.eq_accumulate()
        if depth > 1 and s + depth < W + K:
            ctx.map_submit(rbs[s + depth])
        tot = st if tot is None else {k: tot[k] + v for k, v in st.items()}
    t_map = time.perf_counter() - t0
    eq = ctx.eq_finish()
    t_eqf = time.perf_counter() - t0 - t_map
    lm, uq, tc, le = ctx.model()
    if dist:  # one RCCL all-gather per packed field; every rank merges the others' tables exactly (integer sums)
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
    t_a = time.perf_counter()
    proj = api.normalize_alphas(eq, lm, uq, tc)
    t_norm = time.perf_counter() - t_a
    eff = np.exp(le)
    t_a = time.perf_counter()
    alphas, rep = ctx.em_optimize(eff, proj, api.em_opts())   # the ctx's own classes (incl. merged ones), read from the export resident in HBM
    t_em = time.perf_counter() - t_a
    torch.cuda.synchronize()
    if dist: dist.barrier()
    t1 = time.perf_counter()
    dt = t1 - t0
    if dist:
        tt_ = torch.tensor([dt], device=(torch.device("cpu") if a.debug_one_device else dev), dtype=torch.float64)
        dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
        dt = float(tt_.item())
    stages = ctx.stage_times()
    # EM iteration rate from a fixed-count run on the final table (outside the timed region)
    _, rep_it = api.em_steps(eq, eff, np.maximum(alphas, 1e-3), 200, api.em_opts(), device=local)
    if rank != 0:
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return
    E = len(eq.count); Lb = len(eq.tid); M = idx.num_refs
    em_bytes = 36 * Lb + 16 * E + 64 * M
    em_gbs = em_bytes / (rep_it["ms_per_iter"] * 1e-3) / 1e9
    sb = stage_bytes(tot, K * B, RL)
    stage_rows = {k: {"ms_total": round(v[0], 3), "launches": v[1], "avg_ms": round(v[0] / max(1, v[1]), 4),
        "alg_GBps": round(sb.get(k, 0) / max(v[0], 1e-9) / 1e6, 1)} for k, v in stages.items() if v[1]}
    # roofline: the dominant SINGLE kernel (stages that aggregate many launches — the library sort, the scans, the
    # eq stage's mini-batch chain that overlaps mapping on its own stream — are not kernels and are excluded)
    single = {"k_pack": "k_pack", "k_seed": "k_seed", "k_mems": "k_mems", "k_join_fill": "k_join2", "k_score": "k_score", "k_dp": "k_dp",
        "k_select": "k_select"}
    cand = [k for k in stage_rows if k in single]
    dom = max(cand, key=lambda k: stage_rows[k]["ms_total"]) if cand else None
    roof = None
    if dom:
        per_launch = sb[dom] / max(1, stage_rows[dom]["launches"])
        ach = per_launch / (stage_rows[dom]["avg_ms"] * 1e-3) / 1e9
        traffic = None; tnote = "no PMC profile committed for this kernel"
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")))
            kk = pm["kernels"].get(single[dom])
            if kk and kk.get("fetch_bytes_per_launch") is not None:
                traffic = int(kk["fetch_bytes_per_launch"] + (kk.get("write_bytes_per_launch") or 0))
                tnote = "FETCH_SIZE + WRITE_SIZE per launch from profiles/r02_pmc_traffic.json (rocprofv3 --pmc, separate passes, same workload at --steps 2); PMC cannot be sampled inside the timed run"
        except Exception:
            pass
        roof = {"kernel": single[dom], "bound": "hbm", "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 5),
            "traffic": traffic,
                "avg_launch_ms": stage_rows[dom]["avg_ms"], "alg_bytes_per_launch": int(per_launch), "traffic_note": tnote,
                "alg_bytes_note": "per-kernel byte model = bench.py::stage_bytes (DESIGN.md section 5); k_seed: 64 B filter word per probe + 4 dependent 64 B sectors (pilot, slot record, string-pool word, unitig bounds) per hit + 64 B per uni-MEM (extension words, contig-table bounds, record) + the packed read"}
    cpu = None; parity = None
    if a.cpu_sample > 0 and world == 1 and host_first is not None:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc
        S = min(B, a.cpu_sample)
        oidx = orc.OrcIndex(idx)
        soff = (np.arange(0, 2 * S + 1, dtype=np.uint64) * np.uint64(RL))
        rb = api.make_read_batch(host_first, soff, S, paired=True)
        c0 = time.perf_counter()
        ro, aln, mt, stc = orc.map_batch(oidx, opts, rb, threads=ncores)
        c1 = time.perf_counter()
        ost = orc.OrcState(oidx, opts); ost.eq_accumulate(ro, aln, stc["num_with_joint_hits"]); ost.finish(); eqc = ost.eq_finish()
        lmc, uqc, tcc, lec, _ = ost.model()
        pc = orc.normalize_alphas(M, eqc, lmc, uqc, tcc)
        c2 = time.perf_counter()
        _, repc = orc.em_optimize(eqc, np.exp(lec), pc, api.em_opts())
        c3 = time.perf_counter()
        # the checker's EM loop is single-threaded (order-defined sums); its multi-threaded iteration (same arithmetic,
        # transcripts split over threads) is timed separately and used for the composite so the CPU gets its cores
        em_single_s = c3 - c2
        em_try = {t: orc.em_time_iters(eqc, np.exp(lec), 20, t) / 20.0 * repc["iters"] for t in sorted({min(ncores, 8), min(ncores, 32), ncores})}
        em_thr_n, em_thr_s = min(em_try.items(), key=lambda kv: kv[1])
        if em_single_s < em_thr_s: em_thr_n, em_thr_s = 1, em_single_s       # the CPU side gets its best configuration
        em_cpu_s = orc.em_time_iters(eq, eff, 20, ncores) / 20.0
        t_cpu = (c1 - c0) + (c2 - c1) + em_thr_s
        # parity at bench scale (outside the timed region): the same S pairs through the HIP path on a reset context, compared with
        # what the checker just produced — alignment records, per-read offsets, mapping types, counters, the class table (labels,
        # bins, counts, fixed-point weight sums), the online model, projected counts and the VBEM result
        import hashlib
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        ctx.set_profiling(False); ctx.reset()
        ro_g, aln_g, mt_g, st_g = ctx.map_batch(rb)
        ctx.eq_accumulate(); eq_g = ctx.eq_finish(); lm_g, uq_g, tc_g, le_g = ctx.model()
        p_g = api.normalize_alphas(eq_g, lm_g, uq_g, tc_g)
        a_g, rep_g = ctx.em_optimize(np.exp(le_g), p_g, api.em_opts())
        a_c, _ = orc.em_optimize(eqc, np.exp(lec), pc, api.em_opts())
        checks = {"alignments": sha(aln_g) == sha(aln), "read_offsets": sha(ro_g) == sha(ro), "map_types": sha(mt_g) == sha(mt), "counters": st_g == stc,
            "eq_classes": all(sha(getattr(eq_g, f)) == sha(getattr(eqc, f)) for f in ("off", "tid", "bins", "count", "wq", "h1", "h2")),
            "online_model": sha(lm_g) == sha(lmc) and sha(uq_g) == sha(uqc) and sha(tc_g) == sha(tcc) and sha(le_g) == sha(lec),
            "projected_counts": sha(p_g) == sha(pc), "vbem": rep_g["iters"] == repc["iters"] and sha(a_g) == sha(a_c)}
        parity = {"pairs": S, "transcripts": int(M), "alignments": int(len(aln)), "eq_classes": int(len(eqc.count)), "equal": all(checks.values()),
            "checks": checks, "alignments_sha256": sha(aln_g), "what": "first %d pairs of timed step 0: HIP path vs CPU checker, sha256 of every output array" % S}
        cpu = {"value": round(S / t_cpu / 1e6, 4), "unit": "M read-pairs/s", "cores": ncores, "kind": "port",
               "sample": "%d of the %d pairs of step 0 through the CPU checker (oracle/): map %.2fs (%d threads) + online model / eq-classes %.2fs (1 thread: the mini-batch chain is sequential) + VBEM %d iters %.2fs (best of 1/8/32/%d threads: %d)" % (S,
                   B, c1 - c0, ncores, c2 - c1, repc["iters"], em_thr_s, ncores, em_thr_n),
               "map_only_M_pairs_per_s": round(S / (c1 - c0) / 1e6, 4), "em_iters_per_s_full_table_%dthr" % ncores: round(1.0 / em_cpu_s, 2),
               # what the sample's rates would mean for the whole timed job (a model, not a measurement): per-pair costs scale with the pairs,
               # the EM runs once over the full table for as many iterations as the GPU job needed
               "extrapolated_full_job_M_pairs_per_s": round(K * B / ((K * B) * ((c1 - c0) + (c2 - c1)) / S + rep["iters"] * em_cpu_s) / 1e6, 4)}
    fq = None
    if a.fastq_pairs > 0 and world == 1:
        fq = run_from_fastq(ctx, idx, tx, a.fastq_pairs, RL, min(B, 1000000), min(thr, 64), api, capi)
    out = {
        "metric": baseline_metric(), "value": round(world * K * B / dt / 1e6, 4), "unit": "M read-pairs/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64/i32 (2-bit k-mers, integer scores) + f64 (log-space model, EM)", "data": "synthetic",
        "config": {"workload": "configs[1]: T200k synthetic human-shaped txome index (k=31, m=20), %d x %d = %d synthetic 2x%dbp pairs per GPU, -l IU defaults, VBEM" % (K,
            B, K * B, RL),
                   "transcripts": int(M), "txome_nt": int(tx.total_nt()), "distinct_kmers": int(idx.num_kmers), "unitigs": int(idx.num_unitigs), "index_hbm_bytes": int(idx.device_bytes),
                   "pairs_per_step": B, "parallelism": "reads sharded over %d GPU(s); eq-class tables all-gathered + merged exactly (%s); EM replicated" % (world,
                       "sq_dist_* over RCCL" if sqd is not None else ("torch.distributed" if dist is not None else "single rank"))},
        "breakdown": {"map_eq_s": round(t_map, 4), "tail_s(eq_export+normalize+EM)": round(dt - t_map, 4), "eq_finish_s": round(t_eqf, 4),
            "normalize_alphas_s": round(t_norm, 4), "em_call_s": round(t_em, 4), "em_iters": rep["iters"], "em_converged": rep["converged"],
            "em_device_ms": round(rep["device_ms"], 2),
                      "index_build_s": round(t_index, 1), "mapped_frac": round(tot["num_mapped"] / tot["num_reads"],
                          4), "hits_per_frag": round(tot["num_alignments"] / max(1, tot["num_mapped"]), 3),
                      "eq_classes": E, "label_entries": Lb, "stats": tot},
        "em": {"iters_per_s": round(1e3 / rep_it["ms_per_iter"], 1), "ms_per_iter": round(rep_it["ms_per_iter"], 4), "alg_bytes_per_iter": em_bytes,
            "alg_GBps": round(em_gbs, 1), "frac_of_8TBps": round(em_gbs / 8000.0, 4)},
        "stages": stage_rows, "roofline": roof, "cpu_baseline": cpu, "parity_check": parity, "from_fastq": fq,
    }
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
