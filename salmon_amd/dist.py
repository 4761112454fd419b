"""Multi-GPU plumbing for the eq-class reduction (one process per GPU, torch.distributed).

The mapping path has no collective: reads shard by rank.  After mapping, every rank contributes its
eq-class table; `all_gather_tables` moves the packed arrays with ONE padded all_gather per field
(RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests) and `reduce_model` combines the
per-transcript online state.  Merging is exact (integer counts, fixed-point weight sums), so every
rank ends with bit-identical tables whatever the gather order.
"""
import numpy as np

FIELDS = ["off", "tid", "wq", "count", "bins", "h1", "h2"]


def _gather_np(x, dist, device, world):
    import torch
    x = np.ascontiguousarray(x)
    view = x.view(np.int64) if x.dtype.itemsize == 8 else x.view(np.int32)
    t = torch.from_numpy(view.copy()).to(device)
    n = torch.tensor([t.numel()], device=device, dtype=torch.int64)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    mx = max(1, int(max(int(v) for v in ns)))
    pad = torch.zeros(mx, device=device, dtype=t.dtype)
    pad[: t.numel()] = t
    outs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return [o[: int(k)].cpu().numpy().view(x.dtype) for o, k in zip(outs, ns)]


def all_gather_tables(eq, dist, device):
    """Returns the list (indexed by rank) of every rank's EqClasses (w is left empty: merging uses wq)."""
    from . import api
    world = dist.get_world_size()
    parts = {f: _gather_np(getattr(eq, f), dist, device, world) for f in FIELDS}
    return [api.EqClasses(parts["off"][r], parts["tid"][r], np.zeros(len(parts["tid"][r])), parts["count"][r], parts["wq"][r],
                          parts["bins"][r], parts["h1"][r], parts["h2"][r]) for r in range(world)]


def reduce_model(log_mass, uniq, total, log_eff_len, dist, device):
    """SPEC §MG over torch.distributed (the gloo harness path; the product path is sq_dist_reduce_model over RCCL): uniq / total exact int64
    all-reduce; masses all-gathered and combined by logAdd in rank order (sq_merge_log_masses); effective lengths: rank 0's."""
    import torch
    from . import capi
    world = dist.get_world_size()
    tq = torch.from_numpy(np.stack([uniq.astype(np.int64), total.astype(np.int64)])).to(device)
    dist.all_reduce(tq)
    uq, tc = tq[0].cpu().numpy().astype(np.uint64), tq[1].cpu().numpy().astype(np.uint64)
    mine = torch.from_numpy(np.ascontiguousarray(log_mass, np.float64).copy()).to(device)
    outs = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(outs, mine)
    allm = np.ascontiguousarray(np.stack([o.cpu().numpy() for o in outs]))
    lm = np.zeros(len(log_mass))
    capi.check(capi.lib().sq_merge_log_masses(len(lm), world, allm.ctypes.data, lm.ctypes.data), "sq_merge_log_masses")
    le = torch.from_numpy(np.ascontiguousarray(log_eff_len).copy()).to(device)
    dist.broadcast(le, 0)
    return lm, uq, tc, le.cpu().numpy()


class _DevArray:
    """Zero-copy view of a device allocation for torch (torch.as_tensor honours __cuda_array_interface__)."""
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def merge_all_device(ctx, dist, device):
    """The whole eq-class reduction without leaving HBM: every rank exposes its canonical-order export as torch views
    of the library's device buffers, ONE padded all_gather per field (RCCL over xGMI) lands the other ranks' tables in
    device tensors, and sq_eq_merge_device folds them in (integer counts / fixed-point sums: exact in any order)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    ex = ctx.eq_export_device()
    sizes = torch.tensor([ex["E"], ex["L"]], device=device, dtype=torch.int64)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    EL = [(int(s[0]), int(s[1])) for s in all_sizes]
    gathered = {}
    for f in FIELDS:
        ptr, n, dt = ex[f]
        is64 = np.dtype(dt).itemsize == 8
        tdt = torch.int64 if is64 else torch.int32
        mine = torch.as_tensor(_DevArray(ptr, n, "<i8" if is64 else "<i4"), device=device) if n else torch.zeros(0, device=device, dtype=tdt)
        mx = max(1, max((e + 1 if f == "off" else (e if f in ("count", "h1", "h2") else l)) for e, l in EL))
        pad = torch.zeros(mx, device=device, dtype=tdt); pad[: mine.numel()] = mine
        outs = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(outs, pad)
        gathered[f] = outs
    torch.cuda.synchronize(device)
    for r in range(world):
        if r == rank or EL[r][0] == 0:
            continue
        ctx.eq_merge_device(EL[r][0], EL[r][1], {f: gathered[f][r].data_ptr() for f in FIELDS})
    return EL
