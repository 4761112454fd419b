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
    """uniq/total: exact int64 all-reduce; masses: summed in linear space; effective lengths: rank 0's FLD."""
    import torch
    tq = torch.from_numpy(np.stack([uniq.astype(np.int64), total.astype(np.int64)])).to(device)
    dist.all_reduce(tq)
    uq, tc = tq[0].cpu().numpy().astype(np.uint64), tq[1].cpu().numpy().astype(np.uint64)
    lin = torch.from_numpy(np.where(np.isinf(log_mass), 0.0, np.exp(np.where(np.isinf(log_mass), 0.0, log_mass)))).to(device)
    dist.all_reduce(lin)
    linc = lin.cpu().numpy()
    lm = np.where(linc > 0, np.log(np.maximum(linc, 1e-300)), np.inf)
    le = torch.from_numpy(np.ascontiguousarray(log_eff_len)).to(device)
    dist.broadcast(le, 0)
    return lm, uq, tc, le.cpu().numpy()
