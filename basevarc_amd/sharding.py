"""Site sharding across the GPUs of a node: one process per GPU, no collective on the data path.

Sites are independent (the reference already splits positions over threads with no shared state,
src/BaseVarC.cpp:399-403), so rank r of W simply owns a contiguous site range and all N samples of each of
its sites; the per-site reduction never crosses devices.  The only communication is the gather of the
fixed-size result records to one rank, in position order -- the counterpart of the reference's merge of
per-thread sub-files (src/BaseVarC.cpp:274-295).  torch.distributed supplies it (nccl = RCCL on GPUs,
gloo in the CPU tests).
"""
import numpy as np


def shard_range(n_sites, rank, world):
    """Contiguous, balanced split: rank r gets sites [lo, hi)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(int(n_sites), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def call_sizes(n_sites, sites_per_call, min_calls=1):
    """Sites of a rank's range cut into library calls: as few calls of at most `sites_per_call` sites as cover it, but at
    least `min_calls` (a rank of a multi-GPU run pipelines its calls -- stage 2 of call i under stage 1 of call i + 1 --
    and the fill and drain of that pipeline cost about one call per pass, so a pass should be >= 8 calls), all of the
    same size to within one site: no short tail call.  12,500 sites at 4,000 per call and min_calls 8: 4 x 1563 + 4 x 1562."""
    n_sites, sites_per_call, min_calls = int(n_sites), int(sites_per_call), int(min_calls)
    if n_sites <= 0:
        return []
    if sites_per_call < 1 or min_calls < 1:
        raise ValueError("sites_per_call and min_calls must be >= 1")
    calls = max(min_calls, -(-n_sites // sites_per_call))
    calls = min(calls, n_sites)
    base, extra = divmod(n_sites, calls)
    return [base + (1 if i < extra else 0) for i in range(calls)]


def gather_records(records, dst=0, group=None):
    """records: numpy structured array of this rank's sites (any fixed-size dtype).
    Returns the concatenation over ranks in rank (= position) order on rank `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    raw = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    n = torch.tensor([raw.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes) if sizes else 0
    buf = torch.zeros(max(mx, 1), dtype=torch.uint8, device=dev)
    if raw.size:
        buf[:raw.size] = torch.from_numpy(raw.copy()).to(dev)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)          # records are ~100 B per site: an all_gather is fine
    if rank != dst:
        return None
    parts = [o[:s].cpu().numpy() for o, s in zip(out, sizes)]
    return np.concatenate(parts).view(records.dtype)
