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
