"""N > 1 path on CPU: world_size 2 over gloo.  Each rank takes its site range, produces records for it and
rank 0 receives all records in position order.  There is no GPU here, so the per-site records come from the
oracle (the checker standing in for the device); what is under test is the partition and the gather."""
import os
import socket

import numpy as np
import pytest

from basevarc_amd.sharding import shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 100000, 100003):
        for w in (1, 2, 3, 4, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_call_sizes_give_every_rank_equal_calls_and_no_short_tail():
    """bench.py --gpus W: rank r owns shard_range(100000, r, W) and cuts it into calls with call_sizes(..., 4000, 8 if W > 1
    else 1): at least 8 calls per pass on every rank, sizes within one site of each other (round 2 had 3 x 4000 + 500 at
    W = 8: a quarter of the pass was pipeline fill and drain), never more than the 4000 the single-GPU run uses."""
    from basevarc_amd.sharding import call_sizes
    assert call_sizes(100000, 4000, 1) == [4000] * 25                     # the 1-GPU headline: unchanged
    for w in (2, 4, 8):
        for r in range(w):
            lo, hi = shard_range(100000, r, w)
            sizes = call_sizes(hi - lo, 4000, 8)
            assert sum(sizes) == hi - lo and len(sizes) >= 8
            assert max(sizes) - min(sizes) <= 1 and max(sizes) <= 4000
    assert call_sizes(12500, 4000, 8) == [1563] * 4 + [1562] * 4
    assert call_sizes(5, 4000, 8) == [1] * 5 and call_sizes(0, 4000, 8) == []
    with pytest.raises(ValueError):
        call_sizes(10, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_sites, n_samples, out_path):
    import torch.distributed as dist
    from basevarc_amd.lib import SITE_DTYPE
    from basevarc_amd.sharding import gather_records, shard_range as sr
    from oracle import orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sr(n_sites, rank, world)
    b, q, r = orc.synth_tile(5, lo, hi - lo, n_samples)
    exp, _ = orc.dense_batch(b, q, r, 0.001, use_hist=True, threads=1)
    rec = np.zeros(hi - lo, dtype=SITE_DTYPE)
    for i, e in enumerate(exp):
        rec[i]["called"] = e["called"]; rec[i]["n_alt"] = e["n_alt"]; rec[i]["depth"] = e["depth"]
        rec[i]["var_qual"] = e["var_qual"]; rec[i]["n_passes"] = lo + i          # carries the global site index
    allrec = gather_records(rec, dst=0)
    if rank == 0:
        np.save(out_path, allrec)
    else:
        assert allrec is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_gloo(tmp_path):
    import torch.multiprocessing as mp
    from oracle import orc
    n_sites, n_samples, world = 37, 3000, 2
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_sites, n_samples, out), nprocs=world, join=True)
    got = np.load(out)
    assert len(got) == n_sites
    assert got["n_passes"].tolist() == list(range(n_sites))            # position order, no gap, no duplicate
    b, q, r = orc.synth_tile(5, 0, n_sites, n_samples)
    exp, _ = orc.dense_batch(b, q, r, 0.001, use_hist=True, threads=1)
    assert got["called"].tolist() == [e["called"] for e in exp]
    assert [list(x) for x in got["depth"]] == [e["depth"] for e in exp]


def test_bench_shards_one_workload_when_strong_and_disjoint_ranges_when_weak():
    """bench.py, N > 1 (BASELINE configs[3]): the SAME total_sites are split by site -- rank r owns
    shard_range(total, r, W), the union is the workload, nothing overlaps -- and tiles never straddle a rank."""
    total, tile = 100_000, 4000
    for world in (2, 4, 8):
        owned = [shard_range(total, r, world) for r in range(world)]
        assert owned[0][0] == 0 and owned[-1][1] == total
        assert sum(hi - lo for lo, hi in owned) == total
        for lo, hi in owned:
            sizes = [min(tile, hi - lo - s) for s in range(0, hi - lo, tile)]
            assert sum(sizes) == hi - lo and all(0 < s <= tile for s in sizes)
