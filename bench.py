#!/usr/bin/env python3
"""Benchmark of the basetype hot path on MI355X (BASELINE.json metric: sites/s at N = 1e6 samples).

Workload (BASELINE.json configs[2]): synthetic pileup, 1e5 sites x 1e6 samples = 200 GB of base/qual
bytes, generated ON the device by the library's counter-based generator and kept resident in HBM as
tiles of `--tile-sites` sites.  One step = one tile through the whole path
(bvc_lrt_dense: histogram kernel -> EM/LRT kernel -> result records in HBM); step i uses tile i mod T.
A tile (8 GB) is 30x the 256 MiB Infinity Cache, so every step streams from HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

Sites shard across GPUs with no collective (each rank owns its own site range; weak scaling: per-GPU
work is fixed).  torch.distributed is used only for the barriers and the max-over-ranks of the time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--samples", type=int, default=1_000_000, help="samples per site (N)")
    p.add_argument("--total-sites", type=int, default=100_000, help="sites resident in HBM per GPU")
    p.add_argument("--tile-sites", type=int, default=4000, help="sites per step")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--cpu-sites", type=int, default=-1, help="sites for the CPU baseline (-1: one per core, 0: skip)")
    p.add_argument("--no-verify", action="store_true", help="CPU leg: time the baseline only, skip the wider check")
    p.add_argument("--verify-all", action="store_true",
                   help="after the run, check EVERY resident site against the oracle's histogram form (about a minute)")
    p.add_argument("--no-overlap", action="store_true", help="run the two kernels of a step back to back on one stream")
    p.add_argument("--groups", type=int, default=0, help="population groups (BASELINE configs[4]: 5); 0 = overall call only")
    p.add_argument("--group-layout", choices=("interleaved", "ordered"), default="interleaved",
                   help="group of sample i: i %% k (SURVEY 8d) or contiguous runs of columns (takes the column-range kernel)")
    p.add_argument("--coverage", type=float, default=1.0, help="fraction of samples covered per site (sparse variant)")
    p.add_argument("--profile-every", type=int, default=4,
                   help="time the kernels of every K-th step of the timed region with HIP events (four timing events per "
                        "step cost 10-20 us of stream time; 1 = every step)")
    return p.parse_args()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    a = parse()
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    # One rank per GPU.  (Rehearsal on a 1-GPU box: BVC_BENCH_BACKEND=gloo lets ranks share device 0.)
    backend = os.environ.get("BVC_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # RCCL; used for barrier + max only
        else:
            dist.init_process_group(backend)

    from basevarc_amd import Context
    from basevarc_amd.lib import SITE_DTYPE, results_from_tensor

    n = a.samples
    stride = (n + 15) // 16 * 16
    min_af = min(0.001, 100.0 / n)                               # src/BaseVarC.cpp:541-543
    ctx = Context(dev_index, stream=torch.cuda.current_stream())
    ctx.set_overlap(not a.no_overlap)       # EM/LRT of step i runs under the histogram pass of step i+1

    # ---- resident dataset: as many tiles of the 1e5-site workload as fit (all 25 on a 288 GB MI355X)
    want_tiles = max(1, (a.total_sites + a.tile_sites - 1) // a.tile_sites)
    free_b, _ = torch.cuda.mem_get_info(dev)
    tile_bytes = 2 * a.tile_sites * stride
    fit = int((free_b // (world if backend != "nccl" else 1) - (6 << 30)) // tile_bytes)
    n_tiles = max(1, min(want_tiles, fit))
    site_base = rank * a.total_sites                             # each rank owns its own site range
    log(f"generating {n_tiles} tiles of {a.tile_sites} sites x {n} samples ({n_tiles * tile_bytes / 1e9:.1f} GB) on device")
    tiles = []
    for t in range(n_tiles):
        b = torch.empty((a.tile_sites, stride), dtype=torch.int8, device=dev)
        q = torch.empty((a.tile_sites, stride), dtype=torch.int8, device=dev)
        r = torch.empty(a.tile_sites, dtype=torch.int8, device=dev)
        ctx.synth_dense_device(a.seed, site_base + t * a.tile_sites, b[:, :n], q[:, :n], r,
                               cov_thr16=int(round(a.coverage * 65536)))
        tiles.append((b[:, :n], q[:, :n], r))
    results = [torch.empty(a.tile_sites * SITE_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(n_tiles)]
    torch.cuda.synchronize()
    log("dataset resident; warm-up")

    group_t = grp_results = None
    if a.groups > 0:
        from basevarc_amd.lib import GROUP_DTYPE
        if a.group_layout == "ordered":
            gnp = (np.arange(n) * a.groups // n).astype(np.uint8)    # the same k equal groups as contiguous column runs
        else:
            gnp = (np.arange(n) % a.groups).astype(np.uint8)         # group = sample % k
        group_t = torch.from_numpy(gnp).to(dev)
        grp_results = [torch.empty(a.tile_sites * a.groups * GROUP_DTYPE.itemsize, dtype=torch.uint8, device=dev)
                       for _ in range(n_tiles)]

    def step(i):
        b, q, r = tiles[i % n_tiles]
        if a.groups > 0:
            ctx.lrt_dense_groups_device(b, q, r, min_af, group_t, a.groups, results[i % n_tiles], grp_results[i % n_tiles])
        else:
            ctx.lrt_dense_device(b, q, r, min_af, results[i % n_tiles])

    def barrier():
        ctx.join()                          # every step's results are complete before the clock is read
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # empirical read ceiling: the same 8 GB tile through a plain 16 B/lane streaming kernel
    empirical_gbs = ctx.stream_read_gbs(tiles[0][0].as_strided((a.tile_sites, stride), (stride, 1)))
    for i in range(a.warmup):
        step(n_tiles - 1 - (i % n_tiles))
    barrier()
    log("timed region")
    ctx.set_profiling(True)
    ctx.profile(reset=True)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        if a.profile_every > 1:
            ctx.set_profiling(i % a.profile_every == 0)
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile(reset=True)
    ctx.set_profiling(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    log(f"timed region done: {dt * 1e3:.1f} ms for {a.steps} steps")
    sites_total = a.steps * a.tile_sites * world
    value = sites_total / dt
    hist_ms = prof["hist_ms"] / max(1, prof["hist_launches"])
    em_ms = prof["em_ms"] / max(1, prof["em_launches"])
    alg_bytes = 2.0 * a.tile_sites * n                           # 2 B per (site, sample), read once
    achieved = alg_bytes / (hist_ms * 1e-3) / 1e9 if hist_ms > 0 else 0.0

    out = {
        "metric": "sites/sec at N=1e6 samples; achieved HBM GB/s vs roofline",
        "value": value, "unit": "sites/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"synthetic pileup {a.total_sites} sites x {n} samples per GPU (BASELINE configs[2]), "
                        f"{'dense coverage' if a.coverage >= 1 else f'coverage {a.coverage:g}'}, Q10-40, 20% polymorphic"
                        f"{f', {a.groups} population groups ({a.group_layout})' if a.groups else ''}; step = tile of {a.tile_sites} sites",
            "n_samples": n, "sites_per_step": a.tile_sites, "resident_tiles": n_tiles,
            "resident_GB_per_gpu": round(n_tiles * tile_bytes / 1e9, 1), "min_af": min_af,
            "sharding": f"sites x{world}, no collective", "seed": a.seed, "overlap": not a.no_overlap,
        },
        "roofline": {
            "bound": "hbm", "kernel": ("hist_dense_ranges_kernel" if a.group_layout == "ordered" else "hist_dense_groups_kernel") if a.groups > 0 else "hist_dense_kernel",
            "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(a, n),
            "avg_launch_ms": hist_ms, "launches_timed": int(prof["hist_launches"]), "algorithmic_bytes_per_launch": alg_bytes,
            "empirical_stream_read_GBs": empirical_gbs, "frac_of_empirical": achieved / empirical_gbs if empirical_gbs else None,
        },
        "kernels_ms_per_step": {(("hist_dense_ranges_kernel" if a.group_layout == "ordered" else "hist_dense_groups_kernel") if a.groups > 0 else "hist_dense_kernel"): hist_ms,
                                ("sum_groups + lrt + lrt_groups kernels" if a.groups > 0 else "lrt_kernel"): em_ms},
    }

    if rank == 0:
        last = results_from_tensor(results[(a.steps - 1) % n_tiles])
        out["em_passes_per_site"] = float(last["n_passes"].mean())
        out["called_fraction"] = float(last["called"].mean())
    if rank == 0 and world == 1 and a.cpu_sites != 0:
        # CPU leg (the only place the oracle is used here): the reference path's CPU port timed on a bounded sample
        # of the same workload, and -- since its answers are at hand -- checked against the GPU records of those
        # sites and of a wider sample through the oracle's histogram form.
        step(0)
        ctx.synchronize()
        out["cpu_baseline"] = cpu_baseline(tiles[0], min_af, a, np, results_from_tensor(results[0]))
        if not a.no_verify:
            out["cpu_baseline"]["gpu_check_hist_form"] = spot_check(ctx, tiles, results, min_af, a, np)
            if a.groups > 0:
                out["cpu_baseline"]["gpu_check_groups"] = spot_check_groups(ctx, tiles, results, grp_results, group_t,
                                                                            min_af, a, np)
    if rank == 0 and world == 1 and a.verify_all:
        out["verify_all"] = verify_all(ctx, tiles, results, step, min_af, a, np)
        if a.groups > 0:
            out["verify_all"]["groups_tile0"] = verify_groups_tile(ctx, tiles, results, grp_results, group_t, step, min_af, a, np)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def pmc_traffic(a, n):
    """HBM bytes per hist-kernel launch from a committed rocprofv3 --pmc pass (profiles/pmc_traffic.json),
    corrected as MI355X_MICROARCH.md prescribes; null when no such pass matches this workload."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
        if d.get("n_samples") == n and d.get("sites_per_launch") == a.tile_sites and a.groups == 0 and a.coverage >= 1:
            return d.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def spot_check(ctx, tiles, results, min_af, a, np):
    """Part of the CPU leg, after the timed region: 64 sites of tile 0 against the oracle's histogram form."""
    from basevarc_amd.lib import results_from_tensor
    from oracle import orc
    i = 0                                                    # tile 0 was just recomputed by the CPU leg
    b, q, r = tiles[i]
    res = results_from_tensor(results[i])
    pick = np.linspace(0, a.tile_sites - 1, 64).astype(int)
    bad = 0
    for s in pick:
        cnt = orc.dense_hist(b[s].cpu().numpy(), q[s].cpu().numpy())
        e = orc.hist_lrt(cnt, int(r[s].item()), min_af)
        g = res[s]
        ok = (int(g["called"]) == e["called"] and [int(x) for x in g["depth"]] == e["depth"]
              and [int(g["alt_base"][k]) for k in range(g["n_alt"])] == e["alt_base"]
              and all(abs(float(g["af"][k]) - e["af"][k]) <= 1e-6 for k in range(e["n_alt"]))
              and abs(float(g["var_qual"]) - e["var_qual"]) <= 1e-6 * max(1.0, abs(e["var_qual"])))
        bad += not ok
    return {"sites_checked": int(len(pick)), "mismatches": int(bad)}


def verify_all(ctx, tiles, results, step, min_af, a, np):
    """every site of the resident dataset against the CPU histogram path (oracle, OpenMP over
    sites): tile by tile, device -> host, one C call per tile."""
    from basevarc_amd.lib import results_from_tensor
    from oracle import orc
    t0 = time.perf_counter()
    for i in range(len(tiles)):
        step(i)                                              # results[i] <- tile i, whatever the timed loop left there
    ctx.join()
    ctx.synchronize()
    bad = called = sites = 0
    worst_af = worst_q = 0.0
    for i, (b, q, r) in enumerate(tiles):
        exp, _ = orc.dense_batch(b.cpu().numpy(), q.cpu().numpy(), r.cpu().numpy(), min_af, use_hist=True, threads=16)
        res = results_from_tensor(results[i])
        for s, e in enumerate(exp):
            g = res[s]
            ok = (int(g["called"]) == e["called"] and [int(x) for x in g["depth"]] == e["depth"]
                  and [int(g["alt_base"][k]) for k in range(g["n_alt"])] == e["alt_base"])
            if ok:
                for k in range(e["n_alt"]):
                    d = abs(float(g["af"][k]) - e["af"][k])
                    worst_af = max(worst_af, d)
                    ok = ok and d <= 1e-6
                dq = abs(float(g["var_qual"]) - e["var_qual"]) / max(1.0, abs(e["var_qual"]))
                worst_q = max(worst_q, dq)
                ok = ok and dq <= 1e-6
            bad += not ok
            called += e["called"]
        sites += len(exp)
        log(f"verify-all: tile {i + 1}/{len(tiles)}, {sites} sites, {bad} mismatches")
    return {"sites_checked": int(sites), "called": int(called), "mismatches": int(bad), "max_abs_af_diff": worst_af,
            "max_rel_var_qual_diff": worst_q, "against": "oracle histogram form (CPU, 16 threads)",
            "seconds": time.perf_counter() - t0}


def verify_groups_tile(ctx, tiles, results, grp_results, group_t, step, min_af, a, np):
    """every site of tile 0 against the oracle's restatement of the caller's --group loop
    (histogram form): per-group depths, which groups ran, per-group AF."""
    from basevarc_amd.lib import GROUP_DTYPE
    from oracle import orc
    t0 = time.perf_counter()
    step(0)
    ctx.join()
    ctx.synchronize()
    b, q, r = tiles[0]
    hb, hq, hr = b.cpu().numpy(), q.cpu().numpy(), r.cpu().numpy()
    g = group_t.cpu().numpy()
    gres = grp_results[0].cpu().numpy().view(GROUP_DTYPE).reshape(a.tile_sites, a.groups)
    bad = ran_total = 0
    worst = 0.0
    for s in range(a.tile_sites):
        _, gd, ga, ran, pres = orc.dense_site_groups(hb[s], hq[s], int(hr[s]), min_af, g, a.groups, use_hist=True)
        d = float(np.max(np.abs(gres[s]["af"] - ga))) if ga.size else 0.0
        worst = max(worst, d)
        ok = (np.array_equal(gres[s]["depth"], gd) and np.array_equal(gres[s]["ran"], ran)
              and np.array_equal(gres[s]["present"], pres) and d <= 1e-6)
        bad += not ok
        ran_total += int(np.sum(ran))
        if (s + 1) % 1000 == 0:
            log(f"verify groups: {s + 1}/{a.tile_sites} sites, {bad} mismatches")
    return {"sites_checked": int(a.tile_sites), "group_runs": int(ran_total), "mismatches": int(bad),
            "max_abs_group_af_diff": worst, "seconds": time.perf_counter() - t0}


def spot_check_groups(ctx, tiles, results, grp_results, group_t, min_af, a, np):
    """8 sites of the last processed tile against the oracle's restatement of the caller's --group loop."""
    from basevarc_amd.lib import GROUP_DTYPE
    from oracle import orc
    i = 0
    b, q, r = tiles[i]
    g = group_t.cpu().numpy()
    gres = grp_results[i].cpu().numpy().view(GROUP_DTYPE).reshape(a.tile_sites, a.groups)
    bad = 0
    pick = np.linspace(0, a.tile_sites - 1, 8).astype(int)
    for s in pick:
        _, gd, ga, ran, _ = orc.dense_site_groups(b[s].cpu().numpy(), q[s].cpu().numpy(), int(r[s].item()), min_af, g,
                                                  a.groups, use_hist=True)
        ok = (np.array_equal(gres[s]["depth"], gd) and np.array_equal(gres[s]["ran"], ran)
              and np.allclose(gres[s]["af"], ga, rtol=0, atol=1e-6))
        bad += not ok
    return {"sites_checked": int(len(pick)), "mismatches": int(bad)}


def cpu_baseline(tile, min_af, a, np, gpu_records):
    """The faithful per-sample CPU port (oracle/basetype_oracle.c) on a bounded sample of the same
    workload: one site per host thread (about 10-20 s each at N = 1e6); its answers are compared with
    the GPU records of the same sites."""
    from oracle import orc
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)                                       # the GPU box's CPU share for one GPU
    k = a.cpu_sites if a.cpu_sites > 0 else cores
    log(f"cpu baseline: {k} sites on {min(cores, k)} threads (about 20 s per site per core)")
    b, q, r = tile
    hb, hq, hr = b[:k].cpu().numpy(), q[:k].cpu().numpy(), r[:k].cpu().numpy()
    t0 = time.perf_counter()
    exp, used = orc.dense_batch(hb, hq, hr, min_af, use_hist=False, threads=min(cores, k))
    dt = time.perf_counter() - t0
    bad = 0
    for s, e in enumerate(exp):
        g = gpu_records[s]
        floor = 1e-6 + 2e-10 * abs(e["lr_alt"])             # drift of the per-sample sum
        ok = (int(g["called"]) == e["called"] and [int(x) for x in g["depth"]] == e["depth"]
              and [int(g["alt_base"][i]) for i in range(g["n_alt"])] == e["alt_base"]
              and all(abs(float(g["af"][i]) - e["af"][i]) <= 1e-6 for i in range(e["n_alt"]))
              and abs(float(g["var_qual"]) - e["var_qual"]) <= max(floor, 1e-6 * abs(e["var_qual"])))
        bad += not ok
    return {"value": k / dt, "unit": "sites/s", "cores": int(used), "kind": "port",
            "sample": f"first {k} sites of tile 0 at N={a.samples}, one site per thread, faithful per-sample "
                      f"restatement of BaseType ctor+LRT+EM (oracle/basetype_oracle.c), {dt:.1f} s",
            "gpu_check_same_sites": {"sites": k, "mismatches": int(bad)}}


if __name__ == "__main__":
    main()
