#!/usr/bin/env python3
"""Benchmark of the basetype hot path on MI355X (BASELINE.json metric: sites/s at N = 1e6 samples).

Workload (BASELINE.json configs[2]): synthetic pileup, 1e5 sites x 1e6 samples = 200 GB of base/qual
bytes, generated ON the device by the library's counter-based generator and kept resident in HBM as
tiles of `--tile-sites` sites.  One STEP = one pass of the whole hot path over every resident tile of the
rank, i.e. over the whole 1e5-site workload at N = 1 (25 calls of bvc_lrt_dense: histogram kernel ->
EM/LRT kernel -> result records in HBM).  A tile (8 GB) is 30x the 256 MiB Infinity Cache, so every call
streams from HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

N > 1 (BASELINE.json configs[3]): the SAME 1e5 x 1e6 workload is split by site, rank r owns
shard_range(1e5, r, N) (basevarc_amd/sharding.py) -- strong scaling, no collective on the data path.
torch.distributed (nccl = RCCL) is used only for the barriers and the max-over-ranks of the time.
`--scaling weak` gives every rank its own 1e5 sites instead.

After the headline region rank 0 (N = 1 only) runs short legs over the other single-GPU configurations, each
reported as a sub-record of "legs" with its own roofline: configs[1] (1e4 x 1e4, EM-bound), configs[4]
(k = 5 groups, both sample orders) and the ragged (CSR) entry point at 10 % coverage; then the CPU baseline.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PCIE_PEAK_GBS = 63.0         # PCIe Gen5 x16, one direction: the bound of callers that hand over HOST pointers
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (about 6.3 TB/s is what a copy reaches)
N_SIMD = 1024                # 256 CUs x 4 SIMDs
ENGINE_CLOCK_HZ = 2.4e9      # MI355X peak engine clock
VALU_CYCLES_PER_WAVE_INST = 4     # wave64 FP64 / 32-bit VALU instruction (tools/micro/valu_rate.hip)
# Stage 2's FP64-VALU roofline needs two per-pass figures that only a counter pass can give (SQ_INSTS_VALU of the region kernel
# over sites x EM passes): what a pass NEEDS with every lane group busy and what the kernel EXECUTES at a given depth.  They live
# in profiles/stage2_valu.json next to the hash of the kernel source they were measured on (tools/pmc_stage2.py --json), the way
# pmc_traffic.json ties the HBM traffic to hist_kernel.hip: a later change of em_items.hip turns them "stale" in the bench line
# instead of silently keeping numbers of another kernel.
def stage2_valu():
    p = os.path.join(ROOT, "profiles", "stage2_valu.json")
    try:
        d = json.load(open(p))
    except Exception:
        return None, "no profiles/stage2_valu.json"
    from basevarc_amd.build import code_sha16
    sha = code_sha16(os.path.join(ROOT, "basevarc_amd", "csrc", "em_items.hip"))
    if d.get("em_items_sha16") != sha:
        return None, f"stale: counters taken on em_items.hip {d.get('em_items_sha16')}, tree has {sha}"
    return d, {"file": "profiles/stage2_valu.json", "em_items_sha16": sha, "summary": d.get("summary")}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--samples", type=int, default=1_000_000, help="samples per site (N)")
    p.add_argument("--total-sites", type=int, default=100_000, help="sites of the workload (split over the ranks when strong)")
    p.add_argument("--tile-sites", type=int, default=4000, help="sites per library call")
    p.add_argument("--row-align", type=int, default=128, help="row stride of the tiles is a multiple of this many bytes")
    p.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--cpu-sites", type=int, default=-1, help="sites for the CPU baseline (-1: one per core, 0: skip)")
    p.add_argument("--no-verify", action="store_true", help="CPU leg: time the baseline only, skip the wider check")
    p.add_argument("--no-legs", action="store_true", help="skip the configs[1] / configs[4] / CSR legs")
    p.add_argument("--packed", action="store_true",
                   help="headline region on packed tiles (one byte per sample, bvc_lrt_dense_packed): an additive layout, "
                        "NOT the BASELINE metric's -- for profiling that kernel")
    p.add_argument("--csr-sites", type=int, default=0, help="sites per call of the ragged (CSR) leg (0 = --tile-sites)")
    p.add_argument("--verify-all", action="store_true",
                   help="after the run, check EVERY resident site against the oracle's histogram form (about a minute)")
    p.add_argument("--no-overlap", action="store_true", help="run the two kernels of a call back to back on one stream")
    p.add_argument("--groups", type=int, default=0, help="headline region in group mode (BASELINE configs[4]: 5); 0 = overall call only")
    p.add_argument("--group-layout", choices=("interleaved", "ordered"), default="interleaved",
                   help="group of sample i: i %% k (SURVEY 8d) or contiguous runs of columns (takes the column-range kernel)")
    p.add_argument("--coverage", type=float, default=1.0, help="fraction of samples covered per site (dense tiles with sentinels)")
    p.add_argument("--profile-every", type=int, default=4,
                   help="time the kernels of every K-th call of the timed region with HIP events (four timing events per "
                        "call cost 10-20 us of stream time; 1 = every call)")
    a = p.parse_args()
    if a.packed:                            # profiling mode: no CPU leg (the checks compare two-byte tiles), no groups
        a.cpu_sites, a.no_verify, a.verify_all = 0, True, False
    return a


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def group_labels(np, n, k, layout):
    if layout == "ordered":
        return (np.arange(n) * k // n).astype(np.uint8)          # the same k equal groups as contiguous column runs
    return (np.arange(n) % k).astype(np.uint8)                   # group = sample % k


def hist_kernel_name(groups, layout):
    if groups <= 0:
        return "hist_dense_kernel"
    if layout == "ordered":
        return "hist_dense_ranges_kernel"
    # any order: from 4 groups on the kernel that packs the rows in registers (256 slots x 16 / 8 / 4 copies per histogram)
    return "hist_dense_groups_slots_kernel" if groups >= 4 else "hist_dense_groups_kernel"


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
    shared_device = backend != "nccl"        # gloo rehearsal: the ranks share the devices present
    dev_index = local_rank if not shared_device else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev)  # RCCL; used for barrier + max only
            except Exception as e:                              # the data path needs no collective: time with gloo instead
                print(f"[bench] rank {rank}: nccl rendezvous failed ({e}); using gloo for the barrier", file=sys.stderr, flush=True)
                backend = "gloo-after-nccl-failure"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group(backend)

    from basevarc_amd import Context
    from basevarc_amd.lib import SITE_DTYPE, results_from_tensor
    from basevarc_amd.sharding import call_sizes, shard_range

    n = a.samples
    stride = (n + a.row_align - 1) // a.row_align * a.row_align
    min_af = min(0.001, 100.0 / n)                               # src/BaseVarC.cpp:541-543
    ctx = Context(dev_index, stream=torch.cuda.current_stream())
    ctx.set_overlap(not a.no_overlap)       # EM/LRT of call i runs under the histogram pass of call i+1

    # ---- this rank's part of the workload, resident in HBM (all 25 tiles of 1e5 sites on a 288 GB MI355X at N = 1)
    if a.scaling == "strong":
        lo, hi = shard_range(a.total_sites, rank, world)         # configs[3]: one workload, split by site
    else:
        lo, hi = rank * a.total_sites, (rank + 1) * a.total_sites
    my_sites = hi - lo
    free_b, _ = torch.cuda.mem_get_info(dev)
    budget = free_b // (world if shared_device else 1) - (16 << 30)         # room for the legs and scratch
    fit_sites = max(a.tile_sites, int(budget // (2 * stride)) // a.tile_sites * a.tile_sites)
    res_sites = min(my_sites, fit_sites)                         # sites actually resident (= my_sites on MI355X)
    # calls of a pass: --tile-sites each on one GPU; with several GPUs at least 8 equal calls per rank (the two-stage
    # pipeline's fill and drain cost about one call per pass: 12,500 sites are 8 x 1,563, not 3 x 4,000 + 500)
    tile_sizes = call_sizes(res_sites, a.tile_sites, 8 if world > 1 else 1)
    n_tiles = len(tile_sizes)
    log(f"rank 0: sites [{lo}, {lo + res_sites}) as {n_tiles} tiles x {n} samples ({(1 if a.packed else 2) * res_sites * stride / 1e9:.1f} GB) on device")
    tiles = []
    s0 = lo
    for ts in tile_sizes:
        b = torch.empty((ts, stride), dtype=torch.int8, device=dev)
        q = torch.empty((ts, stride), dtype=torch.int8, device=dev)
        r = torch.empty(ts, dtype=torch.int8, device=dev)
        ctx.synth_dense_device(a.seed, s0, b[:, :n], q[:, :n], r, cov_thr16=int(round(a.coverage * 65536)))
        if a.packed:                        # keep only the packed form of the tile
            pt, bad = ctx.pack_dense_device(b[:, :n], q[:, :n])
            assert bad == 0
            tiles.append((pt, None, r))
            del b, q
        else:
            tiles.append((b[:, :n], q[:, :n], r))
        s0 += ts
    results = [torch.empty(ts * SITE_DTYPE.itemsize, dtype=torch.uint8, device=dev) for ts in tile_sizes]
    torch.cuda.synchronize()
    log("dataset resident; warm-up")

    group_t = grp_results = None
    if a.groups > 0:
        from basevarc_amd.lib import GROUP_DTYPE
        group_t = torch.from_numpy(group_labels(np, n, a.groups, a.group_layout)).to(dev)
        grp_results = [torch.empty(ts * a.groups * GROUP_DTYPE.itemsize, dtype=torch.uint8, device=dev) for ts in tile_sizes]

    def call(i):
        b, q, r = tiles[i]
        if a.packed and a.groups > 0:
            ctx.lrt_dense_groups_packed_device(b, r, min_af, group_t, a.groups, results[i], grp_results[i])
        elif a.packed:
            ctx.lrt_dense_packed_device(b, r, min_af, results[i])
        elif a.groups > 0:
            ctx.lrt_dense_groups_device(b, q, r, min_af, group_t, a.groups, results[i], grp_results[i])
        else:
            ctx.lrt_dense_device(b, q, r, min_af, results[i])

    calls = [0]

    def step():                             # one pass over every resident tile = the rank's whole share of the workload
        for i in range(n_tiles):
            if a.profile_every > 1:
                ctx.set_profiling(calls[0] % a.profile_every == 0)
            calls[0] += 1
            call(i)

    def barrier():
        ctx.join()                          # every call's results are complete before the clock is read
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # empirical read ceiling: whole tiles (4 GB of base bytes each) through a plain 16 B/lane streaming kernel, separately timed
    # passes of three sweeps each over three different tiles, before and after the warm-up (round 3 took ONE measurement, before
    # the warm-up, and the driver's run put it below the kernel it caps).  The plain read is bimodal from process to process and
    # tile to tile on this part -- 7.0-7.15 TB/s mostly, 6.75-6.8 now and then -- while the histogram kernel
    # holds 6.74-6.80; a ceiling is the best the part does on this shape at any time: the maximum of all passes.  The median of
    # the passes after the warm-up rides along.
    def whole(i):
        t = tiles[i][0]
        return t.as_strided((tile_sizes[i], t.stride(0)), (t.stride(0), 1))
    probe = sorted({0, n_tiles // 2, n_tiles - 1})
    reads_before = [ctx.stream_read_gbs(whole(i), repeats=3) for i in probe for _ in range(2)]
    for _ in range(a.warmup):
        step()
    barrier()
    reads = sorted(ctx.stream_read_gbs(whole(i), repeats=3) for i in probe for _ in range(3))
    empirical_gbs, empirical_median = max(reads[-1], max(reads_before)), reads[len(reads) // 2]
    barrier()
    log("timed region")
    ctx.set_profiling(a.profile_every <= 1)
    ctx.profile(reset=True)
    calls[0] = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    dt_own = dt
    prof = ctx.profile(reset=True)
    ctx.set_profiling(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        st = torch.tensor([res_sites], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(st, op=dist.ReduceOp.SUM)
        sites_per_step_all = int(st.item())
    else:
        sites_per_step_all = res_sites
    # every rank's own clock and kernel times, so that a slow rank can be told from a slow path
    my_hist_ms_site = prof["hist_ms"] / max(1, prof["sites"])
    mine = [float(rank), dt_own / a.steps * 1e3, float(res_sites), float(n_tiles), my_hist_ms_site, prof["em_ms"] / max(1, prof["sites"])]
    if world > 1:
        pr = torch.tensor(mine, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        allpr = [torch.zeros_like(pr) for _ in range(world)]
        dist.all_gather(allpr, pr)
        per_rank_raw = [[float(x) for x in t.cpu()] for t in allpr]
    else:
        per_rank_raw = [mine]

    log(f"timed region done: {dt * 1e3:.1f} ms for {a.steps} steps")
    value = a.steps * sites_per_step_all / dt
    # For the record, outside the timed region and on one GPU only: the same steps with every subset of every level run
    # (em_prune = 0, the E+M passes the reference runs).  The headline is bound by the histogram pass either way.
    value_all_run = None
    if world == 1:
        ctx.join(); ctx.set_tuning("em_prune", 0)
        extra = max(2, a.steps // 4)
        step(); ctx.join(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(extra):
            step()
        ctx.join(); torch.cuda.synchronize()
        value_all_run = extra * sites_per_step_all / (time.perf_counter() - t1)
        ctx.join(); ctx.set_tuning("em_prune", 1)
        step(); ctx.join(); torch.cuda.synchronize()              # (the records the checks below read are the default engine's)
        ctx.set_profiling(False); ctx.profile(reset=True)
    # the timed launches are full tiles except possibly the last of a pass: normalise per site
    hist_ms_per_site = prof["hist_ms"] / max(1, prof["sites"])
    em_ms_per_site = prof["em_ms"] / max(1, prof["sites"])
    call_sites = max(tile_sizes)
    hist_ms = hist_ms_per_site * call_sites
    em_ms = em_ms_per_site * call_sites
    alg_bytes = (1.0 if a.packed else 2.0) * call_sites * n   # 2 B per (site, sample), read once (packed: 1 B)
    achieved = alg_bytes / (hist_ms * 1e-3) / 1e9 if hist_ms > 0 else 0.0
    kname = hist_kernel_name(a.groups, a.group_layout).replace("hist_dense_kernel", "hist_packed_kernel").replace("hist_dense_", "hist_packed_") if a.packed else hist_kernel_name(a.groups, a.group_layout)
    traffic, traffic_source = pmc_traffic(a, n, kname)

    out = {
        "metric": "sites/sec at N=1e6 samples; achieved HBM GB/s vs roofline",
        "value": value, "unit": "sites/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
        "scaling": a.scaling if world > 1 else "none", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "value_with_every_subset_run": value_all_run,
        "config": {
            "workload": f"synthetic pileup {a.total_sites} sites x {n} samples (BASELINE configs[2]"
                        f"{'; configs[3]: split by site over the ranks' if world > 1 and a.scaling == 'strong' else ''}), "
                        f"{'dense coverage' if a.coverage >= 1 else f'coverage {a.coverage:g}'}, Q10-40, 20% polymorphic"
                        f"{f', {a.groups} population groups ({a.group_layout})' if a.groups else ''}"
                        f"{', PACKED tiles (1 byte per sample: additive layout, not the BASELINE metric)' if a.packed else ''}; "
                        f"step = one pass over the rank's resident sites in calls of {max(tile_sizes)}",
            "n_samples": n, "sites_per_step": sites_per_step_all, "sites_per_call": max(tile_sizes),
            "calls_per_step_per_gpu": n_tiles,
            "resident_GB_per_gpu": round((1 if a.packed else 2) * res_sites * stride / 1e9, 1),
            "row_stride": stride, "min_af": min_af,
            "sharding": f"sites x{world} ({a.scaling}), no collective; barrier over {backend}" if world > 1 else "1 GPU", "seed": a.seed,
            "overlap": not a.no_overlap,
        },
        "roofline": {
            "bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
            "avg_launch_ms": hist_ms, "launches_timed": int(prof["hist_launches"]), "algorithmic_bytes_per_launch": alg_bytes,
            "empirical_stream_read_GBs": empirical_gbs, "empirical_stream_read_median_GBs": empirical_median,
            "frac_of_empirical": achieved / empirical_gbs if empirical_gbs else None,
            "traffic_note": "HBM bytes per launch from the COMMITTED rocprofv3 counter passes named in traffic_source (FETCH_SIZE and "
                            "WRITE_SIZE in separate --pmc runs, corrected as MI355X_MICROARCH.md prescribes), valid for the kernel source "
                            "hashed there; not re-measured in this run.  achieved / avg_launch_ms ARE measured in this run (HIP events)",
        },
        "per_rank": [{"rank": int(p[0]), "ms_per_step": p[1], "sites": int(p[2]), "calls_per_step": int(p[3]),
                      "hist_ms_per_call": p[4] * max(tile_sizes), "stage2_ms_per_call": p[5] * max(tile_sizes),
                      "hist_frac": ((1.0 if a.packed else 2.0) * n / (p[4] * 1e-3) / 1e9 / HBM_PEAK_GBS) if p[4] > 0 else None}
                     for p in per_rank_raw],
        "kernels_ms_per_call": {kname: hist_ms,
                                ("sum_groups + lrt + lrt_groups kernels" if a.groups > 0 else "stage 2 (region_kernel)"): em_ms},
    }
    if res_sites < my_sites:
        out["config"]["note"] = f"only {res_sites} of this rank's {my_sites} sites fit in device memory; a step covers those"

    if rank == 0:
        last = results_from_tensor(results[0])
        out["em_passes_per_site"] = float(last["n_passes"].mean())
        out["called_fraction"] = float(last["called"].mean())
    if rank == 0 and world == 1 and not a.no_legs and a.groups == 0 and a.coverage >= 1 and not a.packed:
        out["legs"] = run_legs(ctx, a, tiles, tile_sizes, stride, min_af, np, torch, dev)
    if rank == 0 and world == 1 and a.cpu_sites != 0:
        # CPU leg (the only place the oracle is used here): the reference path's CPU port timed on a bounded sample
        # of the same workload, and -- since its answers are at hand -- checked against the GPU records of those
        # sites and of a wider sample through the oracle's histogram form.
        call(0)
        ctx.synchronize()
        out["cpu_baseline"] = cpu_baseline(tiles[0], min_af, a, np, results_from_tensor(results[0]))
        if not a.no_verify:
            out["cpu_baseline"]["gpu_check_hist_form"] = spot_check(tiles, results, min_af, a, np)
            if a.groups > 0:
                out["cpu_baseline"]["gpu_check_groups"] = spot_check_groups(tiles, grp_results, group_t, min_af, a, np)
    if rank == 0 and world == 1 and a.verify_all:
        out["verify_all"] = verify_all(ctx, tiles, results, call, min_af, a, np)
        if a.groups > 0:
            out["verify_all"]["groups_tile0"] = verify_groups_tile(ctx, tiles, grp_results, group_t, call, min_af, a, np)
    if rank == 0:
        rep = ctx.debug_report()
        if rep is not None:                 # -DBVC_CHECK_LDS build (tools/poison_run.sh): never a measurement
            out["diagnostic_build"] = {"lds_violations": rep, "note": "libbvc built with -DBVC_CHECK_LDS: timings are not the product's"}
        if "legs" in out:
            # LAST key of the line, so that a reader who keeps only the tail of stdout still sees every leg: {leg: [value, frac of
            # the leg's bounding roofline]} (sites/s; host_pointer_one_byte: the ragged one-byte call, frac of the PCIe peak)
            out["legs_summary"] = legs_summary(out["legs"])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


# ------------------------------------------------------------------------------------------------ legs
def _same_but_run_counts(rec, rec_all):
    """Records of the default engine against those with every subset run (em_prune = 0): everything the reference defines is the same
    bytes; only the two diagnostics that count what was RUN differ."""
    same = rec.copy()
    same["n_passes"] = rec_all["n_passes"]
    same["n_fits"] = rec_all["n_fits"]
    return same.tobytes() == rec_all.tobytes()


def timed_calls(ctx, fn, n_calls, warm=2, profile_every=4):
    """Runs fn(i) n_calls times, every profile_every-th launch timed by HIP events (timing events on every call would
    cost the two-stream overlap 10-20 us per call); returns (wall seconds, profile dict)."""
    import torch
    for i in range(warm):
        fn(i)
    ctx.join(); torch.cuda.synchronize()
    ctx.set_profiling(False)
    ctx.profile(reset=True)
    t0 = time.perf_counter()
    for i in range(n_calls):
        ctx.set_profiling(i % profile_every == 0)
        fn(i)
    ctx.join(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = ctx.profile(reset=True)
    ctx.set_profiling(False)
    return dt, prof


def run_legs(ctx, a, tiles, tile_sizes, stride, min_af, np, torch, dev):
    from basevarc_amd.lib import GROUP_DTYPE, SITE_DTYPE, results_from_tensor
    legs = {}
    n = a.samples
    full = [i for i, ts in enumerate(tile_sizes) if ts == a.tile_sites] or [0]

    # ---- BASELINE configs[4]: k = 5 population groups on the same tiles, both sample orders
    k = 5
    for layout in ("interleaved", "ordered"):
        g = torch.from_numpy(group_labels(np, n, k, layout)).to(dev)
        use = full[:6]
        res = [torch.empty(tile_sizes[i] * SITE_DTYPE.itemsize, dtype=torch.uint8, device=dev) for i in use]
        gres = [torch.empty(tile_sizes[i] * k * GROUP_DTYPE.itemsize, dtype=torch.uint8, device=dev) for i in use]

        def fn(j, use=use, res=res, gres=gres, g=g):
            i = j % len(use)
            b, q, r = tiles[use[i]]
            ctx.lrt_dense_groups_device(b, q, r, min_af, g, k, res[i], gres[i])
        n_calls = 8 * len(use)
        dt, prof = timed_calls(ctx, fn, n_calls)
        hist_ms = prof["hist_ms"] / max(1, prof["hist_launches"])
        alg = 2.0 * a.tile_sites * n
        kname = hist_kernel_name(k, layout)
        tr, src = pmc_traffic(a, n, kname)
        legs[f"config4_groups5_{layout}"] = {
            "workload": f"BASELINE configs[4]: {k} population groups, group of sample i = "
                        f"{'i % 5' if layout == 'interleaved' else 'i * 5 // N (contiguous runs)'}, N = {n}, "
                        f"{n_calls} calls of {a.tile_sites} sites",
            "value": n_calls * a.tile_sites / dt, "unit": "sites/s", "ms_per_call": dt / n_calls * 1e3,
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": alg / (hist_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": alg / (hist_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": hist_ms,
                         "launches_timed": int(prof["hist_launches"]), "algorithmic_bytes_per_launch": alg,
                         "traffic": tr, "traffic_source": src},
            "stage2_ms_per_call": prof["em_ms"] / max(1, prof["em_launches"]),
        }
        del res, gres

    # ---- additive: packed tiles, one byte per (site, sample) (include/bvc.h, bvc_lrt_dense_packed) -- the same
    # observations as the headline's tiles in half the bytes; records identical to the two-byte path
    use = full[:4]
    packed = []
    for i in use:
        b, q, r = tiles[i]
        pt, bad = ctx.pack_dense_device(b, q)
        assert bad == 0, "the synthetic qualities (10..40) fit the packed byte"
        packed.append((pt, r))
    res = [torch.empty(tile_sizes[i] * SITE_DTYPE.itemsize, dtype=torch.uint8, device=dev) for i in use]

    def fn_packed(j):
        i = j % len(use)
        ctx.lrt_dense_packed_device(packed[i][0], packed[i][1], min_af, res[i])
    n_calls = 25 * len(use)                                      # as many calls as a step of the headline region
    dt, prof = timed_calls(ctx, fn_packed, n_calls)
    two_byte = ctx.lrt_dense_device(tiles[use[0]][0], tiles[use[0]][1], tiles[use[0]][2], min_af)
    ctx.join(); torch.cuda.synchronize()
    same = bool(torch.equal(two_byte, res[0]))
    hist_ms = prof["hist_ms"] / max(1, prof["hist_launches"])
    alg = 1.0 * a.tile_sites * n                                 # ONE byte per (site, sample)
    tr, src = pmc_traffic(a, n, "hist_packed_kernel")
    legs["packed_1_byte_per_sample"] = {
        "workload": f"the headline's tiles packed to one byte per sample (base << 6 | qual), N = {n}, "
                    f"{n_calls} calls of {a.tile_sites} sites over {len(use)} tiles, overlap mode",
        "value": n_calls * a.tile_sites / dt, "unit": "sites/s", "ms_per_call": dt / n_calls * 1e3,
        "records_identical_to_two_byte_path": same,
        "roofline": {"bound": "hbm", "kernel": "hist_packed_kernel", "achieved": alg / (hist_ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (hist_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "avg_launch_ms": hist_ms, "launches_timed": int(prof["hist_launches"]),
                     "algorithmic_bytes_per_launch": alg, "bytes_per_sample": 1, "traffic": tr, "traffic_source": src},
        "stage2_ms_per_call": prof["em_ms"] / max(1, prof["em_launches"]),
    }
    # ... and the same in group mode (k = 5, both label orders): bvc_lrt_dense_groups_packed
    for layout in ("interleaved", "ordered"):
        g = torch.from_numpy(group_labels(np, n, k, layout)).to(dev)
        gres = [torch.empty(tile_sizes[i] * k * GROUP_DTYPE.itemsize, dtype=torch.uint8, device=dev) for i in use]

        def fn_pg(j, g=g, gres=gres):
            i = j % len(use)
            ctx.lrt_dense_groups_packed_device(packed[i][0], packed[i][1], min_af, g, k, res[i], gres[i])
        n_calls = 10 * len(use)
        dt, prof = timed_calls(ctx, fn_pg, n_calls)
        w_res, w_gres = ctx.lrt_dense_groups_device(tiles[use[0]][0], tiles[use[0]][1], tiles[use[0]][2], min_af, g, k)
        ctx.join(); torch.cuda.synchronize()
        same = bool(torch.equal(w_res, res[0]) and torch.equal(w_gres, gres[0]))
        hist_ms = prof["hist_ms"] / max(1, prof["hist_launches"])
        kname = "hist_packed_ranges_kernel" if layout == "ordered" else "hist_packed_groups_kernel"
        tr, src = pmc_traffic(a, n, kname)
        legs[f"packed_groups5_{layout}"] = {
            "workload": f"configs[4] (k = 5 groups, {layout}) on packed tiles, N = {n}, {n_calls} calls of {a.tile_sites} sites",
            "value": n_calls * a.tile_sites / dt, "unit": "sites/s", "ms_per_call": dt / n_calls * 1e3,
            "records_identical_to_two_byte_path": same,
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": alg / (hist_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": alg / (hist_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": hist_ms,
                         "launches_timed": int(prof["hist_launches"]), "algorithmic_bytes_per_launch": alg,
                         "bytes_per_sample": 1, "traffic": tr, "traffic_source": src},
            "stage2_ms_per_call": prof["em_ms"] / max(1, prof["em_launches"]),
        }
        del gres, w_res, w_gres
    del packed, res, two_byte

    # ---- ragged (CSR) entry point at 10 % coverage: what a real low-coverage pileup looks like at N = 1e6
    cov = 0.1
    # (sites per call: --csr-sites; the leg's rate is the same at 4000, 16,000 and 40,000)
    n_csr_tiles, csr_sites = 2, a.csr_sites or a.tile_sites
    csr = []
    slice_sites = 500                                            # generate + compact in slices: 1 GB of scratch
    tmp_b = torch.empty((slice_sites, stride), dtype=torch.int8, device=dev)
    tmp_q = torch.empty((slice_sites, stride), dtype=torch.int8, device=dev)
    for t in range(n_csr_tiles):
        r = torch.empty(csr_sites, dtype=torch.int8, device=dev)
        parts_b, parts_q, parts_s, counts = [], [], [], []
        for c0 in range(0, csr_sites, slice_sites):
            ns = min(slice_sites, csr_sites - c0)
            bb, qq = tmp_b[:ns, :n], tmp_q[:ns, :n]
            ctx.synth_dense_device(a.seed, 10_000_000 + t * csr_sites + c0, bb, qq, r[c0:c0 + ns],
                                   cov_thr16=int(round(cov * 65536)))
            ctx.synchronize()
            m = bb >= 0
            counts.append(m.sum(dim=1))
            parts_b.append(bb[m]); parts_q.append(qq[m])
            parts_s.append(torch.nonzero(m)[:, 1].to(torch.int32))          # which sample an observation is of (the ragged group call)
        cnt = torch.cat(counts).to(torch.int64)
        offs = torch.zeros(csr_sites + 1, dtype=torch.int64, device=dev)
        offs[1:] = torch.cumsum(cnt, 0)
        csr.append((offs, torch.cat(parts_b), torch.cat(parts_q), r, torch.cat(parts_s)))
        del parts_b, parts_q, parts_s, counts
    del tmp_b, tmp_q
    torch.cuda.synchronize()
    covered = float(sum(int(c[0][-1].item()) for c in csr)) / len(csr)
    res = [torch.empty(csr_sites * SITE_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in csr]

    def fn_csr(j):
        o, b, q, r, _ = csr[j % len(csr)]
        ctx.lrt_csr_device(o, b, q, r, min_af, res[j % len(csr)])
    n_calls = max(8, 160000 // csr_sites)
    dt, prof = timed_calls(ctx, fn_csr, n_calls)
    hist_ms = prof["hist_ms"] / max(1, prof["hist_launches"])
    em_ms = prof["em_ms"] / max(1, prof["em_launches"])
    # the same histogram pass with the chip to itself (the two stages of a call back to back on one stream, every call timed)
    ctx.join(); ctx.set_overlap(False)
    _, prof_alone = timed_calls(ctx, fn_csr, 8, warm=2, profile_every=1)
    ctx.set_overlap(not a.no_overlap)
    hist_alone_ms = prof_alone["hist_ms"] / max(1, prof_alone["hist_launches"])
    alg = 2.0 * covered                                          # 2 B per COVERED sample
    rec = results_from_tensor(res[0]).copy()
    # the same leg with every subset of every level run (em_prune = 0: the E+M passes the reference runs), for the record
    ctx.join(); ctx.set_tuning("em_prune", 0)
    dt_all, _ = timed_calls(ctx, fn_csr, max(8, n_calls // 2))
    rec_all = results_from_tensor(res[0]).copy()
    ctx.join(); ctx.set_tuning("em_prune", 1)
    fn_csr(0); ctx.join()
    csr_check = None
    if a.cpu_sites != 0 and not a.no_verify:                     # 16 sites of CSR tile 0 against the oracle's histogram form
        from oracle import orc
        o, b, q, r, _ = csr[0]
        oh = o.cpu().numpy()
        bad = 0
        pick = np.linspace(0, csr_sites - 1, 16).astype(int)
        for s_ in pick:
            bs, qs = b[oh[s_]:oh[s_ + 1]].cpu().numpy(), q[oh[s_]:oh[s_ + 1]].cpu().numpy()
            bad += not record_ok(rec[s_], orc.hist_lrt(orc.dense_hist(bs, qs), int(r[s_].item()), min_af))
        csr_check = {"sites_checked": int(len(pick)), "mismatches": int(bad)}
    legs["csr_coverage10pct"] = {
        "gpu_check_hist_form": csr_check,
        "workload": f"ragged (CSR) sites, bvc_lrt_csr: N = {n} samples at {cov:.0%} coverage = {covered / csr_sites:.0f} "
                    f"observations per site on average, {n_calls} calls of {csr_sites} sites over {len(csr)} tiles",
        "value": n_calls * csr_sites / dt, "unit": "sites/s", "ms_per_call": dt / n_calls * 1e3,
        "bound_by": "stage 2: region_kernel (FP64 VALU issue)",
        "value_with_every_subset_run": max(8, n_calls // 2) * csr_sites / dt_all,
        "em_passes_per_site_of_the_reference": float(rec_all["n_passes"].astype("int64").mean()),
        "records_identical_with_every_subset_run_except_the_run_counts": bool(_same_but_run_counts(rec, rec_all)),
        "roofline": em_roofline(rec, em_ms, dt / n_calls * 1e3, depth=100_000 if abs(covered / csr_sites - 1e5) < 2e4 else 0),
        "hist_roofline": {"bound": "hbm", "kernel": "hist_csr_block_kernel", "achieved": alg / (hist_ms * 1e-3) / 1e9,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (hist_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "avg_launch_ms": hist_ms, "launches_timed": int(prof["hist_launches"]),
                          "algorithmic_bytes_per_launch": alg,
                          "alone": {"avg_launch_ms": hist_alone_ms, "achieved": alg / (hist_alone_ms * 1e-3) / 1e9 if hist_alone_ms > 0 else None,
                                    "frac": alg / (hist_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if hist_alone_ms > 0 else None,
                                    "launches_timed": int(prof_alone["hist_launches"])},
                          "note": "frac / avg_launch_ms: HIP events around hist_wave_kernel + hist_csr_block_kernel underneath the two stage-2 "
                                  "launches they share every SIMD with (the block kernel itself: 0.20-0.24 ms there, kernel trace in "
                                  "profiles/); `alone`: the same pass with the chip to itself, measured in this run"},
    }
    # ---- host-pointer callers (BVC_PTR_HOST): bound by the host link, so they get the one-byte forms.  Never the
    # reported `value` (inputs are not resident); the roofline of this leg is PCIe, 63 GB/s.
    # ---- the --group loop on the same ragged sites (bvc_lrt_csr_groups, k = 5, labels interleaved, every 10th sample in no group):
    # what the host program calls per tile with --group since round 5 (src/BaseVarC.cpp:617-661 on the covered samples only)
    kg = 5
    lab = (np.arange(n) % kg).astype(np.uint8)
    lab[9::10] = 255
    g_t = torch.from_numpy(lab).to(dev)

    def fn_csrg(j):
        o, b, q, r, sm = csr[j % len(csr)]
        fn_csrg.out = ctx.lrt_csr_groups_device(o, b, q, sm, r, min_af, g_t, kg)
    ctx.join()
    n_calls_g = max(8, 80000 // csr_sites)
    dtg, profg = timed_calls(ctx, fn_csrg, n_calls_g)
    fn_csrg(0)                                                   # (the records compared below: tile 0's)
    ctx.join(); ctx.synchronize()
    hist_g_ms = profg["hist_ms"] / max(1, profg["hist_launches"])
    em_g_ms = profg["em_ms"] / max(1, profg["em_launches"])
    alg_g = 6.0 * covered                                        # base + quality + 4-byte sample index per COVERED sample
    same_overall = bool(_same_but_run_counts(results_from_tensor(fn_csrg.out[0]).copy(), rec))
    legs[f"csr_groups{kg}_coverage10pct"] = {
        "workload": f"ragged (CSR) sites with population groups, bvc_lrt_csr_groups: N = {n} samples at {cov:.0%} coverage, k = {kg} groups "
                    f"(labels interleaved, a tenth of the samples in none), {n_calls_g} calls of {csr_sites} sites",
        "value": n_calls_g * csr_sites / dtg, "unit": "sites/s", "ms_per_call": dtg / n_calls_g * 1e3,
        "overall_records_identical_to_bvc_lrt_csr": same_overall,
        "stage2_ms_per_call": em_g_ms, "bound_by": "stage 2 of the overall and the per-group calls (FP64 VALU issue)",
        "roofline": {"bound": "hbm", "kernel": "hist_csr_groups_kernel", "achieved": alg_g / (hist_g_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": alg_g / (hist_g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": hist_g_ms,
                     "launches_timed": int(profg["hist_launches"]), "algorithmic_bytes_per_launch": alg_g,
                     "note": "6 bytes per covered observation: base, quality and the 4-byte index of its sample; one workgroup per site, "
                             "LDS atomics into (k + 1) x 512 counters -- the leg is bound by stage 2, not by this kernel"},
    }
    del g_t
    hp = {}
    o, b, q, r, _ = csr[0]
    pk_h = ((b.to(torch.uint8) << 6) | q.to(torch.uint8)).cpu().numpy()
    o_h, r_h = o.cpu().numpy(), r.cpu().numpy()
    ctx.join(); ctx.synchronize()
    ctx.lrt_csr_packed(o_h, pk_h, r_h, min_af)                   # warm-up: staging buffers
    t0 = time.perf_counter()
    for _ in range(3):
        rec_h = ctx.lrt_csr_packed(o_h, pk_h, r_h, min_af)
    dth = (time.perf_counter() - t0) / 3
    moved = pk_h.nbytes + o_h.nbytes + r_h.nbytes + rec_h.nbytes
    hp["ragged_one_byte"] = {"sites_per_s": csr_sites / dth, "GBs": moved / dth / 1e9, "frac": moved / dth / 1e9 / PCIE_PEAK_GBS,
                             "ms_per_call": dth * 1e3, "bytes_per_call": int(moved),
                             "records_identical_to_device_pointer_call": bool(rec_h.tobytes() == results_from_tensor(res[0]).tobytes()),
                             "workload": f"bvc_lrt_csr_packed, BVC_PTR_HOST, {csr_sites} sites of {covered / csr_sites:.0f} observations"}
    del pk_h
    hs = min(1000, tile_sizes[0])
    tb, tq, tr_ = tiles[full[0]]
    pt_h = ctx.pack_dense_device(tb[:hs], tq[:hs])[0].cpu().numpy()
    rr_h = tr_[:hs].cpu().numpy()
    ctx.lrt_dense_packed(pt_h, rr_h, min_af)
    t0 = time.perf_counter()
    for _ in range(2):
        rec_h = ctx.lrt_dense_packed(pt_h, rr_h, min_af)
    dth = (time.perf_counter() - t0) / 2
    moved = pt_h.nbytes + rr_h.nbytes + rec_h.nbytes
    hp["dense_one_byte"] = {"sites_per_s": hs / dth, "GBs": moved / dth / 1e9, "frac": moved / dth / 1e9 / PCIE_PEAK_GBS,
                            "ms_per_call": dth * 1e3, "bytes_per_call": int(moved),
                            "workload": f"bvc_lrt_dense_packed, BVC_PTR_HOST, {hs} sites x {n} samples, pageable host memory, "
                                        "chunked staging (upload of chunk i + 1 under the kernels of chunk i)"}
    legs["host_pointer_one_byte"] = {"bound": "pcie", "peak": PCIE_PEAK_GBS, "unit": "GB/s", "value": hp["dense_one_byte"]["sites_per_s"],
                                     "value_unit": "sites/s (dense, N = 1e6)", **hp}
    del csr, res, pt_h

    # ---- BASELINE configs[1]: 1e4 sites x 1e4 samples, EM to convergence: the EM/LRT kernel is the bound
    ns1, n1 = 10_000, 10_000
    st1 = (n1 + a.row_align - 1) // a.row_align * a.row_align
    t1 = []
    for t in range(2):
        b = torch.empty((ns1, st1), dtype=torch.int8, device=dev)
        q = torch.empty((ns1, st1), dtype=torch.int8, device=dev)
        r = torch.empty(ns1, dtype=torch.int8, device=dev)
        ctx.synth_dense_device(a.seed, t * ns1, b[:, :n1], q[:, :n1], r)
        t1.append((b[:, :n1], q[:, :n1], r))
    res = [torch.empty(ns1 * SITE_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in t1]
    m1 = min(0.001, 100.0 / n1)

    def fn1(j):
        b, q, r = t1[j % 2]
        ctx.lrt_dense_device(b, q, r, m1, res[j % 2])
    n_calls = 40
    dt, prof = timed_calls(ctx, fn1, n_calls)
    em_ms = prof["em_ms"] / max(1, prof["em_launches"])
    hist_ms = prof["hist_ms"] / max(1, prof["hist_launches"])
    rec = results_from_tensor(res[0]).copy()
    # (untimed) the same tile with every subset run: the passes the REFERENCE runs on it, and that nothing else of a record moves
    ctx.join(); ctx.set_tuning("em_prune", 0)
    dt_all, _ = timed_calls(ctx, fn1, n_calls // 2)
    rec_all = results_from_tensor(res[0]).copy()
    ctx.join(); ctx.set_tuning("em_prune", 1)
    ref_passes = float(rec_all["n_passes"].astype("int64").sum())
    legs["config1_1e4x1e4"] = {
        "workload": f"BASELINE configs[1]: synthetic pileup {ns1} sites x {n1} samples, EM to convergence, "
                    f"{n_calls} calls of {ns1} sites",
        "value": n_calls * ns1 / dt, "unit": "sites/s", "ms_per_call": dt / n_calls * 1e3,
        "bound_by": "stage 2: region_kernel (FP64 VALU issue)",
        "roofline": em_roofline(rec, em_ms, dt / n_calls * 1e3, depth=n1),
        "hist_wave_kernel_ms_per_call_under_the_em": hist_ms,     # 0.05 ms alone (profiles/r02_kernel_stats_legs.csv)
        "value_with_every_subset_run": (n_calls // 2) * ns1 / dt_all,
        "em_passes_per_site_of_the_reference": ref_passes / ns1,
        "records_identical_with_every_subset_run_except_the_run_counts": bool(_same_but_run_counts(rec, rec_all)),
    }
    del t1, res
    legs["producer_bgzf_1e5_coverage10pct"] = producer_leg(ctx, min_af, np, torch, dev)
    return legs


def producer_leg(ctx, min_af, np, torch, dev, n=100_000, cov=0.1, batch=500, tile_pos=435, n_tiles=2):
    """SURVEY 8(f1), the producer of the hot path's input: temp-batch BGZF blocks -> inflate -> line index -> token
    parse -> LRT -> records, tile by tile through bvc_pileup_begin_bgzf / bvc_pileup_finish_called as the host program calls them
    (compressed blocks in page-locked memory, host pointers: the link is inside the timed region).  The batches are written by the
    host library's generator (the reference's text form, zlib level 6); the tiles are cut where the host program would cut them.
    The inflate kernel is then timed alone on the same blocks, device-resident."""
    import ctypes as C
    import shutil
    import tempfile
    import zlib
    from basevarc_amd import build as bld
    from basevarc_amd.lib import BVC_PTR_DEVICE, _dev_ptr
    _, hostlib = bld.build_host()
    H = C.CDLL(hostlib)
    H.bvchost_write_synth_batches.restype = C.c_int64
    H.bvchost_write_synth_batches.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32]
    npos = tile_pos * n_tiles
    d = tempfile.mkdtemp(prefix="bvc_bench_producer_")
    try:
        out = os.path.join(d, "p.out")
        os.makedirs(out + ".tmp.thread.0", exist_ok=True)
        entries = H.bvchost_write_synth_batches(out.encode(), n, npos, 1, batch, int(round(cov * 1000)), 11, 0)
        assert entries > 0
        files = sorted(os.listdir(out + ".tmp.thread.0"), key=lambda f: int(f.split(".")[1]))       # batch.<i>: sample order
        per_batch = []                                           # per batch: [(payload, isize, crc, lines ending in the block)], skip, samples
        for f in files:
            raw = open(os.path.join(out + ".tmp.thread.0", f), "rb").read()
            at, blks, first = 0, [], True
            skip = n_in = 0
            while at < len(raw):
                bsize = (raw[at + 16] | (raw[at + 17] << 8)) + 1
                payload = raw[at + 18:at + bsize - 8]
                crc, isize = int.from_bytes(raw[at + bsize - 8:at + bsize - 4], "little"), int.from_bytes(raw[at + bsize - 4:at + bsize], "little")
                at += bsize
                if isize == 0:
                    continue
                text = zlib.decompress(payload, -15)
                if first:
                    skip = text.index(b"\n") + 1
                    n_in = text[:skip].count(b"\t")
                    first = False
                    nl = text.count(b"\n") - 1
                else:
                    nl = text.count(b"\n")
                blks.append((payload, isize, crc, nl))
            per_batch.append((blks, skip, n_in))
        nb = len(per_batch)
        # the CPU path beside it: the host library's own block inflate and token parser (what the host program's CPU feed runs per
        # position loop) on the first eight batches, one thread, scaled to all of them
        H.bvchost_bench_inflate.restype = C.c_double
        H.bvchost_bench_inflate.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int64)]
        H.bvchost_bench_parse.restype = C.c_double
        H.bvchost_bench_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int64)]
        cpu_infl = cpu_parse = 0.0
        n_cpu = min(8, nb)
        for f in files[:n_cpu]:
            raw = open(os.path.join(out + ".tmp.thread.0", f), "rb").read()
            ob, ne = C.c_int64(0), C.c_int64(0)
            t_i = H.bvchost_bench_inflate(raw, len(raw), 1, C.byref(ob))
            text = b"".join(zlib.decompress(raw[a:b], -15) for a, b in _bgzf_payload_ranges(raw))
            text = text[text.index(b"\n") + 1:]
            t_p = H.bvchost_bench_parse(text, len(text), 1, C.byref(ne))
            assert t_i > 0 and t_p > 0 and ob.value > 0
            cpu_infl += t_i; cpu_parse += t_p
        cpu_seconds = (cpu_infl + cpu_parse) * nb / n_cpu
        sample0 = np.concatenate([[0], np.cumsum([p[2] for p in per_batch])[:-1]]).astype(np.int32)
        n_in_batch = np.array([p[2] for p in per_batch], dtype=np.int32)
        assert int(n_in_batch.sum()) == n
        # tile k takes, of every batch, the blocks up to the one in which its last line ends (what is left of that block stays on the device)
        tiles = []
        used = [0] * nb
        for k in range(n_tiles):
            comp, blocks, bob = bytearray(), [], []
            for b, (blks, _, _) in enumerate(per_batch):
                want, have, took = (k + 1) * tile_pos, sum(x[3] for x in blks[:used[b]]), 0
                while have < want and used[b] < len(blks):
                    payload, isize, crc, nl = blks[used[b]]
                    blocks.append((len(comp), len(payload), isize, crc))
                    comp += payload + b"\0" * ((-len(payload)) % 4)
                    have += nl; used[b] += 1; took += 1
                bob.append(took)
            tiles.append((comp, blocks, bob))
        text_bytes = sum(x[1] for blks, _, _ in per_batch for x in blks)
        comp_bytes = sum(len(t[0]) for t in tiles)
        n_blocks = sum(len(t[1]) for t in tiles)
        cap = max(len(t[0]) for t in tiles) + 64
        addr, pinned = ctx.host_alloc(cap)
        skips = [p[1] for p in per_batch]
        ref = np.zeros(tile_pos, dtype=np.int8)

        def one_pass():
            called = got = 0
            for k, (comp, blocks, bob) in enumerate(tiles):
                pinned[:len(comp)] = np.frombuffer(bytes(comp), dtype=np.uint8)
                t0 = time.perf_counter()
                r = ctx.pileup_begin_bgzf(pinned[:len(comp)], blocks, bob, skips if k == 0 else None, sample0, n_in_batch, tile_pos, k == 0)
                assert r["rc"] == 0 and r["T"] == tile_pos, (k, r["rc"], r["T"], r["error"])
                o = ctx._pileup_finish(tile_pos, r["n_entries"], r["n_indels"], r["indel_text_bytes"], ref, min_af, [0] * 5, None, 0,
                                       called_only=True)
                one_pass.seconds += time.perf_counter() - t0
                called += int(o["results"]["called"].sum()); got += int(o["entry_off"][-1])
            return called, got
        try:
            one_pass.seconds = 0.0
            one_pass()                                           # warm-up: the context's buffers
            one_pass.seconds = 0.0
            reps = 3
            for _ in range(reps):
                called, got = one_pass()
            dt = one_pass.seconds / reps
        finally:
            ctx.host_free(addr)
        assert got > 0 and called > 0, (got, called)
        # the inflate kernel alone on the same blocks, device-resident (bvc_inflate_blocks: inflate_kernel + crc32_kernel)
        BLOCK = np.dtype([("comp_off", "<i8"), ("out_off", "<i8"), ("comp_len", "<i4"), ("isize", "<i4"), ("crc32", "<u4"), ("check_crc", "<u4")])
        tab = np.zeros(n_blocks, dtype=BLOCK)
        allc, i, oat = bytearray(), 0, 0
        for comp, blocks, _ in tiles:
            for (co, cl, isz, crc) in blocks:
                tab[i] = (len(allc) + co, oat, cl, isz, crc, 1)
                oat += isz; i += 1
            allc += comp
        d_comp = torch.from_numpy(np.frombuffer(bytes(allc) + b"\0" * 16, dtype=np.uint8).copy()).to(dev)
        d_tab = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
        d_out = torch.empty(oat + 64, dtype=torch.uint8, device=dev)
        d_st = torch.empty(n_blocks, dtype=torch.int32, device=dev)

        def infl():
            ctx._check(ctx._L.bvc_inflate_blocks(ctx._h, _dev_ptr(d_comp), d_comp.numel(), _dev_ptr(d_tab), n_blocks, _dev_ptr(d_out), d_out.numel(),
                                                 _dev_ptr(d_st), BVC_PTR_DEVICE))
        infl(); ctx.synchronize()
        assert int(d_st.abs().sum()) == 0
        t0 = time.perf_counter()
        for _ in range(5):
            infl()
        ctx.synchronize()
        dti = (time.perf_counter() - t0) / 5
        alg = float(comp_bytes + text_bytes)                     # a compressed byte read and a text byte written per text byte
        return {
            "workload": f"temp-batch BGZF blocks of N = {n} samples at {cov:.0%} coverage ({nb} batches of {batch}, text form, zlib level 6): "
                        f"{n_tiles} tiles of {tile_pos} positions = {n_blocks} blocks, {comp_bytes / 1e6:.1f} MB compressed, {text_bytes / 1e6:.1f} MB of text; "
                        "bvc_pileup_begin_bgzf + bvc_pileup_finish_called per tile, one context, host pointers (compressed blocks page-locked)",
            "value": npos / dt, "unit": "positions/s", "ms_per_tile": dt / n_tiles * 1e3, "entries_parsed": int(got), "entries_written_by_the_generator": int(entries),
            "called_positions": int(called),
            "text_GBs": text_bytes / dt / 1e9, "compressed_GBs_over_the_link": comp_bytes / dt / 1e9,
            "cpu_path": {"value": npos / cpu_seconds, "unit": "positions/s", "cores": 1,
                         "inflate_s": cpu_infl * nb / n_cpu, "parse_s": cpu_parse * nb / n_cpu,
                         "what": f"host/inflate.cpp + host/pileup.cpp (the host program's CPU feed) on {n_cpu} of the {nb} batches, one thread, "
                                 "scaled to all of them; no LRT, no link"},
            "bound_by": "the calls' two round trips and the inflate kernel's latency (one wavefront's serial walk per block); the host program "
                        "runs several contexts side by side (profiles/r05_host/README.txt)",
            "roofline": {"bound": "hbm", "kernel": "inflate_kernel (+ crc32_kernel), alone on the same blocks, device-resident",
                         "achieved": alg / dti / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / dti / 1e9 / HBM_PEAK_GBS,
                         "avg_launch_ms": dti * 1e3, "text_GBs": text_bytes / dti / 1e9, "blocks": int(n_blocks),
                         "algorithmic_bytes_per_launch": alg,
                         "note": "1 compressed + 1 text byte per text byte; the kernel is bound by SCALAR INSTRUCTION ISSUE, not by HBM -- "
                                 "475 k instructions per 64 KiB block, one wavefront per block (DESIGN.md 3.8, profiles/r05_inflate.txt) -- so "
                                 "the fraction of the HBM peak is small by construction; the figure to read is text_GBs"},
        }
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _bgzf_payload_ranges(raw):
    at = 0
    while at < len(raw):
        bsize = (raw[at + 16] | (raw[at + 17] << 8)) + 1
        if int.from_bytes(raw[at + bsize - 4:at + bsize], "little"):
            yield at + 18, at + bsize - 8
        at += bsize


def legs_summary(legs):
    def sig(x):
        return float(f"{x:.4g}")
    out = {}
    for name, leg in legs.items():
        if name == "host_pointer_one_byte":
            out[name] = [sig(leg["ragged_one_byte"]["sites_per_s"]), sig(leg["ragged_one_byte"]["frac"])]
        else:
            out[name] = [sig(leg["value"]), sig(leg["roofline"]["frac"])]
    return out


def em_roofline(rec, em_launch_ms, call_ms, depth=0):
    """FP64-VALU issue roofline of stage 2 where it is the bound: issue slots x 4 cycles against 1024 SIMDs x 2.4 GHz.
    Numerator = E+M passes the call RAN (singleEM calls; record field n_passes -- the engine does not run the subsets a bound rules
    out, include/bvc.h "em_prune") x the slots a pass NEEDS in the region kernel with every lane group busy -- so lockstep idling (a wavefront runs until its slowest fit stops), the site phases
    between the levels and launch tails all show up as lost fraction.  The denominator is the wall time per call: the stage 2
    of consecutive calls runs on two streams side by side.  Both per-pass figures come from profiles/stage2_valu.json and are
    null ("stale") when that file was measured on another em_items.hip."""
    passes = float(rec["n_passes"].astype("int64").sum())
    v, src = stage2_valu()
    needed = v["needed_per_pass"] if v else None
    executed = v["executed_per_pass"].get(str(depth)) if v else None
    peak = N_SIMD * ENGINE_CLOCK_HZ / VALU_CYCLES_PER_WAVE_INST          # wave-instructions per second, whole chip
    ach = passes * needed / (call_ms * 1e-3) if needed and call_ms > 0 else None
    return {"bound": "fp64_valu_issue", "kernel": "region_kernel (em_items.hip)", "achieved": ach / 1e9 if ach else None, "peak": peak / 1e9,
            "unit": "G issue slots/s", "frac": ach / peak if ach else None, "ms_per_call": call_ms,
            "avg_launch_ms": em_launch_ms, "launches_overlap": True,
            "em_passes_per_site": passes / max(1, len(rec)), "issue_slots_per_pass": needed,
            "valu_executed_per_pass": executed,
            "valu_executed_per_reference_pass": (v.get("executed_per_reference_pass") or {}).get(str(depth)) if v else None,
            "valu_source": src,
            "frac_executed": (passes * executed / (call_ms * 1e-3) / peak) if executed and call_ms > 0 else None,
            "note": "no MFMA: the EM is a scalar recurrence per class, not a contraction; peak = 1024 SIMDs x 2.4 GHz / 4 cycles "
                    "(the chip holds about 2.17 GHz under this load)"}


def pmc_traffic(a, n, kernel):
    """HBM bytes per launch of `kernel` from a committed rocprofv3 --pmc pass (profiles/pmc_traffic.json, written by
    tools/pmc_summary.py from the passes of tools/profile_round.sh), corrected as MI355X_MICROARCH.md prescribes.
    Returned only when the pass was taken on this workload shape AND on the kernel source now in the tree (sha256
    of hist_kernel.hip); otherwise (None, reason)."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
    except Exception:
        return None, "no profiles/pmc_traffic.json"
    src = os.path.join(ROOT, "basevarc_amd", "csrc", "hist_kernel.hip")
    from basevarc_amd.build import code_sha16
    sha = code_sha16(src)                                        # comments and blank space do not count
    e = d.get("kernels", {}).get(kernel)
    if not e:
        return None, f"no PMC pass for {kernel} in profiles/pmc_traffic.json"
    if e.get("n_samples") != n or e.get("sites_per_launch") != a.tile_sites or e.get("row_align", 16) != a.row_align:
        return None, "PMC pass was taken on another workload shape"
    if d.get("hist_kernel_sha16") != sha:
        return None, f"stale: PMC pass taken on hist_kernel.hip {d.get('hist_kernel_sha16')}, tree has {sha}"
    return e.get("hbm_bytes_per_launch"), {"file": "profiles/pmc_traffic.json", "hist_kernel_sha16": sha,
                                           "summary": d.get("summary"), "commit": d.get("commit")}


# ------------------------------------------------------------------------------------------------ checks
def record_ok(g, e, floor=1e-6):
    return (int(g["called"]) == e["called"] and [int(x) for x in g["depth"]] == e["depth"]
            and [int(g["alt_base"][k]) for k in range(g["n_alt"])] == e["alt_base"]
            and all(abs(float(g["af"][k]) - e["af"][k]) <= 1e-6 for k in range(e["n_alt"]))
            and abs(float(g["var_qual"]) - e["var_qual"]) <= max(floor, 1e-6 * abs(e["var_qual"])))


def spot_check(tiles, results, min_af, a, np):
    """Part of the CPU leg, after the timed region: 64 sites of tile 0 against the oracle's histogram form."""
    from basevarc_amd.lib import results_from_tensor
    from oracle import orc
    b, q, r = tiles[0]
    res = results_from_tensor(results[0])
    pick = np.linspace(0, b.shape[0] - 1, 64).astype(int)
    bad = 0
    for s in pick:
        cnt = orc.dense_hist(b[s].cpu().numpy(), q[s].cpu().numpy())
        bad += not record_ok(res[s], orc.hist_lrt(cnt, int(r[s].item()), min_af))
    return {"sites_checked": int(len(pick)), "mismatches": int(bad)}


def verify_all(ctx, tiles, results, call, min_af, a, np):
    """every site of the resident dataset against the CPU histogram path (oracle, OpenMP over
    sites): tile by tile, device -> host, one C call per tile."""
    from basevarc_amd.lib import results_from_tensor
    from oracle import orc
    t0 = time.perf_counter()
    for i in range(len(tiles)):
        call(i)                                              # results[i] <- tile i, whatever the timed loop left there
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


def verify_groups_tile(ctx, tiles, grp_results, group_t, call, min_af, a, np):
    """every site of tile 0 against the oracle's restatement of the caller's --group loop
    (histogram form): per-group depths, which groups ran, per-group AF."""
    from basevarc_amd.lib import GROUP_DTYPE
    from oracle import orc
    t0 = time.perf_counter()
    call(0)
    ctx.join()
    ctx.synchronize()
    b, q, r = tiles[0]
    ns = b.shape[0]
    hb, hq, hr = b.cpu().numpy(), q.cpu().numpy(), r.cpu().numpy()
    g = group_t.cpu().numpy()
    gres = grp_results[0].cpu().numpy().view(GROUP_DTYPE).reshape(ns, a.groups)
    bad = ran_total = 0
    worst = 0.0
    for s in range(ns):
        _, gd, ga, ran, pres = orc.dense_site_groups(hb[s], hq[s], int(hr[s]), min_af, g, a.groups, use_hist=True)
        d = float(np.max(np.abs(gres[s]["af"] - ga))) if ga.size else 0.0
        worst = max(worst, d)
        ok = (np.array_equal(gres[s]["depth"], gd) and np.array_equal(gres[s]["ran"], ran)
              and np.array_equal(gres[s]["present"], pres) and d <= 1e-6)
        bad += not ok
        ran_total += int(np.sum(ran))
        if (s + 1) % 1000 == 0:
            log(f"verify groups: {s + 1}/{ns} sites, {bad} mismatches")
    return {"sites_checked": int(ns), "group_runs": int(ran_total), "mismatches": int(bad),
            "max_abs_group_af_diff": worst, "seconds": time.perf_counter() - t0}


def spot_check_groups(tiles, grp_results, group_t, min_af, a, np):
    """8 sites of tile 0 against the oracle's restatement of the caller's --group loop."""
    from basevarc_amd.lib import GROUP_DTYPE
    from oracle import orc
    b, q, r = tiles[0]
    ns = b.shape[0]
    g = group_t.cpu().numpy()
    gres = grp_results[0].cpu().numpy().view(GROUP_DTYPE).reshape(ns, a.groups)
    bad = 0
    pick = np.linspace(0, ns - 1, 8).astype(int)
    for s in pick:
        _, gd, ga, ran, _ = orc.dense_site_groups(b[s].cpu().numpy(), q[s].cpu().numpy(), int(r[s].item()), min_af, g,
                                                  a.groups, use_hist=True)
        ok = (np.array_equal(gres[s]["depth"], gd) and np.array_equal(gres[s]["ran"], ran)
              and np.allclose(gres[s]["af"], ga, rtol=0, atol=1e-6))
        bad += not ok
    return {"sites_checked": int(len(pick)), "mismatches": int(bad)}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cgroup_cpu_quota():
    """CPUs this process may use by the cgroup's quota (cpu.max of cgroup v2, cfs_quota of v1), or None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_baseline(tile, min_af, a, np, gpu_records):
    """The faithful per-sample CPU port (oracle/basetype_oracle.c) on a bounded sample of the same workload: first
    ONE site on one thread (north_star / SURVEY 8d: the reference at --thread 1), then one site per host thread
    (about 10-20 s each at N = 1e6); its answers are compared with the GPU records of the same sites.  Beside it, as
    BASELINE.md section 4 asks: the same port at N = 1e4 and the CPU histogram form (the GPU kernels' algorithm) at
    this N, each on one thread and on all of them, so that the algorithmic and the hardware gain can be told apart.
    `cores` is the number of threads the run was GIVEN; what it got is `effective_parallelism` (its rate over the
    one-thread rate) -- a cgroup CPU quota below the visible core count shows up there, and in `cgroup_cpu_quota`."""
    from oracle import orc
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)                                       # the GPU box's CPU share for one GPU
    k = a.cpu_sites if a.cpu_sites > 0 else cores
    b, q, r = tile
    k = min(k, b.shape[0])
    n_hist = min(max(64, 16 * cores), b.shape[0])                # rows for the histogram-form leg
    hb, hq, hr = b[:n_hist].cpu().numpy(), q[:n_hist].cpu().numpy(), r[:n_hist].cpu().numpy()
    log("cpu baseline: 1 site on 1 thread")
    t0 = time.perf_counter()
    orc.dense_batch(hb[:1], hq[:1], hr[:1], min_af, use_hist=False, threads=1)
    dt1 = time.perf_counter() - t0
    log(f"cpu baseline: {k} sites on {min(cores, k)} threads (about 20 s per site per core)")
    t0 = time.perf_counter()
    exp, used = orc.dense_batch(hb[:k], hq[:k], hr[:k], min_af, use_hist=False, threads=min(cores, k))
    dt = time.perf_counter() - t0
    bad = 0
    for s, e in enumerate(exp):
        # the per-sample double sum of the reference drifts by up to N*u*|loglik|
        bad += not record_ok(gpu_records[s], e, floor=1e-6 + 2e-10 * abs(e["lr_alt"]))

    def rate(fn, sites):
        t = time.perf_counter()
        fn()
        return sites / (time.perf_counter() - t)
    # histogram form at this N (the histogram is built from the rows inside the timed call, as the GPU path does)
    log("cpu baseline: histogram form, 1 thread and all threads")
    h1 = rate(lambda: orc.dense_batch(hb[:16], hq[:16], hr[:16], min_af, use_hist=True, threads=1), 16)
    hall = rate(lambda: orc.dense_batch(hb, hq, hr, min_af, use_hist=True, threads=cores), n_hist)
    # the faithful port at N = 1e4 (BASELINE configs[1]'s depth): sites of the same generator
    log("cpu baseline: faithful port at N = 1e4")
    n4 = 10_000
    m4 = min(0.001, 100.0 / n4)
    sb, sq, sr = orc.synth_tile(a.seed, 0, 8 * cores, n4)
    f1 = rate(lambda: orc.dense_batch(sb[:8], sq[:8], sr[:8], m4, use_hist=False, threads=1), 8)
    fall = rate(lambda: orc.dense_batch(sb, sq, sr, m4, use_hist=False, threads=cores), 8 * cores)
    return {"value": k / dt, "unit": "sites/s", "cores": int(used), "kind": "port", "cpu_model": cpu_model(),
            "effective_parallelism": (k / dt) / (1.0 / dt1), "cgroup_cpu_quota": cgroup_cpu_quota(),
            "visible_cpus": os.cpu_count(),
            "sample": f"first {k} sites of tile 0 at N={a.samples}, one site per thread, faithful per-sample "
                      f"restatement of BaseType ctor+LRT+EM (oracle/basetype_oracle.c), {dt:.1f} s",
            "single_thread": {"value": 1.0 / dt1, "unit": "sites/s", "cores": 1,
                              "sample": f"site 0 of tile 0 at N={a.samples} on one thread, {dt1:.1f} s"},
            "faithful_port_N1e4": {"one_thread": f1, "all_threads": fall, "threads": cores, "unit": "sites/s",
                                   "effective_parallelism": fall / f1,
                                   "sample": f"8 / {8 * cores} sites of the synthetic generator at N=10000"},
            "histogram_form": {"one_thread": h1, "all_threads": hall, "threads": cores, "unit": "sites/s",
                               "effective_parallelism": hall / h1,
                               "sample": f"16 / {n_hist} sites of tile 0 at N={a.samples}: class histogram from the rows, then the "
                                         "EM/LRT on <= 512 classes (ORC_MODE_HIST: the GPU kernels' algorithm on the CPU)"},
            "calibration": "the reference binary cannot be built here (DESIGN.md section 5), so the port's wall time is not "
                           "calibrated against it; SURVEY.md's 17.3 s/site was measured on another CPU (2.1 GHz Xeon)",
            "gpu_check_same_sites": {"sites": k, "mismatches": int(bad)}}


if __name__ == "__main__":
    main()
