#!/usr/bin/env python3
"""Every site of the BASELINE configs[2] workload (1e5 sites x 1e6 samples, generated tile by tile): the records of the
packed entry points against those of the two-byte entry points, byte for byte -- plain calls and group mode (k = 5, both
label orders).  usage: tools/verify_packed.py [total_sites]   (needs a GPU)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from basevarc_amd import Context
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    n, ts, k = 1_000_000, 4000, 5
    stride = (n + 127) // 128 * 128
    ctx = Context(0)
    m = min(0.001, 100.0 / n)
    b = torch.empty((ts, stride), dtype=torch.int8, device="cuda")[:, :n]
    q = torch.empty((ts, stride), dtype=torch.int8, device="cuda")[:, :n]
    r = torch.empty(ts, dtype=torch.int8, device="cuda")
    p = torch.empty((ts, stride), dtype=torch.uint8, device="cuda")[:, :n]
    labels = {"interleaved": torch.from_numpy((np.arange(n) % k).astype(np.uint8)).cuda(),
              "ordered": torch.from_numpy((np.arange(n) * k // n).astype(np.uint8)).cuda()}
    out = {"workload": f"{total} sites x {n} samples, seed 1, tiles of {ts}", "sites": 0, "called": 0,
           "plain_mismatching_sites": 0, "group_mismatching_sites": {"interleaved": 0, "ordered": 0}, "unrepresentable": 0}
    t0 = time.time()
    for s0 in range(0, total, ts):
        ns = min(ts, total - s0)
        ctx.synth_dense_device(1, s0, b[:ns], q[:ns], r[:ns])
        _, bad = ctx.pack_dense_device(b[:ns], q[:ns], p[:ns])
        out["unrepresentable"] += bad
        w = ctx.lrt_dense_device(b[:ns], q[:ns], r[:ns], m)
        g = ctx.lrt_dense_packed_device(p[:ns], r[:ns], m)
        ctx.synchronize()
        out["plain_mismatching_sites"] += int((w.view(ns, -1) != g.view(ns, -1)).any(dim=1).sum())
        from basevarc_amd.lib import results_from_tensor
        out["called"] += int(results_from_tensor(w)["called"].sum())
        if s0 % (5 * ts) == 0:                                    # group mode on every fifth tile (both label orders)
            for lay, gt in labels.items():
                w1, wg = ctx.lrt_dense_groups_device(b[:ns], q[:ns], r[:ns], m, gt, k)
                g1, gg = ctx.lrt_dense_groups_packed_device(p[:ns], r[:ns], m, gt, k)
                ctx.synchronize()
                diff = (w1.view(ns, -1) != g1.view(ns, -1)).any(dim=1) | (wg.view(ns, -1) != gg.view(ns, -1)).any(dim=1)
                out["group_mismatching_sites"][lay] += int(diff.sum())
        out["sites"] += ns
    out["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
