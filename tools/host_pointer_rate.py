#!/usr/bin/env python3
"""PCIe-inclusive rate of the dense entry point when the boundary is handed HOST buffers (BVC_PTR_HOST): the tile is
staged through device memory in chunks, the upload of chunk i+1 under the kernels of chunk i (bvc_api.hip, run_chunks).
Pageable and pinned host memory.  usage: tools/host_pointer_rate.py [n_sites] [n_samples]   (needs a GPU)"""
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
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    ctx = Context(0)
    min_af = min(0.001, 100.0 / n)
    db = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    dq = torch.empty((ns, n), dtype=torch.int8, device="cuda")
    dr = torch.empty(ns, dtype=torch.int8, device="cuda")
    ctx.synth_dense_device(1, 0, db, dq, dr)
    ctx.synchronize()
    ref = dr.cpu().numpy()
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            hb = torch.empty((ns, n), dtype=torch.int8, pin_memory=True); hq = torch.empty((ns, n), dtype=torch.int8, pin_memory=True)
            hb.copy_(db); hq.copy_(dq)
            b, q = hb.numpy(), hq.numpy()
        else:
            b, q = db.cpu().numpy(), dq.cpu().numpy()
        for chunk_kib in (524288, 131072):
            ctx.set_tuning("host_chunk_kib", chunk_kib)
            ctx.lrt_dense(b[:64], q[:64], ref[:64], min_af)                       # warm the staging buffers
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                rec = ctx.lrt_dense(b, q, ref, min_af)
                best = min(best, time.perf_counter() - t0)
            print(json.dumps({"layout": "two bytes per sample", "host_memory": kind, "host_chunk_kib": chunk_kib, "n_sites": ns, "n_samples": n,
                              "seconds": round(best, 4), "sites_per_s": round(ns / best, 1),
                              "host_to_device_GBs": round(2.0 * ns * n / best / 1e9, 2), "called": int(rec["called"].sum())}))
        # the packed layout (one byte per sample, written by the producer on the host): half the PCIe bytes
        p = np.where((b >= 0) & (b < 4) & (q >= 0) & (q < 63), (b.astype(np.uint8) << 6) | q.astype(np.uint8), 0xFF).astype(np.uint8)
        if kind == "pinned":
            hp = torch.empty((ns, n), dtype=torch.uint8, pin_memory=True); hp.copy_(torch.from_numpy(p)); p = hp.numpy()
        ctx.set_tuning("host_chunk_kib", 524288)
        ctx.lrt_dense_packed(p[:64], ref[:64], min_af)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            recp = ctx.lrt_dense_packed(p, ref, min_af)
            best = min(best, time.perf_counter() - t0)
        print(json.dumps({"layout": "packed, one byte per sample", "host_memory": kind, "n_sites": ns, "n_samples": n,
                          "seconds": round(best, 4), "sites_per_s": round(ns / best, 1),
                          "host_to_device_GBs": round(1.0 * ns * n / best / 1e9, 2),
                          "records_identical": bool(recp.tobytes() == rec.tobytes())}))


if __name__ == "__main__":
    main()
