#!/usr/bin/env python3
"""CPU baselines of SURVEY.md 8d / BASELINE.md 4 on the GPU box's host cores (no GPU used).

Times the faithful per-sample port of the reference path (oracle/basetype_oracle.c) and its histogram form,
at N = 1e4 and N = 1e6, on 1 thread and on all (<= 16) cores.  Writes profiles/<tag>_cpu_baselines.json.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    cores = min(16, len(os.sched_getaffinity(0)))
    model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")
    out = {"cpu_model": model, "cores_used": cores, "rows": []}
    for n, k1, kall in ((10000, 8, 128), (1000000, 1, cores)):
        m = min(0.001, 100.0 / n)
        b, q, r = orc.synth_tile(1, 0, max(k1, kall), n)
        for name, use_hist in (("faithful per-sample port", False), ("histogram form", True)):
            for threads, k in ((1, k1), (cores, kall)):
                if use_hist:
                    k = max(k, kall)
                t0 = time.perf_counter()
                orc.dense_batch(b[:k], q[:k], r[:k], m, use_hist=use_hist, threads=threads)
                dt = time.perf_counter() - t0
                row = {"n_samples": n, "path": name, "threads": threads, "sites": k, "seconds": dt, "sites_per_s": k / dt}
                out["rows"].append(row)
                print(row, flush=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_cpu_baselines.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
