#!/usr/bin/env python3
"""One compact line from a bench.py JSON line on stdin (used by the sweep scripts).  usage: bench.py ... | bench_line.py <label>"""
import json
import sys

d = json.loads(sys.stdin.readline())
calls = d["config"]["calls_per_step_per_gpu"]
print(" ".join(sys.argv[1:]), "sites/s", round(d["value"]), "ms/call", round(d["ms_per_step"] / calls, 4),
      "kernels", {k: round(v, 4) for k, v in d["kernels_ms_per_call"].items()}, "frac", round(d["roofline"]["frac"], 4), flush=True)
legs = d.get("legs") or {}
if legs:
    print("    legs:", {k: round(v["value"] / 1e6, 3) for k, v in legs.items()}, flush=True)
