"""Stage 2 (EM/LRT) alone on resident histograms: ms per call for both engines, records compared.
usage: python tools/em_stage2.py [n_samples] [n_sites] [repeats]   (GPU box)"""
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from basevarc_amd import Context, caller_min_af
from basevarc_amd.lib import SITE_DTYPE

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
ctx = Context(0)
m = caller_min_af(N)
rows = min(S, max(1, (2 << 30) // N))
counts = torch.empty((S, 512), dtype=torch.int32, device=dev)
r = torch.empty(S, dtype=torch.int8, device=dev)
b = torch.empty((rows, N), dtype=torch.int8, device=dev); q = torch.empty_like(b)
for s0 in range(0, S, rows):
    ns = min(rows, S - s0)
    ctx.synth_dense_device(1, s0, b[:ns], q[:ns], r[s0:s0 + ns])
    ctx.hist_dense_device(b[:ns], q[:ns], counts[s0:s0 + ns])
ctx.synchronize()
del b, q
recs = {}
# engine 1 = one wavefront per site; 0 = item engine; "0a" = item engine with em_prune = 0 (every subset run: the counts of
# passes are then the reference's, and what engine 0 saves shows in the last column)
for engine in (1, "0a", 0):
    ctx.set_tuning("em_engine", 0 if engine == "0a" else engine)
    ctx.set_tuning("em_prune", 0 if engine == "0a" else 1)
    out = ctx.lrt_hist_device(counts, r, m)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        ctx.lrt_hist_device(counts, r, m, out)
    ctx.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / R
    rec = np.frombuffer(out.cpu().numpy().tobytes(), dtype=SITE_DTYPE)
    recs[engine] = rec
    print(f"engine {engine}: {ms:.3f} ms per call of {S} sites x N={N}  ({ms * 1e6 / max(1, int(rec['n_passes'].sum())):.4f} ns per site-pass; "
          f"passes/site {rec['n_passes'].mean():.1f}, called {int(rec['called'].sum())})")
a, c = recs[1], recs["0a"]
z = recs[0].copy(); z["n_passes"] = c["n_passes"]; z["n_fits"] = c["n_fits"]
print("item engine with / without the skipped subsets: every other field the same bytes:", z.tobytes() == c.tobytes(),
      f" passes run {recs[0]['n_passes'].mean():.1f} of {c['n_passes'].mean():.1f} per site")
same_int = all(np.array_equal(a[k], c[k]) for k in ("called", "n_alt", "alt_base", "depth", "n_passes", "n_fits", "n_kept", "kept", "status"))
print("integer fields identical:", same_int, " max |d af|", float(np.abs(a["af"] - c["af"]).max()),
      " max rel d chi", float((np.abs(a["chi"] - c["chi"]) / np.maximum(1.0, np.abs(a["chi"]))).max()),
      " max |d var_qual|", float(np.nanmax(np.abs(a["var_qual"] - c["var_qual"]))))
