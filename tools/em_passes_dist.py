"""Distribution of EM work per site on a bench tile (GPU): n_passes percentiles, by call state, and EM kernel time
for several tile sizes / caps.  usage: python tools/em_passes_dist.py [n_samples] [n_sites]"""
import sys, time
import numpy as np
import torch
from basevarc_amd import Context, caller_min_af
from basevarc_amd.lib import SITE_DTYPE

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
dev = torch.device("cuda:0")
ctx = Context(0)
b = torch.empty((S, N), dtype=torch.int8, device=dev); q = torch.empty_like(b); r = torch.empty(S, dtype=torch.int8, device=dev)
ctx.synth_dense_device(1, 0, b, q, r)
counts = ctx.hist_dense_device(b, q)
ctx.synchronize()
cnp = counts.cpu().numpy()
res = ctx.lrt_dense_device(b, q, r, caller_min_af(N))
ctx.synchronize()
rec = np.frombuffer(res.cpu().numpy().tobytes(), dtype=SITE_DTYPE)
p = rec["n_passes"].astype(np.int64)
print("sites", S, "N", N, "passes mean", p.mean(), "pcts 50/90/99/max", np.percentile(p, [50, 90, 99]), p.max())
print("fits mean", rec["n_fits"].mean(), "called", rec["called"].mean())
for name, m in (("called", rec["called"] == 1), ("not called", rec["called"] == 0)):
    print(name, m.sum(), "passes mean", p[m].mean(), "max", p[m].max(), "p99", np.percentile(p[m], 99))
d = np.sort(rec["depth"], axis=1)
sec = d[:, 2]
print("corr(passes, second depth)", np.corrcoef(p, sec)[0, 1], "corr(passes, n_kept)", np.corrcoef(p, rec["n_kept"])[0, 1])
h, e = np.histogram(p, bins=[0, 200, 400, 600, 800, 1000, 1500, 2000, 3000, 5000, 100000])
print("hist", list(zip(e[:-1], h)))
# static stride imbalance for grid G
for G in (2048, 2560, 6144):
    w = np.zeros(G); 
    for s in range(S): w[s % G] += p[s]
    print("static grid", G, "max wave passes", w.max(), "mean", w.mean(), "ideal(sum/G)", p.sum() / G)
# EM-only timing via lrt_hist-like path: time lrt_dense minus hist? use profiling
ctx.set_profiling(True)
for cap in (0, 4, 6, 8, 12, 16, 24, 32):
    ctx.set_tuning("em_waves_per_cu", cap)
    for _ in range(3):
        ctx.lrt_dense_device(b, q, r, caller_min_af(N), res)
    ctx.synchronize(); ctx.profile(reset=True)
    for _ in range(5):
        ctx.lrt_dense_device(b, q, r, caller_min_af(N), res)
    ctx.synchronize()
    print("cap", cap, ctx.profile(reset=True))
