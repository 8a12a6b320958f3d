#!/usr/bin/env python3
"""Stage-2 speed by quality spectrum (GPU): the number of distinct quality values per allele decides the path: <= 32 the
narrow region kernel (em_items.hip), 33..48 the wide one, more (or qualities 0 / 1) the one-wavefront-per-site kernels, where it decides which
lrt_kernel<NS> variant a site takes (<= 32: NS = 2, <= 64: NS = 4, <= 128: NS = 8).  The SURVEY's generator draws
Q from 10..40 (31 values, NS = 2); real Illumina data has ~40 (NS = 4), the int8 range allows 128 (NS = 8).
Tiles are made with torch on the device; 16 sites per spectrum are checked against the oracle's histogram form.
usage: python tools/em_wide_quals.py [n_samples] [n_sites]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from basevarc_amd import Context, caller_min_af  # noqa: E402
from basevarc_amd.lib import results_from_tensor  # noqa: E402
from oracle import orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
dev = torch.device("cuda:0")
ctx = Context(0)
g = torch.Generator(device=dev); g.manual_seed(7)
m = caller_min_af(N)
BINNED = torch.tensor([2, 12, 23, 37], dtype=torch.int8, device=dev)          # NovaSeq-style binned qualities
for name, qlo, qhi in (("binned Q2/12/23/37 (4 values)", 0, 3), ("Q10-40 (31 values)", 10, 40), ("Q2-41 (40 values)", 2, 41), ("Q2-93 (92 values)", 2, 93)):
    q = torch.randint(qlo, qhi + 1, (S, N), generator=g, device=dev, dtype=torch.int8)
    if name.startswith("binned"):
        q = BINNED[q.long()]
    ref = torch.randint(0, 4, (S, 1), generator=g, device=dev, dtype=torch.int8)
    af = torch.where(torch.rand((S, 1), generator=g, device=dev) < 0.2, torch.rand((S, 1), generator=g, device=dev) * 0.3, torch.zeros((S, 1), device=dev))
    u = torch.rand((S, N), generator=g, device=dev)
    true = torch.where(u < af, (ref + 1) % 4, ref.expand(S, N)).to(torch.int8)
    err = torch.rand((S, N), generator=g, device=dev) < torch.pow(10.0, -q.float() / 10.0)
    shift = torch.randint(1, 4, (S, N), generator=g, device=dev, dtype=torch.int8)
    b = torch.where(err, (true + shift) % 4, true).to(torch.int8).contiguous()
    del u, true, err, shift
    r = ref.reshape(S).contiguous()
    out = ctx.lrt_dense_device(b, q, r, m)
    ctx.synchronize()
    ctx.set_profiling(True); ctx.profile(reset=True)
    for _ in range(10):
        ctx.lrt_dense_device(b, q, r, m, out)
    ctx.synchronize()
    p = ctx.profile(reset=True); ctx.set_profiling(False)
    rec = results_from_tensor(out)
    bad = 0
    for s in range(0, S, S // 16):
        e = orc.hist_lrt(orc.dense_hist(b[s].cpu().numpy(), q[s].cpu().numpy()), int(r[s].item()), m)
        ok = (int(rec[s]["called"]) == e["called"] and [int(x) for x in rec[s]["depth"]] == e["depth"]
              and all(abs(float(rec[s]["af"][k]) - e["af"][k]) <= 1e-6 for k in range(e["n_alt"]))
              and (abs(float(rec[s]["var_qual"]) - e["var_qual"]) <= 1e-6 * max(1.0, abs(e["var_qual"]))
                   or (np.isnan(rec[s]["var_qual"]) and np.isnan(e["var_qual"]))))
        bad += not ok
    em = p["em_ms"] / p["em_launches"]
    passes = float(rec["n_passes"].astype(np.int64).mean())
    print(f"{name}: EM {em:.4f} ms per {S} sites = {S / em / 1e3:.2f} M sites/s, {passes:.0f} passes/site, "
          f"{em * 1e6 / (S * passes):.1f} ns per site-pass chip-wide, hist {p['hist_ms'] / p['hist_launches']:.4f} ms, "
          f"called {float(rec['called'].mean()):.3f}, oracle mismatches {bad}/16", flush=True)
    del b, q
