"""The ragged (CSR) bench leg by itself: N calls of bvc_lrt_csr on device pointers in overlap mode; prints ms per call.
Under `rocprofv3 --kernel-trace` the trace shows which kernels of consecutive calls really run side by side.
usage: python tools/csr_probe.py [sites] [calls] [samples] [coverage]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basevarc_amd import Context
from basevarc_amd.lib import SITE_DTYPE
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
cov = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
dev = torch.device("cuda:0")
ctx = Context(0, stream=torch.cuda.current_stream())
ctx.set_overlap(True)
min_af = min(0.001, 100.0 / N)
tiles = []
slice_sites = 500
tb = torch.empty((slice_sites, N), dtype=torch.int8, device=dev); tq = torch.empty_like(tb)
for t in range(2):
    r = torch.empty(S, dtype=torch.int8, device=dev)
    pb, pq, cn = [], [], []
    for c0 in range(0, S, slice_sites):
        ns = min(slice_sites, S - c0)
        ctx.synth_dense_device(1, 10_000_000 + t * S + c0, tb[:ns], tq[:ns], r[c0:c0 + ns], cov_thr16=int(round(cov * 65536)))
        ctx.synchronize()
        m = tb[:ns] >= 0
        cn.append(m.sum(dim=1)); pb.append(tb[:ns][m]); pq.append(tq[:ns][m])
    offs = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    offs[1:] = torch.cumsum(torch.cat(cn).to(torch.int64), 0)
    tiles.append((offs, torch.cat(pb), torch.cat(pq), r))
del tb, tq
res = [torch.empty(S * SITE_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in tiles]
def call(j):
    o, b, q, r = tiles[j % 2]
    ctx.lrt_csr_device(o, b, q, r, min_af, res[j % 2])
for j in range(4): call(j)
ctx.join(); torch.cuda.synchronize()
t0 = time.perf_counter()
for j in range(K): call(j)
ctx.join(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{S} sites x {N} samples at {cov:.0%}: {dt / K * 1e3:.4f} ms per call, {K * S / dt:.4g} sites/s")
ctx.close()
