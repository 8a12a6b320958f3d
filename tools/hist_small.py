"""Histogram stage on calls with few sites (the split > 1 path).  usage: python tools/hist_small.py"""
import torch
from basevarc_amd import Context
N = 1000000
dev = torch.device("cuda:0")
ctx = Context(0)
for S in (8, 32, 64, 128, 256, 512):
    b = torch.empty((S, N), dtype=torch.int8, device=dev); q = torch.empty_like(b); r = torch.empty(S, dtype=torch.int8, device=dev)
    ctx.synth_dense_device(1, 0, b, q, r)
    c = ctx.hist_dense_device(b, q); ctx.synchronize()
    ref = c.clone()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    e0.record()
    for _ in range(20):
        ctx.hist_dense_device(b, q, c)
    e1.record(); torch.cuda.synchronize()
    assert torch.equal(ref, c)
    ms = e0.elapsed_time(e1) / 20
    print(f"sites {S:4d}: {ms:.4f} ms per call, {2.0 * S * N / (ms * 1e-3) / 1e9:.0f} GB/s")
