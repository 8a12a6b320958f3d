#!/usr/bin/env python3
"""Throughput of inflate_kernel.hip (bvc_inflate_blocks, device pointers) on temp-batch pileup text: n blocks of 65,280 bytes of text at
the given coverage, deflated by zlib at the given level (the reference's bgzf_write uses zlib's default, 6; the bench's generator 1).
usage: tools/inflate_bench.py [n_blocks=4096] [coverage=0.1] [level=6]   (GPU)"""
import os
import sys
import time
import zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from basevarc_amd import Context
from basevarc_amd.lib import BVC_PTR_DEVICE, _dev_ptr

n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rng = np.random.default_rng(3)
BLOCK = np.dtype([("comp_off", "<i8"), ("out_off", "<i8"), ("comp_len", "<i4"), ("isize", "<i4"), ("crc32", "<u4"), ("check_crc", "<u4")])
CHECK = int(os.environ.get("BVC_BENCH_CRC", "1"))
# 64 different blocks of text, repeated: the kernel does not care, the host's generator is the slow part
texts = []
for k in range(64):
    toks = [("%d,%d,%d,%d,%d " % (rng.integers(4), rng.integers(20, 61), rng.integers(10, 41), rng.integers(1, 150), rng.integers(2)))
            if rng.random() < cov else ". " for _ in range(40000)]
    texts.append("".join(toks).encode()[:65280])
comps = []
for t in texts:
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comps.append(co.compress(t) + co.flush())
comp = bytearray()
tab = np.zeros(n_blocks, dtype=BLOCK)
for i in range(n_blocks):
    c = comps[i % 64]
    tab[i] = (len(comp), i * 65280, len(c), 65280, zlib.crc32(texts[i % 64]) & 0xffffffff, CHECK)
    comp += c + b"\0" * ((-len(c)) % 4)
ctx = Context(0)
dev = torch.device("cuda:0")
d_comp = torch.from_numpy(np.frombuffer(bytes(comp) + b"\0" * 16, dtype=np.uint8).copy()).to(dev)
d_tab = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
d_out = torch.empty(n_blocks * 65280, dtype=torch.uint8, device=dev)
d_st = torch.empty(n_blocks, dtype=torch.int32, device=dev)
L = ctx._L


def run():
    ctx._check(L.bvc_inflate_blocks(ctx._h, _dev_ptr(d_comp), d_comp.numel(), _dev_ptr(d_tab), n_blocks, _dev_ptr(d_out), d_out.numel(), _dev_ptr(d_st),
                                    BVC_PTR_DEVICE))


run(); ctx.synchronize()
assert int(d_st.abs().sum()) == 0
got = d_out[:65280 * 64].cpu().numpy().tobytes()
assert all(got[k * 65280:(k + 1) * 65280] == texts[k] for k in range(64))
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    run()
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{n_blocks} blocks of 65280 bytes, coverage {cov}, zlib level {level}: compressed {len(comp) / n_blocks:.0f} bytes per block "
      f"({65280 * n_blocks / len(comp):.2f} x); {dt * 1e3:.2f} ms per launch = {65280 * n_blocks / dt / 1e9:.2f} GB/s of text, "
      f"{dt / n_blocks * 1024 * 1e6:.1f} us per block per 1024 wavefront slots")
