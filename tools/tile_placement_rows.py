"""Round-5 experiment on the headline kernel: does a tile laid out as ONE buffer with a site's base row and quality row adjacent
(quals = bases + stride, row_stride = 2 * stride -- expressible through the existing ABI, no kernel change) even out the tile-to-tile
spread of hist_dense_kernel that two separately allocated 4 GB arrays show (profiles/r04_tile_placement.txt: 1.158 .. 1.261 ms)?
Both layouts are held resident together (as the bench holds its 25 tiles) and timed alternately in one process.
usage: python tools/tile_placement_rows.py [tiles=10] [n_sites=4000] [n_samples=1000000]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basevarc_amd import Context

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
stride = (N + 127) // 128 * 128
dev = torch.device("cuda:0")
ctx = Context(0)
counts = torch.empty((S, 512), dtype=torch.int32, device=dev)
counts2 = torch.empty((S, 512), dtype=torch.int32, device=dev)
r = torch.empty(S, dtype=torch.int8, device=dev)


def timed(b, q, c, reps=8):
    ctx.hist_dense_device(b, q, c); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.hist_dense_device(b, q, c)
    ctx.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


two, one = [], []
for t in range(T):
    b = torch.empty((S, stride), dtype=torch.int8, device=dev); q = torch.empty((S, stride), dtype=torch.int8, device=dev)
    ctx.synth_dense_device(1, t * S, b[:, :N], q[:, :N], r)
    two.append((b[:, :N], q[:, :N]))
    buf = torch.empty((S, 2 * stride), dtype=torch.int8, device=dev)
    bi, qi = buf[:, :N], buf[:, stride:stride + N]              # quals = bases + stride, row stride 2 * stride
    bi.copy_(b[:, :N]); qi.copy_(q[:, :N])
    one.append((bi, qi, buf))
ctx.synchronize(); torch.cuda.synchronize()
tot = [0.0, 0.0]
rows = []
for rnd in range(3):
    for t in range(T):
        a = timed(two[t][0], two[t][1], counts)
        v = timed(one[t][0], one[t][1], counts2)
        if rnd == 0:
            assert torch.equal(counts, counts2)
        rows.append((rnd, t, a, v))
        print(f"round {rnd} tile {t}: two allocations {a:.4f} ms   rows interleaved in one buffer {v:.4f} ms", flush=True)
for t in range(T):
    a = sum(x[2] for x in rows if x[1] == t) / 3; v = sum(x[3] for x in rows if x[1] == t) / 3
    tot[0] += a; tot[1] += v
    print(f"tile {t}: mean of 3: two allocations {a:.4f} ms  interleaved rows {v:.4f} ms  ratio {v / a:.4f}")
print(f"mean over {T} tiles: two allocations {tot[0] / T:.4f} ms ({2.0 * S * N / (tot[0] / T) / 1e9:.3f} TB/s), "
      f"interleaved rows {tot[1] / T:.4f} ms ({2.0 * S * N / (tot[1] / T) / 1e9:.3f} TB/s), ratio {tot[1] / tot[0]:.4f}")
a_all = [x[2] for x in rows]; v_all = [x[3] for x in rows]
print(f"spread: two allocations {min(a_all):.4f} .. {max(a_all):.4f} ms, interleaved rows {min(v_all):.4f} .. {max(v_all):.4f} ms")
