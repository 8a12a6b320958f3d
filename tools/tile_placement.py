"""Does the histogram kernel's rate depend on WHERE its two input arrays lie?  (GPU)
The kernel trace of the bench shows per-tile durations that are stable per tile (std 0.5 %) and differ by up to 9 % between tiles.
(1) the same kernel on six separately allocated tile pairs; (2) one big allocation, the bases array fixed and the quals array at a
sweep of byte offsets behind it.  usage: python tools/tile_placement.py [n_sites=4000] [n_samples=1000000]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basevarc_amd import Context

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
stride = (N + 127) // 128 * 128
dev = torch.device("cuda:0")
ctx = Context(0)
counts = torch.empty((S, 512), dtype=torch.int32, device=dev)
r = torch.empty(S, dtype=torch.int8, device=dev)


def timed(b, q, reps=8):
    import time
    ctx.hist_dense_device(b, q, counts); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.hist_dense_device(b, q, counts)
    ctx.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


print("## separately allocated tile pairs (virtual addresses of bases / quals, ms per launch, TB/s)")
keep = []
for t in range(6):
    b = torch.empty((S, stride), dtype=torch.int8, device=dev); q = torch.empty((S, stride), dtype=torch.int8, device=dev)
    ctx.synth_dense_device(1, t * S, b[:, :N], q[:, :N], r)
    ms = timed(b[:, :N], q[:, :N])
    print(f"tile {t}: bases {b.data_ptr():#x} quals {q.data_ptr():#x}  {ms:.4f} ms  {2.0 * S * N / ms / 1e9:.3f} TB/s", flush=True)
    keep.append((b, q))
del keep
torch.cuda.empty_cache()
print("## one allocation: bases at its start, quals `skew` bytes behind the end of the bases array rounded up to 1 GiB")
tile = S * stride
gib = 1 << 30
qbase = (tile + gib - 1) // gib * gib
big = torch.empty(qbase + tile + (64 << 20), dtype=torch.int8, device=dev)
print(f"allocation at {big.data_ptr():#x}", flush=True)
b = big[:tile].view(S, stride)
for skew in (0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22,
             (1 << 21) + 4096, (1 << 20) + 256, 3 * 4096, 5 * 256, 3 << 20, 17 << 20, 33 << 20):
    q = big[qbase + skew:qbase + skew + tile].view(S, stride)
    ctx.synth_dense_device(1, 0, b[:, :N], q[:, :N], r)
    ms = timed(b[:, :N], q[:, :N])
    print(f"skew {skew:>9d}: {ms:.4f} ms  {2.0 * S * N / ms / 1e9:.3f} TB/s", flush=True)
del big, b, q
torch.cuda.empty_cache()
print("## one arena for six tile pairs (one hipMalloc), carved [b0 q0 b1 q1 ...]; then the same arena carved [b0..b5 q0..q5]")
pairs = 6
arena = torch.empty(2 * pairs * tile + (2 << 20), dtype=torch.int8, device=dev)
base = (-arena.data_ptr()) % (2 << 20)                           # carve from a 2 MiB boundary
print(f"arena at {arena.data_ptr():#x} (+{base})", flush=True)
for layout in ("interleaved", "split"):
    for t in range(pairs):
        ob = base + (2 * t * tile if layout == "interleaved" else t * tile)
        oq = base + ((2 * t + 1) * tile if layout == "interleaved" else (pairs + t) * tile)
        b = arena[ob:ob + tile].view(S, stride); q = arena[oq:oq + tile].view(S, stride)
        ctx.synth_dense_device(1, t * S, b[:, :N], q[:, :N], r)
        ms = timed(b[:, :N], q[:, :N])
        print(f"{layout} pair {t}: {ms:.4f} ms  {2.0 * S * N / ms / 1e9:.3f} TB/s", flush=True)
