"""A/B of two builds of libbvc on the SAME tile pairs (physical placement differs from process to process, so a variant must be
compared inside one process): ms per launch of bvc_hist_dense for the product library and for basevarc_amd/_variants/libbvc_<v>.so.
usage: python tools/tile_placement_ab.py <variant> [pairs=10]"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from basevarc_amd import Context
from basevarc_amd.lib import BVC_PTR_DEVICE

variant = sys.argv[1]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
S, N = 4000, 1_000_000
stride = (N + 127) // 128 * 128
dev = torch.device("cuda:0")
ctx = Context(0)
L2 = C.CDLL(os.path.join(ROOT, "basevarc_amd", "_variants", f"libbvc_{variant}.so"))
vp, i64 = C.c_void_p, C.c_int64
L2.bvc_create.restype = C.c_int; L2.bvc_create.argtypes = [C.POINTER(vp), C.c_int]
L2.bvc_hist_dense.restype = C.c_int
L2.bvc_hist_dense.argtypes = [vp, i64, i64, i64, vp, vp, vp, C.c_int]
L2.bvc_synchronize.restype = C.c_int; L2.bvc_synchronize.argtypes = [vp]
h2 = vp()
assert L2.bvc_create(C.byref(h2), 0) == 0
counts = torch.empty((S, 512), dtype=torch.int32, device=dev)
counts2 = torch.empty((S, 512), dtype=torch.int32, device=dev)
r = torch.empty(S, dtype=torch.int8, device=dev)


def t_product(b, q, reps=8):
    ctx.hist_dense_device(b, q, counts); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.hist_dense_device(b, q, counts)
    ctx.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


def t_variant(b, q, reps=8):
    args = (h2, S, N, b.stride(0), b.data_ptr(), q.data_ptr(), counts2.data_ptr(), BVC_PTR_DEVICE)
    assert L2.bvc_hist_dense(*args) == 0; L2.bvc_synchronize(h2)
    t0 = time.perf_counter()
    for _ in range(reps):
        L2.bvc_hist_dense(*args)
    L2.bvc_synchronize(h2)
    return (time.perf_counter() - t0) * 1e3 / reps


keep = []
tot = [0.0, 0.0]
for t in range(pairs):
    b = torch.empty((S, stride), dtype=torch.int8, device=dev); q = torch.empty((S, stride), dtype=torch.int8, device=dev)
    ctx.synth_dense_device(1, t * S, b[:, :N], q[:, :N], r); ctx.synchronize()
    a1 = t_product(b[:, :N], q[:, :N]); v1 = t_variant(b[:, :N], q[:, :N]); a2 = t_product(b[:, :N], q[:, :N]); v2 = t_variant(b[:, :N], q[:, :N])
    same = bool(torch.equal(counts, counts2))
    print(f"pair {t}: product {a1:.4f} {a2:.4f} ms   {variant} {v1:.4f} {v2:.4f} ms   counts equal {same}", flush=True)
    tot[0] += min(a1, a2); tot[1] += min(v1, v2)
    keep.append((b, q))
print(f"mean: product {tot[0] / pairs:.4f} ms, {variant} {tot[1] / pairs:.4f} ms")
