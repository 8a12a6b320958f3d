#!/usr/bin/env python3
"""CPU micro-benchmarks of the host feed's two costs per text temp batch (no GPU): block inflate and token parse, on a batch file
written by the host library's own generator.  usage: tools/host_micro.py [samples_in_batch=500] [positions=4000] [coverage_permille=100]"""
import ctypes as C
import gzip
import os
import sys
import tempfile
import shutil
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basevarc_amd import build as b

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
npos = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
cov = int(sys.argv[3]) if len(sys.argv) > 3 else 100
_, hostlib = b.build_host() if os.environ.get("BVC_HOST_MICRO_FULL") else (None, None)
if hostlib is None:
    import subprocess
    HOST = os.path.join(ROOT, "basevarc_amd", "host")
    hostlib = os.path.join(ROOT, "basevarc_amd", "libbvchost.so")
    srcs = [os.path.join(HOST, f) for f in b.HOST_SOURCES + ["capi.cpp"]]
    if not os.path.exists(hostlib) or max(os.path.getmtime(f) for f in srcs + [os.path.join(HOST, h) for h in os.listdir(HOST) if h.endswith(".h")]) > os.path.getmtime(hostlib):
        subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-fPIC", "-pthread", "-I", os.path.join(ROOT, "include"),
                               "-shared", "-o", hostlib] + srcs + ["-lz"])
H = C.CDLL(hostlib)
H.bvchost_write_synth_batches.restype = C.c_int64
H.bvchost_write_synth_batches.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32]
H.bvchost_bench_parse.restype = C.c_double
H.bvchost_bench_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int64)]
H.bvchost_bench_inflate.restype = C.c_double
H.bvchost_bench_inflate.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int64)]
d = tempfile.mkdtemp(prefix="bvc_host_micro_")
try:
    os.makedirs(os.path.join(d, "o.tmp.thread.0"))
    H.bvchost_write_synth_batches(os.path.join(d, "o").encode(), n, npos, 1, n, cov, 7, 0)
    raw = open(os.path.join(d, "o.tmp.thread.0", "batch.0"), "rb").read()
    text = gzip.decompress(raw)
    text = text[text.index(b"\n") + 1:]                           # (the names line is not a position)
    ob = C.c_int64(0)
    ti = min(H.bvchost_bench_inflate(raw, len(raw), 5, C.byref(ob)) for _ in range(3))
    import time
    t0 = time.perf_counter()
    for _ in range(3):
        zlib.decompress(raw, 31) if False else gzip.decompress(raw)
    tz = (time.perf_counter() - t0) / 3
    ent = C.c_int64(0)
    tp = min(H.bvchost_bench_parse(text, len(text), 5, C.byref(ent)) for _ in range(3))
    mb = len(text) / 1e6
    print(f"batch of {n} samples x {npos} positions at {cov / 10:.0f} %: {mb:.1f} MB of text in {len(raw) / 1e6:.2f} MB of BGZF; {ent.value} entries")
    print(f"  inflate (inflate.cpp) {ob.value / ti / 1e6:8.0f} MB/s   {ti / mb * 0.3 * 1e6:6.1f} us per 300 KB position     (python gzip/zlib {len(text) / tz / 1e6:.0f} MB/s)")
    print(f"  parse                 {len(text) / tp / 1e6:8.0f} MB/s   {tp / mb * 0.3 * 1e6:6.1f} us per 300 KB position     {tp / max(1, ent.value) * 1e9:.1f} ns per entry")
finally:
    shutil.rmtree(d, ignore_errors=True)
