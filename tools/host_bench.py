#!/usr/bin/env python3
"""Throughput of the host side of `BaseVarC basetype` (phase 2: temp-batch text -> tiles -> libbvc -> CVG/VCF)
on synthetic temp-batch files.  Needs a GPU.  usage: tools/host_bench.py [n_samples] [n_positions] [threads]"""
import gzip
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import orc  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    npos = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    thread = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    batch = 500
    from basevarc_amd import build as b
    exe, _ = b.build_host()
    d = tempfile.mkdtemp(prefix="bvc_host_bench_")
    out = os.path.join(d, "bench.out")
    start = 1000
    # reference: all 'A' region; contig "chrS"
    fa = os.path.join(d, "s.fa")
    length = start + npos + 2000
    with open(fa, "w") as f:
        f.write(">chrS\n" + "A" * length + "\n")
    open(fa + ".fai", "w").write(f"chrS\t{length}\t6\t{length}\t{length + 1}\n")
    open(os.path.join(d, "bam.list"), "w").write("".join(f"s{i}.bam\n" for i in range(n)))
    bq = orc.synth_tile(11, 0, npos, n, cov_thr16=45875)            # 70 % coverage
    B, Q = bq[0], bq[1]
    rng = np.random.default_rng(0)
    mapq = rng.integers(20, 61, (npos, n))
    rpr = rng.integers(1, 151, (npos, n))
    strand = rng.integers(0, 2, (npos, n))
    nb = 1 + (n - 1) // batch
    window = npos % thread + npos // thread
    total = 0
    for t in range(thread):
        os.makedirs(f"{out}.tmp.thread.{t}", exist_ok=True)
        lo, hi = min(npos, t * window), (npos if t == thread - 1 else min(npos, (t + 1) * window))
        for ib in range(nb):
            cols = range(ib * batch, min(n, (ib + 1) * batch))
            lines = ["".join(f"S{j}\t" for j in cols) + "\n"]
            for p in range(lo, hi):
                lines.append("".join(f"{B[p, j]},{mapq[p, j]},{Q[p, j]},{rpr[p, j]},{strand[p, j]} " if B[p, j] >= 0 else ". "
                                     for j in cols) + "\n")
            text = "".join(lines).encode()
            total += len(text)
            # BGZF-compatible: plain gzip members are not BGZF; write through the host library's own writer instead
            raw = os.path.join(d, f"raw.{t}.{ib}")
            open(raw, "wb").write(text)
            _bgzf_write(raw, f"{out}.tmp.thread.{t}/batch.{ib}")
            os.remove(raw)
    t0 = time.perf_counter()
    r = subprocess.run([exe, "basetype", "--rerun", "-t", str(thread), "-b", str(batch), "-i", os.path.join(d, "bam.list"),
                        "-s", f"chrS:{start}-{start + npos}", "-r", fa, "-o", out], capture_output=True, text=True,
                       env=dict(os.environ, BVC_HOST_PROFILE="1"))
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    for l in r.stderr.splitlines():
        if l.startswith("[profile]"):
            print(l)
    n_cvg = gzip.decompress(open(out + ".cvg.gz", "rb").read()).count(b"\n") - 3
    n_vcf = sum(1 for l in gzip.decompress(open(out + ".vcf.gz", "rb").read()).split(b"\n") if l and not l.startswith(b"#"))
    print({"n_samples": n, "positions": npos, "threads": thread, "text_MB": total / 1e6, "seconds": dt,
           "positions_per_s": npos / dt, "text_MB_per_s": total / 1e6 / dt, "cvg_lines": n_cvg, "vcf_lines": n_vcf})


def _bgzf_write(src, dst):
    """BGZF via Python: 64 KiB-bounded gzip members with the BC extra field + EOF block (SAM spec 4.1)."""
    import struct
    import zlib
    data = open(src, "rb").read()
    with open(dst, "wb") as f:
        for i in range(0, len(data), 0xff00):
            chunk = data[i:i + 0xff00]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = co.compress(chunk) + co.flush()
            bsize = len(comp) + 25
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + comp +
                    struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
        f.write(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))


if __name__ == "__main__":
    main()
