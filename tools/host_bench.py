#!/usr/bin/env python3
"""End-to-end throughput of the host side of `BaseVarC basetype`, phase 2 (temp batches -> tiles -> libbvc -> CVG/VCF),
on synthetic temp-batch files in both forms (the reference's text and the additive binary one).  Needs a GPU.

usage: tools/host_bench.py [n_samples] [n_positions] [threads] [coverage] [batch]
The files are written by the host library's own generator (bvchost_write_synth_batches) under a temp directory.
BVC_HOST_BENCH_FORMATS=text,raw restricts the forms; BVC_HOST_BENCH_INFLATE=1,2,4 runs the compute phase once per value of
BVC_HOST_INFLATE_THREADS on the same files (default: the program's own default only).
BVC_HOST_BENCH_VARIANTS="A=1,B=2;A=0" runs the compute phase once per ';'-separated set of environment settings on the same files
(e.g. "BVC_HOST_DEVICE_PARSE=1;BVC_HOST_DEVICE_PARSE=0"); the outputs of all variants must be identical.
BVC_HOST_BENCH_GROUPS=k adds --group with k population groups (sample j in group j mod k, every 10th sample in none).
"""
import ctypes as C
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    npos = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    thread = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    cov = float(sys.argv[4]) if len(sys.argv) > 4 else 0.7
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else 500
    from basevarc_amd import build as b
    exe, hostlib = b.build_host()
    H = C.CDLL(hostlib)
    H.bvchost_write_synth_batches.restype = C.c_int64
    H.bvchost_write_synth_batches.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32]
    d = tempfile.mkdtemp(prefix="bvc_host_bench_", dir=os.environ.get("BVC_BENCH_TMP", None))
    try:
        start = 1000
        fa = os.path.join(d, "s.fa")
        length = start + npos + 2000
        with open(fa, "w") as f:
            f.write(">chrS\n" + "A" * length + "\n")
        open(fa + ".fai", "w").write(f"chrS\t{length}\t6\t{length}\t{length + 1}\n")
        open(os.path.join(d, "bam.list"), "w").write("".join(f"s{i}.bam\n" for i in range(n)))
        formats = os.environ.get("BVC_HOST_BENCH_FORMATS", "text,bin,raw").split(",")
        inflates = os.environ.get("BVC_HOST_BENCH_INFLATE", "").split(",")
        variants = [dict(kv.split("=", 1) for kv in v.split(",") if kv) for v in os.environ.get("BVC_HOST_BENCH_VARIANTS", "").split(";")]
        runs = [(infl, var) for infl in inflates for var in variants]
        group_args = []
        k_groups = int(os.environ.get("BVC_HOST_BENCH_GROUPS", "0"))
        if k_groups > 0:
            gf = os.path.join(d, "groups.txt")
            open(gf, "w").write("".join(f"S{j} G{j % k_groups:02d}\n" for j in range(n) if j % 10 != 9))
            group_args = ["-g", gf]
        variant_outputs = {}
        for fmt in formats:
            out = os.path.join(d, f"bench_{fmt}.out")
            for t in range(thread):
                os.makedirs(f"{out}.tmp.thread.{t}", exist_ok=True)
            t0 = time.perf_counter()
            entries = H.bvchost_write_synth_batches(out.encode(), n, npos, thread, batch, int(round(cov * 1000)), 11, {"text": 0, "bin": 1, "raw": 2}[fmt])
            gen_s = time.perf_counter() - t0
            assert entries >= 0
            size = sum(os.path.getsize(os.path.join(dp, f)) for t in range(thread)
                       for dp, _, fs in os.walk(f"{out}.tmp.thread.{t}") for f in fs)
            for ii, (infl, var) in enumerate(runs):
                env = dict(os.environ, BVC_HOST_PROFILE="1")
                env.update(var)
                keep = ["--keep_tmp"] if ii + 1 < len(runs) else []       # the batches serve every run of the sweep
                if infl:
                    env["BVC_HOST_INFLATE_THREADS"] = infl
                t0 = time.perf_counter()
                import shlex
                prefix = shlex.split(os.environ.get("BVC_HOST_BENCH_PREFIX", ""))     # e.g. "rocprofv3 --kernel-trace --stats -d <dir> --"
                r = subprocess.run(prefix + [exe, "basetype", "--rerun", "-t", str(thread), "-b", str(batch), "-i", os.path.join(d, "bam.list"),
                                    "-s", f"chrS:{start}-{start + npos}", "-r", fa, "-o", out] + keep + group_args, capture_output=True, text=True, env=env)
                dt = time.perf_counter() - t0
                assert r.returncode == 0, r.stderr[-2000:]
                prof = [l for l in r.stderr.splitlines() if l.startswith("[profile]")]
                n_cvg = gzip.decompress(open(out + ".cvg.gz", "rb").read()).count(b"\n") - 3
                n_vcf = sum(1 for l in gzip.decompress(open(out + ".vcf.gz", "rb").read()).split(b"\n") if l and not l.startswith(b"#"))
                import re
                loops = [float(m.group(1)) for m in (re.search(r"position loop ([0-9.e+-]+) s", l) for l in prof) if m]
                body = [gzip.decompress(open(out + k, "rb").read()) for k in (".cvg.gz", ".vcf.gz")]
                variant_outputs.setdefault(fmt, body)
                assert variant_outputs[fmt] == body, f"variant {var} writes other outputs than the first run"
                print(json.dumps({"tmp_format": fmt, "inflate_threads": infl or "default", "variant": var, "groups": k_groups, "positions_per_s_in_the_position_loops": round(npos / max(loops), 1) if loops else None, "n_samples": n, "positions": npos, "threads": thread, "coverage": cov,
                                  "entries": entries, "batch_files_MB": round(size / 1e6, 1), "generate_s": round(gen_s, 2),
                                  "seconds": round(dt, 3), "positions_per_s": round(npos / dt, 1),
                                  "entries_per_s": round(entries / dt), "cvg_lines": n_cvg, "vcf_lines": n_vcf,
                                  "profile": prof if thread <= 4 else prof[-1:] + prof[:2]}), flush=True)
            for k in (".cvg.gz", ".vcf.gz"):
                os.replace(out + k, os.path.join(d, f"{fmt}{k}"))
        same = all(gzip.decompress(open(os.path.join(d, formats[0] + k), "rb").read()) ==
                   gzip.decompress(open(os.path.join(d, f + k), "rb").read()) for k in (".cvg.gz", ".vcf.gz") for f in formats[1:])
        print(json.dumps({"outputs_of_all_forms_identical": same, "forms": formats}))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
