#!/usr/bin/env python3
"""Wall time of `BaseVarC basetype` on the reference's own test data (test/test.sh:3: 100 BAMs,
chr17:41197700-41276155, -q 20; 78,455 candidate positions, 66,614 covered) -- BASELINE configs[0], the plumbing case.
Load phase (BAM -> temp batches) and compute phase (temp batches -> libbvc -> CVG/VCF) timed separately.  Needs a GPU.
usage: python tools/time_testdata.py [threads ...]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from basevarc_amd import build as b  # noqa: E402
from tests import hostref  # noqa: E402

exe, _ = b.build_host()
for thread in [int(x) for x in sys.argv[1:]] or [1, 4]:
    for fmt in ("text", "raw"):
        with tempfile.TemporaryDirectory() as d:
            fa = hostref.write_fasta(os.path.join(d, "chr17.fa"))
            lst = hostref.write_bam_list(os.path.join(d, "bam.list"))
            out = os.path.join(d, "test.out")
            base = [exe, "basetype", "-q", "20", "-t", str(thread), "-b", "10", "-i", lst, "-s", hostref.REGION, "-r", fa,
                    "-o", out, "--tmp-format", fmt]
            t0 = time.perf_counter()
            subprocess.run(base + ["--load"], check=True, capture_output=True)
            t1 = time.perf_counter()
            subprocess.run(base + ["--rerun"], check=True, capture_output=True)
            t2 = time.perf_counter()
            print(json.dumps({"threads": thread, "tmp_format": fmt, "load_s": round(t1 - t0, 2), "compute_s": round(t2 - t1, 2),
                              "covered_positions_per_s_compute": round(66614 / (t2 - t1))}), flush=True)
