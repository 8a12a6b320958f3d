#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output (kernel trace, --stats, --pmc passes) into profiles/.

usage: tools/pmc_summary.py <round tag> <dir>[,<dir>...] [n_samples sites_per_launch row_align]
each <dir> holds the pmc_* output directories of one bench configuration (tools/profile_round.sh)

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half the bytes of a wide (16 B/lane) coalesced streaming read, so the read side is doubled
for the streaming histogram kernel; WRITE_SIZE is taken as is.  Counters come from separate --pmc passes.
"""
import csv
import glob
import re
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(lrt_groups_kernel|lrt_kernel)<(\d+)", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    for k in ("region_walk_kernel", "region_kernel", "group_comb_kernel", "group_records_kernel", "hist_dense_slots_kernel", "group_slots_kernel", "hist_dense_groups_bytes_kernel", "hist_dense_groups_slots_kernel", "hist_dense_groups_kernel", "hist_dense_ranges_kernel", "hist_wave_kernel", "hist_packed_groups_kernel", "hist_packed_ranges_kernel", "hist_packed_kernel", "pack_dense_kernel",
              "hist_dense_kernel", "hist_csr_block_kernel", "group_bounds_kernel", "var_qual_kernel",
              "synth_dense_kernel", "sum_groups_kernel", "stream_read_kernel"):
        if k in name:
            return k
    return name.split("(")[0][:60]


def counters(d):
    """Counters of the NEWEST run under d (gpurun merges every call's files into the same directory)."""
    out = defaultdict(lambda: defaultdict(list))
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return out
    f = max(files, key=os.path.getmtime)
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        out[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        out[k]["_dur_ns"].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    return out


def main():
    tag, bases = sys.argv[1], sys.argv[2].split(",")
    n_samples = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
    sites = int(sys.argv[4]) if len(sys.argv) > 4 else 4000
    row_align = int(sys.argv[5]) if len(sys.argv) > 5 else 128
    lines = [f"# rocprofv3 PMC summary ({tag})", "",
             "Per-dispatch averages; each counter group from its own `rocprofv3 --pmc ... --kernel-trace` pass "
             "(serial mode: one kernel on the chip at a time).", ""]
    summary = {}
    for base in bases:
        for sub in sorted(glob.glob(os.path.join(base, "pmc_*"))):
            if not os.path.isdir(sub):
                continue
            c = counters(sub)
            lines += [f"## {os.path.basename(base.rstrip('/'))}: pass {os.path.basename(sub)}", "",
                      "| kernel | dispatches | avg ms | counter | avg value |", "|---|---|---|---|---|"]
            for k in sorted(c):
                if k.startswith("__amd") or k in ("synth_dense_kernel", "stream_read_kernel"):
                    continue
                dur = c[k].pop("_dur_ns")
                # group mode launches both the any-order and the column-range kernel; the one whose turn it is not
                # returns at once (a few microseconds) and has nothing to report
                if k.startswith("hist_") and sum(dur) / len(dur) < 2e4:
                    continue
                # skip warm-up dispatch of each kernel
                for name, vals in sorted(c[k].items()):
                    v = vals[1:] if len(vals) > 2 else vals
                    dd = dur[1:] if len(dur) > 2 else dur
                    lines.append(f"| {k} | {len(vals)} | {sum(dd) / len(dd) / 1e6:.4f} | {name} | {sum(v) / len(v):.6g} |")
                    summary.setdefault(k, {})[name] = sum(v) / len(v)
                    summary[k]["ms_under_pmc_" + name] = sum(dd) / len(dd) / 1e6
            lines.append("")
    # HBM traffic per launch of every histogram kernel that has a FETCH_SIZE pass -> profiles/pmc_traffic.json, keyed by
    # kernel and stamped with the sha256 of hist_kernel.hip (bench.py drops the figure when the source has changed)
    import hashlib
    import subprocess
    sys.path.insert(0, ROOT)
    from basevarc_amd.build import code_sha16
    sha = code_sha16(os.path.join(ROOT, "basevarc_amd", "csrc", "hist_kernel.hip"))
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = None
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        doc = json.load(open(path))
        if doc.get("hist_kernel_sha16") != sha or "kernels" not in doc:
            doc = {}
    except Exception:
        doc = {}
    doc.update({"hist_kernel_sha16": sha, "commit": commit, "summary": f"profiles/{tag}_pmc_summary.md",
                "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of 16 B/lane streams), WRITE_SIZE KiB x 1024"})
    doc.setdefault("kernels", {})
    for kname in ("hist_dense_kernel", "hist_dense_groups_kernel", "hist_dense_groups_slots_kernel", "hist_dense_ranges_kernel", "hist_packed_kernel",
                  "hist_packed_groups_kernel", "hist_packed_ranges_kernel"):
        bps = 1 if "packed" in kname else 2          # bytes per (site, sample) of the kernel's input layout
        alg = float(bps) * n_samples * sites
        h = summary.get(kname, {})
        if "FETCH_SIZE" not in h or h["FETCH_SIZE"] < 1000:        # the kernel that returned at once has no traffic
            continue
        # gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide (16 B/lane) coalesced streaming read -- checked on a
        # known byte count with tools/micro/read_bw.hip.  The doubling is applied ONLY to the seven streaming histogram
        # kernels listed above, whose input is read that way; the byte-wise kernels, group_bounds_kernel, the stage-2
        # kernels and everything else appear in the tables above with their raw counter values, uncorrected.
        fetch = h["FETCH_SIZE"] * 1024 * 2
        write = h.get("WRITE_SIZE", 0.0) * 1024
        doc["kernels"][kname] = {
            "n_samples": n_samples, "sites_per_launch": sites, "row_align": row_align,
            "fetch_size_kib_raw": h["FETCH_SIZE"], "write_size_kib_raw": h.get("WRITE_SIZE"),
            "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write,
            "hbm_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": alg,
            "traffic_over_algorithmic": (fetch + write) / alg,
            "correction_applied": "read = FETCH_SIZE KiB x 1024 x 2 (16 B/lane streaming loads: gfx950 counts half, validated "
                                  "with tools/micro/read_bw.hip); write = WRITE_SIZE KiB x 1024 (as is)"}
        lines += [f"## HBM traffic of {kname} per launch", "",
                  f"- FETCH_SIZE raw {h['FETCH_SIZE']:.1f} KiB -> read bytes (x1024 x2) = {fetch:.4g}",
                  f"- WRITE_SIZE raw {h.get('WRITE_SIZE', 0):.1f} KiB -> write bytes = {write:.4g}",
                  f"- algorithmic bytes ({bps} B x {sites} sites x {n_samples} samples) = {alg:.4g}",
                  f"- traffic / algorithmic = {(fetch + write) / alg:.4f}", ""]
    json.dump(doc, open(path, "w"), indent=1)
    open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.md"), "w").write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
