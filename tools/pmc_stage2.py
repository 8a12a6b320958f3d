"""Per-kernel counter averages of the newest rocprofv3 --pmc CSV under a directory (stage-2 kernels of em_stage2.py).
usage: python tools/pmc_stage2.py <dir> [<dir> ...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(region_wide_probe_kernel|region_kernel<\d+>|classes_kernel|fit_kernel|decide_kernel|lrt_groups_kernel|lrt_kernel|var_qual_kernel|group_comb_kernel|group_records_kernel|hist_\w+kernel|synth_dense_kernel)", name)
    return m.group(1) if m else name[:40]


for d in sys.argv[1:]:
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(d, "no counter csv"); continue
    f = max(files, key=os.path.getmtime)
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"]) + f" grid{int(row['Grid_Size']) // max(1, int(row['Workgroup_Size']))}"
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        acc[k]["_dur_us"].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    print("##", os.path.relpath(d))
    for k, c in sorted(acc.items()):
        print(f"{k:40s}", " ".join(f"{name}={sum(v) / len(v):.4g}" for name, v in sorted(c.items())), f"n={len(c['_dur_us'])}")
