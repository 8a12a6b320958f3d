#!/bin/bash
# Round 5: kernel trace of the host program's compute phase with EIGHT threads (device inflate + parse), N = 1e5 at 10 %: how busy the device is.
# usage (GPU box): bash tools/r05_host_trace8.sh <tag> [threads=8] [positions per thread=6000]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/${1:-r05trace8}; mkdir -p $O
T=${2:-8}; P=${3:-6000}
cd /tmp; export TMPDIR=/tmp; cd $R
export BVC_HOST_BENCH_FORMATS=text BVC_HOST_BENCH_VARIANTS="BVC_HOST_PROFILE=2"
BVC_HOST_BENCH_PREFIX="rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv --" timeout -k 10 900 python tools/host_bench.py 100000 $((P * T)) $T 0.1 500 > $O/host.jsonl 2> $O/host.err
echo rc=$?
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/trace/**/t_kernel_trace.csv", recursive=True) + glob.glob("$O/trace/t_kernel_trace.csv")
rows = list(csv.DictReader(open(f[0])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
t0, t1 = iv[0][0], max(e for _, e in iv)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = collections.Counter(); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
    tot[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
print("kernels", len(rows), "span %.3f s, device busy (union of kernel intervals) %.3f s = %.1f %%; sum of kernel durations %.3f s" % ((t1 - t0) / 1e9, busy / 1e9, 100.0 * busy / (t1 - t0), sum(tot.values()) / 1e9))
for k, v in tot.most_common(12): print("  %-42s calls %5d  total %8.2f ms  mean %8.1f us" % (k, cnt[k], v / 1e6, v / 1e3 / cnt[k]))
PY
