#!/bin/bash
# Round 5: kernel trace of the host program's compute phase (device inflate + parse), ONE thread, N = 1e5 at 10 %: which kernels a tile costs.
# usage (GPU box): bash tools/r05_host_trace.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/${1:-r05trace}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd $R
export BVC_HOST_BENCH_FORMATS=text BVC_HOST_BENCH_VARIANTS="BVC_HOST_DEVICE_INFLATE=1"
BVC_HOST_BENCH_PREFIX="rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv --" timeout -k 10 600 python tools/host_bench.py 100000 1500 1 0.1 500 > $O/host.jsonl 2> $O/host.err
echo rc=$?
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && column -s, -t < $f | cut -c1-200 | head -30
