#!/bin/bash
# Round 5: the host program's compute phase with its three feeds -- BGZF blocks inflated and parsed on the device (default), inflated on
# the CPU and parsed on the device (BVC_HOST_DEVICE_INFLATE=0), inflated and parsed on the CPU (BVC_HOST_DEVICE_PARSE=0) -- on the
# same synthetic text batches: N = 1e5 samples at 10 % coverage, batches of 500 samples, 1500 positions per thread; without and with
# --group (k = 5).  usage (GPU box): bash tools/r05_host.sh <tag> [threads...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/${1:-r05host}_$(date +%m%d_%H%M%S); mkdir -p $O; shift
nproc > $O/cpus.txt; cat /sys/fs/cgroup/cpu.max >> $O/cpus.txt 2>/dev/null
export BVC_HOST_BENCH_FORMATS=text
export BVC_HOST_BENCH_VARIANTS="${BVC_HOST_BENCH_VARIANTS:-BVC_HOST_DEVICE_INFLATE=1;BVC_HOST_DEVICE_INFLATE=0;BVC_HOST_DEVICE_PARSE=0}"
for t in ${@:-1 4 16}; do
  for g in 0 5; do
    BVC_HOST_BENCH_GROUPS=$g timeout -k 10 900 python tools/host_bench.py 100000 $((${POS_PER_THREAD:-1500} * t)) $t 0.1 500 > $O/host_1e5_t${t}_g$g.jsonl 2> $O/host_1e5_t${t}_g$g.err
    echo "threads $t groups $g rc=$?" | tee -a $O/progress
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/host_*.jsonl")):
    for l in open(f):
        d=json.loads(l)
        if 'tmp_format' in d:
            print(f.split('/')[-1], 'threads', d['threads'], 'groups', d['groups'], d['variant'], 'loop pos/s', d['positions_per_s_in_the_position_loops'], 'end-to-end', d['positions_per_s'], 'vcf', d['vcf_lines'])
            for p in d['profile'][:2]: print('   ', p[:600])
PY
