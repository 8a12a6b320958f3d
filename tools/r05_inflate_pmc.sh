#!/bin/bash
# Round 5: instruction counts of inflate_kernel (rocprofv3 --pmc, one pass per counter set) on tools/inflate_bench.py's blocks.
# usage (GPU box): bash tools/r05_inflate_pmc.sh <tag> [n_blocks=1024]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/${1:-r05infpmc}; mkdir -p $O
N=${2:-1024}
cd /tmp; export TMPDIR=/tmp; cd $R
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $O/$tag -o p --output-format csv -- python3 tools/inflate_bench.py $N 0.1 6 > $O/$tag.log 2>&1
  echo "$set rc=$?"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "inflate_kernel" in r["Kernel_Name"]:
            t = tot[r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1] += 1
n_launch = None
for k, (v, n) in sorted(tot.items()):
    print(k, v, "rows", n)
PY
