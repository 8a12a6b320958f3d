#!/bin/bash
# Group mode (k = 5, N = 1e6, labels i % 5) on the GPU box: the any-order histogram kernel A/B -- loads issued ahead
# (BVC_GROUP_PIPE=1) or two chunks loaded then counted (=0) -- timing alone and under the EM kernels, then PMC passes
# (FETCH_SIZE, SQ_*) for each, serial mode.  usage: bash tools/profile_groups.sh <tag> "<BVC_GROUP_PIPE values>"
# Output under gpurun_out/<tag>/; tools/pmc_summary.py <tag> gpurun_out/<tag>/gs<N> summarises a pass set.
set -e
TAG=${1:-rXX_groups}
GS=${2:-"0 1"}
EXTRA=$3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
B="--groups 5 --cpu-sites 0 --no-verify --no-legs --total-sites 24000 $EXTRA"
for gs in $GS; do
  for mode in "" "--no-overlap"; do
    BVC_GROUP_PIPE=$gs python bench.py $B $mode --steps 8 --warmup 1 2>/dev/null | python tools/bench_line.py group_pipe $gs $mode | tee -a $O/timing.txt
  done
done
cd /tmp && export TMPDIR=/tmp
for gs in $GS; do
  export BVC_GROUP_PIPE=$gs
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/gs$gs/pmc_FETCH_SIZE -- python3 $R/bench.py $B --no-overlap --steps 2 --warmup 1 > $O/gs${gs}_pmc_FETCH_SIZE.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/gs$gs/pmc_SQ -- python3 $R/bench.py $B --no-overlap --steps 2 --warmup 1 > $O/gs${gs}_pmc_SQ.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/gs$gs/pmc_SQ2 -- python3 $R/bench.py $B --no-overlap --steps 2 --warmup 1 > $O/gs${gs}_pmc_SQ2.log 2>&1
done
cat $O/timing.txt
