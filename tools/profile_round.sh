#!/bin/bash
# Standard per-round capture on the GPU box: bench lines (overlap, serial), rocprofv3 kernel stats of both,
# PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_*) in serial mode.  usage: bash tools/profile_round.sh r01
# Output under gpurun_out/<tag>_*; tools/pmc_summary.py turns it into profiles/.
set -e
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${TAG}_bench_overlap.log 2> $O/${TAG}_bench_overlap.err
python bench.py --no-overlap --cpu-sites 0 > $O/${TAG}_bench_serial.log 2> $O/${TAG}_bench_serial.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_overlap -- python3 $R/bench.py --steps 50 --warmup 2 --cpu-sites 0 --no-verify > $O/${TAG}_trace_overlap.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_serial -- python3 $R/bench.py --steps 50 --warmup 2 --cpu-sites 0 --no-verify --no-overlap > $O/${TAG}_trace_serial.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 6 --warmup 1 --total-sites 16000 --cpu-sites 0 --no-overlap --no-verify > $O/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_SQ -- python3 $R/bench.py --steps 6 --warmup 1 --total-sites 16000 --cpu-sites 0 --no-overlap --no-verify > $O/pmc_SQ.log 2>&1
cd $R
grep '^{' $O/${TAG}_bench_overlap.log | cut -c1-300
grep '^{' $O/${TAG}_bench_serial.log | cut -c1-300
