#!/bin/bash
# Standard per-round capture on the GPU box: the default bench line (overlap) and a serial one, rocprofv3 kernel
# stats of both, and PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_*) in serial mode for the three histogram kernels of the
# dense entry points: plain, any-order groups, groups ordered by column.  usage: bash tools/profile_round.sh r02
# Output under gpurun_out/<tag>/<part>_<time>/; `python tools/pmc_summary.py <tag> gpurun_out/<tag>/dense,gpurun_out/<tag>/groups_interleaved,gpurun_out/<tag>/groups_ordered`
# (+ gpurun_out/<tag>/packed) turns the passes into profiles/<tag>_pmc_summary.md and profiles/pmc_traffic.json.
set -e
# A gpurun call is limited to 20 minutes: `bash tools/profile_round.sh r03 trace`, `... r03 pmc1`, `... r03 pmc2`, `... r03 align`
# run the parts one call each (no second argument: everything).
TAG=${1:-rXX}
PART=${2:-all}
R=$GRAFT_REPO_ROOT
# every attempt of a part gets a directory of its own: a failing log is never overwritten by a retry
O=$R/gpurun_out/$TAG/${PART}_$(date +%m%d_%H%M%S)
mkdir -p $O
cd $R
if [ $PART = all ] || [ $PART = trace ]; then
python bench.py > $O/bench_overlap.json 2> $O/bench_overlap.err
python bench.py --no-overlap --cpu-sites 0 --no-legs > $O/bench_serial.json 2> $O/bench_serial.err
cd /tmp && export TMPDIR=/tmp
Q="--cpu-sites 0 --no-verify --no-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_overlap -- python3 $R/bench.py --steps 4 --warmup 1 $Q > $O/trace_overlap.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -- python3 $R/bench.py --steps 4 --warmup 1 $Q --no-overlap > $O/trace_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_legs -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sites 0 --no-verify > $O/trace_legs.log 2>&1
fi
cd /tmp && export TMPDIR=/tmp
Q="--cpu-sites 0 --no-verify --no-legs"
P="--steps 2 --warmup 1 --total-sites 16000 $Q --no-overlap"
CFGS=()
if [ $PART = all ] || [ $PART = pmc1 ]; then CFGS+=("dense:" "groups_interleaved:--groups 5" "groups_ordered:--groups 5 --group-layout ordered"); fi
if [ $PART = all ] || [ $PART = pmc2 ]; then CFGS+=("packed:--packed" "packed_groups_interleaved:--packed --groups 5" "packed_groups_ordered:--packed --groups 5 --group-layout ordered"); fi
for cfg in "${CFGS[@]}"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$name/pmc_$c -- python3 $R/bench.py $P $flags > $O/${name}_pmc_$c.log 2>&1
  done
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/$name/pmc_SQ -- python3 $R/bench.py $P $flags > $O/${name}_pmc_SQ.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/$name/pmc_SQ2 -- python3 $R/bench.py $P $flags > $O/${name}_pmc_SQ2.log 2>&1
done
cd $R
if [ $PART = all ] || [ $PART = align ]; then
for a in 16 128 256 1024 4096; do
  python bench.py --row-align $a --steps 5 --warmup 1 $Q 2>/dev/null | python tools/bench_line.py row_align $a | tee -a $O/row_align.txt
done
fi
[ -f $O/bench_overlap.json ] && cut -c1-400 $O/bench_overlap.json
find $O -name "*.csv" -size +8M -delete
echo part $PART done
