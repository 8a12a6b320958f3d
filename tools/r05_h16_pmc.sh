#!/bin/bash
# Round 5: counters of the any-order group histogram on packed rows (k = 5, N = 1e6) with 16 copies of 32-bit counters (group_h16 = 0)
# and with 32 conflict-free copies of 16-bit counter pairs (group_h16 = 1): LDS bank conflicts, LDS busy, VALU instructions.
# Separate --pmc passes, kernel-trace off (gpurun refuses other combinations).  usage (GPU box): bash tools/r05_h16_pmc.sh <tag>
set -e
O=$PWD/gpurun_out/$1; mkdir -p $O
R=$PWD; cd /tmp; export TMPDIR=/tmp; cd $R
for h in 0 1; do
  export BVC_GROUP_H16=$h
  timeout -k 10 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/lds_h16_$h -o p --output-format csv -- python3 bench.py --groups 5 --no-legs --cpu-sites 0 --steps 1 --warmup 0 --total-sites 8000 --packed --no-overlap > $O/lds_h16_$h.log 2>&1
  echo lds $h done >> $O/progress
  timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $O/valu_h16_$h -o p --output-format csv -- python3 bench.py --groups 5 --no-legs --cpu-sites 0 --steps 1 --warmup 0 --total-sites 8000 --packed --no-overlap > $O/valu_h16_$h.log 2>&1
  echo valu $h done >> $O/progress
done
python tools/pmc_stage2.py $O/lds_h16_0 $O/valu_h16_0 $O/lds_h16_1 $O/valu_h16_1 > $O/pmc.txt 2>&1
grep -E "##|hist_packed_groups" $O/pmc.txt
find $O -name "*.csv" -size +2M -delete
