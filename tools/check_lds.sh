#!/bin/bash
# Diagnostic: the region kernels built with -DBVC_CHECK_LDS (every LDS index derived from LDS contents is checked; a
# violation traps and the process dies with a queue error) under the loads of the bench.  Build first:
#   cd basevarc_amd && BVC_EXTRA_FLAGS=-DBVC_CHECK_LDS python build.py      (and rebuild without the flag afterwards)
# usage on the GPU box: bash tools/check_lds.sh <tag>
set -e
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 200 python bench.py --no-legs --cpu-sites 0 --steps 150 --warmup 2 > $O/headline.json 2> $O/headline.err && echo headline ok >> $O/progress
timeout -k 10 200 python bench.py --packed --no-legs --cpu-sites 0 --steps 100 --warmup 2 > $O/packed.json 2> $O/packed.err && echo packed ok >> $O/progress
timeout -k 10 200 python bench.py --groups 5 --no-legs --cpu-sites 0 --steps 40 --warmup 2 > $O/groups.json 2> $O/groups.err && echo groups ok >> $O/progress
timeout -k 10 300 python bench.py --cpu-sites 0 --steps 5 > $O/legs.json 2> $O/legs.err && echo legs ok >> $O/progress
for cfg in "10000 10000" "100000 4000" "1000000 4000"; do timeout -k 10 200 python tools/em_stage2.py $cfg 60 > $O/s2_${cfg// /_}.txt 2>&1 && echo "stage2 $cfg ok" >> $O/progress; done
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -q -x -k "engine or region or capacity or fuzz or config" > $O/tests.log 2>&1; tail -1 $O/tests.log >> $O/progress
cat $O/progress
grep -il "aperture\|exception\|abort\|violation" $O/*.err $O/*.txt $O/*.log 2>/dev/null || echo "no queue errors"
