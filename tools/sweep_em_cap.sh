#!/bin/bash
# Sweep the EM grid cap (waves per CU) in overlap mode.  usage (GPU box): bash tools/sweep_em_cap.sh "6 8 10 12 16"
cd $GRAFT_REPO_ROOT
for w in ${1:-6 8 10 12 16}; do
  BVC_EM_WAVES_PER_CU=$w python bench.py --steps 100 --warmup 5 --cpu-sites 0 --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cap',$w,'sites/s',round(d['value']),'step',round(d['ms_per_step'],4),'k',d['kernels_ms_per_step'])"
done
