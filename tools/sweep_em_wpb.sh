#!/bin/bash
# A/B the EM workgroup shape (1 or 4 waves) against the EM grid cap, overlap mode.  usage: bash tools/sweep_em_wpb.sh "8 10 12"
cd $GRAFT_REPO_ROOT
for w in ${1:-8 10 12}; do
 for b in 1 4; do
  BVC_EM_WPB=$b BVC_EM_WAVES_PER_CU=$w python bench.py --steps 100 --warmup 5 --cpu-sites 0 --no-verify $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('wpb',$b,'cap',$w,'sites/s',round(d['value']),'step',round(d['ms_per_step'],4),'k',d['kernels_ms_per_step'])"
 done
done
