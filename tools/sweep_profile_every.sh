#!/bin/bash
# Cost of the per-launch HIP timing events on the step time.  usage: bash tools/sweep_profile_every.sh "1 4 1000"
cd $GRAFT_REPO_ROOT
for k in ${1:-1 4 1000}; do
  BVC_EM_WPB=4 BVC_EM_WAVES_PER_CU=8 python bench.py --steps 100 --warmup 5 --cpu-sites 0 --no-verify --profile-every $k 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('every',$k,'sites/s',round(d['value']),'step',round(d['ms_per_step'],4),'k',d['kernels_ms_per_step'])"
done
