#!/bin/bash
# EM grid cap sweep (waves per CU) for the EM-bound shapes.  usage: bash tools/sweep_em_cap.sh "8 12 16 24"
cd $GRAFT_REPO_ROOT
for w in ${1:-8 12 16 24}; do
  BVC_EM_WAVES_PER_CU=$w python bench.py --samples 10000 --total-sites 40000 --tile-sites 10000 --steps 10 --warmup 2 --cpu-sites 0 --no-verify --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('N=1e4 tile 1e4 cap',$w,'sites/s',round(d['value']),'ms/call',round(d['ms_per_step']/d['config']['calls_per_step_per_gpu'],4),d['kernels_ms_per_call'])"
  BVC_EM_WAVES_PER_CU=$w python bench.py --samples 10000 --total-sites 40000 --tile-sites 10000 --steps 10 --warmup 2 --cpu-sites 0 --no-verify --no-legs --no-overlap 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('N=1e4 tile 1e4 serial cap',$w,'sites/s',round(d['value']),'ms/call',round(d['ms_per_step']/d['config']['calls_per_step_per_gpu'],4),d['kernels_ms_per_call'])"
done
for w in ${2:-6 8 12}; do
  BVC_EM_WAVES_PER_CU=$w python bench.py --steps 4 --warmup 1 --cpu-sites 0 --no-verify --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('N=1e6 cap',$w,'sites/s',round(d['value']),'ms/call',round(d['ms_per_step']/d['config']['calls_per_step_per_gpu'],4),d['kernels_ms_per_call'])"
done
