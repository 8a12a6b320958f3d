#!/bin/bash
# Sweep the bench tile size in overlap mode.  usage (GPU box): bash tools/sweep_tile.sh "4000 4096 8000"
cd $GRAFT_REPO_ROOT
for t in ${1:-4000 4096 8000}; do
  tot=$(( (100000 / t) * t ))
  python bench.py --steps $(( 400000 / t )) --warmup 5 --cpu-sites 0 --no-verify --tile-sites $t --total-sites $tot 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('tile',$t,'sites/s',round(d['value']),'step',round(d['ms_per_step'],4),'k',d['kernels_ms_per_step'],'frac',round(d['roofline']['frac'],4))"
done
