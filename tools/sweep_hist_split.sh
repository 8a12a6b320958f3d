#!/bin/bash
# Histogram kernel: parts per site (interleaved chunks, merged with global atomics).  usage: bash tools/sweep_hist_split.sh "1 2 4 8 16"
cd $GRAFT_REPO_ROOT
for s in ${1:-1 2 4 8 16}; do
  BVC_HIST_SPLIT=$s python bench.py --steps 60 --warmup 5 --cpu-sites 0 --no-verify $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('split',$s,'sites/s',round(d['value']),'step',round(d['ms_per_step'],4),'k',d['kernels_ms_per_step'],'GB/s',round(d['roofline']['achieved']))"
done
