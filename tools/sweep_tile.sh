#!/bin/bash
# Sweep the sites per library call in overlap mode.  usage (GPU box): bash tools/sweep_tile.sh "4000 4096 8000"
cd $GRAFT_REPO_ROOT
for t in ${1:-4000 4096 8000}; do
  tot=$(( (100000 / t) * t ))
  python bench.py --steps 4 --warmup 1 --cpu-sites 0 --no-verify --no-legs --tile-sites $t --total-sites $tot 2>/dev/null | python tools/bench_line.py tile $t
done
