#!/bin/bash
# Histogram kernel: parts per site (interleaved chunks, merged with global atomics).  usage: bash tools/sweep_hist_split.sh "1 2 4 8 16"
cd $GRAFT_REPO_ROOT
for s in ${1:-1 2 4 8 16}; do
  BVC_HIST_SPLIT=$s python bench.py --steps 4 --warmup 1 --cpu-sites 0 --no-verify --no-legs $2 2>/dev/null | python tools/bench_line.py split $s
done
