#!/bin/bash
# A/B the EM workgroup shape (1 or 4 waves) against the EM grid cap, overlap mode.  usage: bash tools/sweep_em_wpb.sh "8 10 12"
cd $GRAFT_REPO_ROOT
for w in ${1:-8 10 12}; do
 for b in 1 4; do
  BVC_EM_WPB=$b BVC_EM_WAVES_PER_CU=$w python bench.py --steps 4 --warmup 1 --cpu-sites 0 --no-verify --no-legs $2 2>/dev/null | python tools/bench_line.py wpb $b cap $w
 done
done
