#!/bin/bash
# Group mode (k = 5), overlap: EM grid cap sweep.  usage: bash tools/sweep_groups.sh "4 6 8 12" ["--group-layout ordered"]
cd $GRAFT_REPO_ROOT
for w in ${1:-8}; do
  BVC_EM_WAVES_PER_CU=$w python bench.py --groups 5 $2 --steps 4 --warmup 1 --cpu-sites 0 --no-verify --no-legs --total-sites 40000 2>/dev/null | python tools/bench_line.py cap $w $2
done
