#!/bin/bash
# EM grid cap sweep (waves per CU) for the EM-bound shape (N = 1e4) and the headline.  usage: bash tools/sweep_em_cap.sh "8 12 16 24" "6 8 12"
cd $GRAFT_REPO_ROOT
S="--samples 10000 --total-sites 40000 --tile-sites 10000 --steps 10 --warmup 2 --cpu-sites 0 --no-verify --no-legs"
for w in ${1:-8 12 16 24}; do
  BVC_EM_WAVES_PER_CU=$w python bench.py $S 2>/dev/null | python tools/bench_line.py N=1e4 overlap cap $w
  BVC_EM_WAVES_PER_CU=$w python bench.py $S --no-overlap 2>/dev/null | python tools/bench_line.py N=1e4 serial cap $w
done
for w in ${2:-6 8 12}; do
  BVC_EM_WAVES_PER_CU=$w python bench.py --steps 4 --warmup 1 --cpu-sites 0 --no-verify --no-legs 2>/dev/null | python tools/bench_line.py N=1e6 cap $w
done
