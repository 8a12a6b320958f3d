#!/bin/bash
# Experiment driver (GPU box): rebuilds libbvc with different histogram-kernel geometry and EM residency caps.
set -e
cd $GRAFT_REPO_ROOT
for cfg in "256 4" "512 2" "256 2"; do
  set -- $cfg
  BVC_EXTRA_FLAGS="-DBVC_HIST_THREADS=$1 -DBVC_HIST_UNROLL=$2" python -c "from basevarc_amd import build as b; b.build(force=True)" > /dev/null 2>&1
  for w in 8 12 16; do
    BVC_EM_WAVES_PER_CU=$w timeout -k 10 200 python bench.py --steps 4 --warmup 1 --cpu-sites 0 --no-verify --no-legs --total-sites 40000 2>/dev/null | python tools/bench_line.py threads $1 unroll $2 cap $w
  done
done
python -c "from basevarc_amd import build as b; b.build(force=True)" > /dev/null 2>&1
