#!/bin/bash
# Cost of the per-launch HIP timing events on the call time.  usage: bash tools/sweep_profile_every.sh "1 4 1000"
cd $GRAFT_REPO_ROOT
for k in ${1:-1 4 1000}; do
  python bench.py --steps 4 --warmup 1 --cpu-sites 0 --no-verify --no-legs --profile-every $k 2>/dev/null | python tools/bench_line.py profile-every $k
done
