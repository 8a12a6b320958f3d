# Experiment driver (GPU box): rebuilds libbvc with different histogram-kernel geometry and EM residency caps.
set -e
cd $GRAFT_REPO_ROOT
for cfg in "256 4" "512 2" "256 2"; do
  set -- $cfg
  BVC_EXTRA_FLAGS="-DBVC_HIST_THREADS=$1 -DBVC_HIST_UNROLL=$2" python -c "from basevarc_amd import build as b; b.build(force=True)" > /dev/null 2>&1
  for w in 8 12 16; do
  BVC_EM_WAVES_PER_CU=$w timeout -k 10 200 python bench.py --cpu-sites 0 --no-verify --steps 100 --total-sites 40000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"threads $1 unroll $2 cap $w overlap\", round(d[\"value\"]), round(d[\"roofline\"][\"frac\"],4), d[\"kernels_ms_per_step\"])"
  done
done
