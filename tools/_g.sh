cd $GRAFT_REPO_ROOT
for cs in 4000 16000 40000; do
  SECONDS=0; python bench.py --steps 3 --warmup 1 --cpu-sites 0 --no-verify --csr-sites $cs > gpurun_out/_t.json 2>gpurun_out/_t.err; echo "wall $SECONDS s"
  python - <<PY
import json
d=json.loads(open('gpurun_out/_t.json').read().strip().splitlines()[-1])
v=d['legs']['csr_coverage10pct']
print('csr_sites $cs', round(v['value']), 'ms/call', round(v['ms_per_call'],3), 'em frac', round(v['roofline']['frac'],3), 'hist ms', round(v['hist_roofline']['avg_launch_ms'],3), 'hist frac', round(v['hist_roofline']['frac'],3))
PY
done
