cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz or random_sites or edge or wide_quality or chi_sweep or set_base or min_af or golden or group" 2>&1 | tail -3 || exit 1
for rep in 1 2; do
python bench.py --steps 6 --warmup 1 --cpu-sites 0 --no-verify --no-legs --no-overlap 2>/dev/null | python tools/bench_line.py serial headline
done
python bench.py --steps 6 --warmup 1 --cpu-sites 0 --no-verify > gpurun_out/_t.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/_t.json').read().strip().splitlines()[-1])
print('headline', round(d['value']), {k: round(v,3) for k,v in d['kernels_ms_per_call'].items()}, ' '.join(f"{k.split('_')[0]}:{round(v['value'])}" for k,v in d['legs'].items()))
PY
