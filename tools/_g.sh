cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 1 --cpu-sites 0 --no-verify --total-sites 40000 > gpurun_out/_t.json 2>gpurun_out/_t.err || tail -5 gpurun_out/_t.err
python - <<PY
import json
d=json.loads(open('gpurun_out/_t.json').read().strip().splitlines()[-1])
for k,v in d['legs'].items(): print(k, round(v['value']), 'ms/call', round(v['ms_per_call'],3), 'hist', round(v['roofline'].get('avg_launch_ms',0),3), 'frac', round(v['roofline']['frac'],3), 'stage2', v.get('stage2_ms_per_call'), v.get('records_identical_to_two_byte_path'))
PY
for lay in interleaved ordered; do
python bench.py --packed --groups 5 --group-layout $lay --steps 4 --warmup 1 --total-sites 40000 --no-overlap 2>/dev/null | python tools/bench_line.py packed groups $lay serial
python bench.py --packed --groups 5 --group-layout $lay --steps 4 --warmup 1 --total-sites 40000 2>/dev/null | python tools/bench_line.py packed groups $lay overlap
done
