cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pack" 2>&1 | tail -3 || exit 1
show() {
python - "$1" <<PY
import json,sys
d=json.loads(open('gpurun_out/_t.json').read().strip().splitlines()[-1])
v=d['legs']['packed_1_byte_per_sample']
print(sys.argv[1], 'packed', round(v['value']), 'ms/call', round(v['ms_per_call'],3), 'hist', round(v['roofline']['avg_launch_ms'],3), 'frac', round(v['roofline']['frac'],3), 'em', round(v['stage2_ms_per_call'],3), 'same', v['records_identical_to_two_byte_path'], '| headline', round(d['value']))
PY
}
for cfg in "0 0" "8 1" "12 1" "6 2" "8 2" "12 2" "8 3"; do
  set -- $cfg
  BVC_EM_WAVES_PER_CU=$1 BVC_EM_STREAMS=$2 python bench.py --steps 3 --warmup 1 --cpu-sites 0 --no-verify --total-sites 40000 > gpurun_out/_t.json 2>gpurun_out/_t.err || tail -5 gpurun_out/_t.err
  show "waves $1 streams $2"
done
python bench.py --steps 3 --warmup 1 --cpu-sites 0 --no-verify --total-sites 40000 --no-overlap > gpurun_out/_t.json 2>gpurun_out/_t.err
show "serial"
