cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "group" 2>&1 | tail -5 || exit 1
for rep in 1 2; do
python bench.py --groups 5 --steps 6 --warmup 1 --cpu-sites 0 --no-verify --no-legs --total-sites 40000 2>/dev/null | python tools/bench_line.py overlap interleaved
python bench.py --groups 5 --steps 6 --warmup 1 --cpu-sites 0 --no-verify --no-legs --total-sites 40000 --no-overlap 2>/dev/null | python tools/bench_line.py serial interleaved
done
