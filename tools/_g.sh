cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch
p=torch.cuda.get_device_properties(0)
print('sharedMemPerBlock', getattr(p,'shared_memory_per_block',None), 'optin', getattr(p,'shared_memory_per_block_optin',None), 'per multiprocessor', getattr(p,'shared_memory_per_multiprocessor',None))
PY
for kib in 0 96 128 160; do
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "group_kernel_variants" 2>&1 | tail -1
BVC_GROUP_LDS_KIB=$kib python bench.py --groups 5 --steps 4 --warmup 1 --cpu-sites 0 --no-verify --no-legs --total-sites 40000 2>/dev/null | python tools/bench_line.py lds $kib overlap interleaved
BVC_GROUP_LDS_KIB=$kib python bench.py --groups 5 --steps 4 --warmup 1 --cpu-sites 0 --no-verify --no-legs --total-sites 40000 --no-overlap 2>/dev/null | python tools/bench_line.py lds $kib serial interleaved
done
