cd $GRAFT_REPO_ROOT
for lay in interleaved ordered; do
for cfg in "0 0" "6 2" "8 2" "12 2" "6 3" "8 3" "16 1"; do
  set -- $cfg
  BVC_EM_WAVES_PER_CU=$1 BVC_EM_STREAMS=$2 python bench.py --packed --groups 5 --group-layout $lay --steps 4 --warmup 1 --total-sites 40000 2>/dev/null | python tools/bench_line.py $lay waves $1 streams $2
done
done
