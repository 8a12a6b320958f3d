cd $GRAFT_REPO_ROOT
for cfg in "1 8" "2 4" "2 6" "2 8" "3 4"; do
  set -- $cfg
  for lay in "" "--group-layout ordered"; do
    BVC_EM_STREAMS=$1 BVC_EM_WAVES_PER_CU=$2 python bench.py --groups 5 $lay --steps 4 --warmup 1 --cpu-sites 0 --no-verify --no-legs --total-sites 40000 2>/dev/null | python tools/bench_line.py streams $1 cap $2 groups $lay
  done
done
for cfg in "1 8" "2 4"; do
  set -- $cfg
  BVC_EM_STREAMS=$1 BVC_EM_WAVES_PER_CU=$2 python bench.py --steps 6 --warmup 1 --cpu-sites 0 --no-verify --no-legs 2>/dev/null | python tools/bench_line.py streams $1 cap $2 headline
done
