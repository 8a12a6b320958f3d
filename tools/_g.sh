cd $GRAFT_REPO_ROOT
for prio in 3 0 1; do
  BVC_EXTRA_FLAGS="-DBVC_PACKED_PRIO=$prio" python -c "from basevarc_amd import build; build.build(force=True)" || exit 1
  for cfg in "0 0" "12 2" "16 1"; do
    set -- $cfg
    BVC_EM_WAVES_PER_CU=$1 BVC_EM_STREAMS=$2 python bench.py --packed --steps 4 --warmup 1 --total-sites 40000 2>/dev/null | python tools/bench_line.py prio $prio waves $1 streams $2
  done
done
