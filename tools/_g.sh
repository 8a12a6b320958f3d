cd $GRAFT_REPO_ROOT
run() {
  for k in 5 16 2; do
  python bench.py --groups $k --group-layout ordered --steps 8 --warmup 1 --cpu-sites 0 --no-verify --no-legs --total-sites 40000 --no-overlap --profile-every 1 2>/dev/null | python tools/bench_line.py "$1" serial ordered k $k
  done
  python bench.py --groups 5 --group-layout ordered --steps 8 --warmup 1 --cpu-sites 0 --no-verify --no-legs --total-sites 40000 2>/dev/null | python tools/bench_line.py "$1" overlap ordered k 5
}
for rep in 1 2; do
  BVC_EXTRA_FLAGS="-DBVC_RANGES_SWIZZLE" python -c "from basevarc_amd import build; build.build(force=True)" && run swizzle
  python -c "from basevarc_amd import build; build.build(force=True)" && run plain
done
