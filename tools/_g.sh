cd $GRAFT_REPO_ROOT
for th in 256 1024 512; do
  BVC_EXTRA_FLAGS="-DBVC_HIST_THREADS=$th" python -c "from basevarc_amd import build; build.build(force=True)" || exit 1
  python bench.py --packed --steps 4 --warmup 1 --total-sites 40000 2>/dev/null | python tools/bench_line.py threads $th packed overlap
  python bench.py --packed --steps 4 --warmup 1 --total-sites 40000 --no-overlap 2>/dev/null | python tools/bench_line.py threads $th packed serial
  python bench.py --steps 4 --warmup 1 --total-sites 40000 --cpu-sites 0 --no-verify --no-legs 2>/dev/null | python tools/bench_line.py threads $th two-byte overlap
done
