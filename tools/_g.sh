cd $GRAFT_REPO_ROOT
for shape in "100000 40000 160000" "10000 100000 400000" "10000 40000 160000"; do
  set -- $shape
  for pair in 0 1; do
   for ov in "" "--no-overlap"; do
    BVC_EM_PAIR=$pair python bench.py --samples $1 --tile-sites $2 --total-sites $3 --steps 3 --warmup 1 --cpu-sites 0 --no-verify --no-legs $ov 2>/dev/null | python tools/bench_line.py N $1 tile $2 pair $pair $ov
   done
  done
done
