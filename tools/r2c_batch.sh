set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
export BVC_BENCH_TMP=/tmp
python tools/host_bench.py 2000 5000 1 0.7 500 > $O/host_2000_t1.jsonl 2>&1
python tools/host_bench.py 100000 1500 1 0.1 5000 > $O/host_1e5_t1.jsonl 2>&1
python tools/host_bench.py 100000 3000 4 0.1 5000 > $O/host_1e5_t4.jsonl 2>&1
python tools/host_bench.py 100000 1000 4 0.7 5000 > $O/host_1e5_t4_cov70.jsonl 2>&1
tail -n 3 $O/host_*.jsonl | cut -c1-400
bash tools/sweep_groups.sh "8 12 16 20 24" > $O/sweep_groups_interleaved.txt 2>&1
bash tools/sweep_groups.sh "8 12 16 20 24" "--group-layout ordered" > $O/sweep_groups_ordered.txt 2>&1
cat $O/sweep_groups_*.txt
