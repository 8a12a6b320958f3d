#!/bin/bash
# Round 5: the driver's multi-GPU command line (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
# --master-port P bench.py --gpus N --steps K --warmup W) rehearsed with N = 5 ranks sharing this box's ONE card (BVC_BENCH_BACKEND=gloo:
# the ranks rendezvous over gloo and all use device 0).  The pool's process guard lets six processes use a card at once and the launcher
# counts as one (a run with six ranks was killed by it: "7 processes had the GPU open"), so five ranks it is, and the driver's
# N = 8 line itself cannot be rehearsed here; what this run covers is everything that depends on N only through WORLD_SIZE: rendezvous,
# shard_range / call_sizes for a world that does not divide the workload (50,003 sites over 5: 10,001 / 10,000 sites a rank), the
# barrier, the max-over-ranks clock, the per_rank array.  NOT a scaling measurement: the ranks share one GPU.
# usage (GPU box): bash tools/r05_rehearsal.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/${1:-r05_rehearsal}; mkdir -p $O
export BVC_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 5 --steps 3 --warmup 1 --total-sites 50003 \
    > $O/bench_5_ranks_one_card.json 2> $O/bench_5_ranks_one_card.err
echo "rc=$?" | tee $O/verdict
python - <<PY
import json
l=[x for x in open("$O/bench_5_ranks_one_card.json") if x.strip().startswith("{")]
d=json.loads(l[-1])
print("n_gpus", d["n_gpus"], "scaling", d["scaling"], "sites_per_step", d["config"]["sites_per_step"], "sharding:", d["config"]["sharding"])
print("per_rank:", [(p["rank"], p["sites"], p["calls_per_step"], round(p["ms_per_step"],1)) for p in d["per_rank"]])
assert d["n_gpus"] == 5 and len(d["per_rank"]) == 5 and sum(p["sites"] for p in d["per_rank"]) == d["config"]["sites_per_step"] == 50003
print("value (five ranks on ONE card, not a scaling figure):", d["value"])
PY
