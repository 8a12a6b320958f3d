#!/bin/bash
# Round-5 evidence runs on the GPU box: every resident site against the CPU histogram path (plain and both group
# layouts), seeds 1-3, five repeats of the default line.  Output: gpurun_out/r05_evidence/
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05_evidence}; mkdir -p $O   # (one attempt per tag: a second one goes to <tag>_<time>, never over a log)
[ -f $O/seeds.jsonl ] && O=${O}_$(date +%H%M%S) && mkdir -p $O
python bench.py --verify-all --no-legs --cpu-sites 32 > $O/verify_all_config3.json 2> $O/verify_all_config3.err
python bench.py --groups 5 --verify-all --no-legs --cpu-sites 0 --steps 4 > $O/verify_groups_interleaved.json 2> $O/verify_groups_interleaved.err
python bench.py --groups 5 --group-layout ordered --verify-all --no-legs --cpu-sites 0 --steps 4 > $O/verify_groups_ordered.json 2> $O/verify_groups_ordered.err
for s in 1 2 3; do python bench.py --seed $s --cpu-sites 0 --no-legs 2>/dev/null >> $O/seeds.jsonl; done
for r in 1 2 3 4 5; do python bench.py --cpu-sites 0 2>/dev/null >> $O/repeat5.jsonl; done
python tools/verify_packed.py > $O/packed_equals_two_byte_all_sites.json 2> $O/packed_equals_two_byte.err
O=$O python - <<'PY'
import json, os
O = os.environ['O']
for f in ("verify_all_config3","verify_groups_interleaved","verify_groups_ordered"):
    d=json.load(open(f"{O}/{f}.json")); print(f, d["value"], d.get("verify_all"), d.get("cpu_baseline",{}).get("gpu_check_same_sites"))
for l in open(f"{O}/seeds.jsonl"): d=json.loads(l); print("seed", d["config"]["seed"], d["value"], d["roofline"]["frac"])
for l in open(f"{O}/repeat5.jsonl"): d=json.loads(l); print("repeat", d["value"], d["roofline"]["frac"], {k:round(v["value"]) for k,v in d["legs"].items()})
PY
