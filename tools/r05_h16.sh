#!/bin/bash
# Round 5: the H16 any-order group counters (group_h16 = 1: 32 conflict-free copies of 16-bit counter pairs) against the 16-copy kernels,
# on the bench's own group workloads (k = 5, labels interleaved, N = 1e6; 20,000 sites resident = 5 calls of 4000): sites/s and the
# histogram kernel's HIP-event time per call.  usage (GPU box): bash tools/r05_h16.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/${1:-r05h16}_$(date +%m%d_%H%M%S); mkdir -p $O
for form in "--packed" ""; do
  for h in 0 1; do
    for ov in "" "--no-overlap"; do
      name=$(echo "groups5_${form:---twobyte}_h16_${h}_${ov:-overlap}" | tr -d ' -')
      BVC_GROUP_H16=$h timeout -k 10 300 python bench.py --steps 6 --warmup 2 --total-sites 20000 --groups 5 --group-layout interleaved $form $ov --cpu-sites 0 --no-legs > $O/$name.json 2> $O/$name.err
      echo "$name rc=$?" | tee -a $O/progress
    done
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().split("\n")[-1])
        print(f.split('/')[-1], 'sites/s %.4g' % d['value'], d['roofline']['kernel'], 'ms/call %.4f' % d['roofline']['avg_launch_ms'], 'frac %.3f' % d['roofline']['frac'], d['kernels_ms_per_call'])
    except Exception as e:
        print(f, 'unreadable', e)
PY
