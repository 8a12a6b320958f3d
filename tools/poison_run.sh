#!/bin/bash
# Diagnostic run on the GPU box, ONCE per change (never in a loop): libbvc built with -DBVC_POISON -DBVC_CHECK_LDS
# (bvc_device.h: every kernel fills its whole LDS allocation with 0xFF first, every device scratch buffer of a context is
# filled with 0xFF, every data-derived LDS address / index is bound-checked and a violation RECORDED, not trapped), then the
# GPU parity suite and one short pass of every bench leg under it.  A read-before-write of LDS or scratch shows as a parity
# failure (the poison reaches a record) or as a recorded violation (tests/conftest.py fails the test with the record).
# The product library is rebuilt afterwards.  usage: bash tools/poison_run.sh <tag>
# Every attempt writes to a directory of its own (gpurun_out/<tag>_<time>): a failing log is never overwritten.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-poison}_$(date +%m%d_%H%M%S)
mkdir -p $O
cd $R
# exported for the whole run: build.needs_build() compares the flags a library was built with (libbvc.so.flags) with this variable,
# so the tests' and the bench's own build() calls keep the diagnostic library instead of rebuilding the product
export BVC_EXTRA_FLAGS="-DBVC_POISON -DBVC_CHECK_LDS"
python -c "from basevarc_amd import build; build.build(force=True)" > $O/build.log 2>&1 || { echo "poison build failed" | tee $O/verdict; exit 1; }
nm -D basevarc_amd/libbvc.so | grep -c bvc_debug_report > $O/has_debug_export
rc=0
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_host.py > $O/pytest_gpu.log 2>&1 || rc=$?
tail -3 $O/pytest_gpu.log
echo "pytest rc=$rc" > $O/verdict
if [ $rc = 0 ]; then
  # the bench's phases (headline warm-up included) under the same build; short, once
  timeout -k 10 600 python bench.py --steps 2 --warmup 1 --cpu-sites 0 > $O/bench.json 2> $O/bench.err || rc=$?
  echo "bench rc=$rc" >> $O/verdict
  python -c "import json,sys; d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); print('bench lds_violations:', d.get('diagnostic_build'))" >> $O/verdict 2>&1
fi
grep -il "aperture\|exception\|violation" $O/*.err $O/*.log 2>/dev/null | sed 's/^/mentions a fault: /' >> $O/verdict
unset BVC_EXTRA_FLAGS
python -c "from basevarc_amd import build; build.build(force=True)" > $O/rebuild.log 2>&1 && echo "product library rebuilt" >> $O/verdict
cat $O/verdict
exit $rc
