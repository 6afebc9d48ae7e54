#!/bin/bash
# Wide step workgroups (E up to 128): parity suite, then a sweep of E; runs ON THE GPU BOX.  usage: tools/exp_wide.sh <tag>
set -u
TAG=${1:-wide}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; echo "exit $?" >> $OUT/gpu_tests.log
grep -E "passed|failed" $OUT/gpu_tests.log | tail -2
: > $OUT/sweep.jsonl
one() {
  env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.readline())
print(json.dumps({'cfg': '$*', 'value': b['value'], 'ms_per_step': b['ms_per_step'], 'kernel_us': b['roofline']['kernel_us']}))" >> $OUT/sweep.jsonl
}
one URGYM_VERBOSE=1
for E in 46 64 80 88 90 92 96 104 112 128; do one URGYM_STEP_ENVS=$E; done
cat $OUT/sweep.jsonl
URGYM_VERBOSE=1 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>&1 | grep urgym
