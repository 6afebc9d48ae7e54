#!/bin/bash
# Set-up cache + REFILL_MIN variants of the wide step workgroups; runs ON THE GPU BOX.  usage: tools/exp_wide3.sh <tag>
set -u
TAG=${1:-wide3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; echo "exit $?" >> $OUT/gpu_tests.log
grep -E "passed|failed" $OUT/gpu_tests.log | tail -2
HIPCC=/opt/rocm/bin/hipcc
for RM in 4 8 16 24; do
  (cd ur_gym_amd/csrc && $HIPCC -O3 -std=c++17 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wno-unused-value -DURGYM_REFILL_MIN=$RM -shared -o build/liburgym_rm$RM.so urgym_hip.hip 2>/dev/null) &
done
wait
: > $OUT/sweep.jsonl
one() {
  env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.readline())
print(json.dumps({'cfg': '$*', 'value': b['value'], 'ms_per_step': b['ms_per_step'], 'kernel_us': b['roofline']['kernel_us']}))" >> $OUT/sweep.jsonl
}
for E in 46 91 96 128; do one URGYM_STEP_ENVS=$E; done
for RM in 4 8 16 24; do
  for E in 46 91 128; do one URGYM_LIB=$R/ur_gym_amd/csrc/build/liburgym_rm$RM.so URGYM_STEP_ENVS=$E; done
done
cat $OUT/sweep.jsonl
