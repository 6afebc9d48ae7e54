#!/bin/bash
# Experiment (GPU box): draw threshold of the work pool (lanes that must be idle before a wave draws: -DURGYM_REFILL_MIN), builds in build/.
set -u
TAG=${1:-exp_refill}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B=$R/ur_gym_amd/csrc/build
run() { name=$1; lib=$2; shift 2
  URGYM_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline "$@" | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'variant': '$name', 'args': '$*', 'value_M': round(d['value'] / 1e6, 2), 'ms_per_step': round(d['ms_per_step'], 5), 'kernel_us': round(d['roofline']['kernel_us'], 2)}))" | tee -a $OUT/results.jsonl; }
: > $OUT/results.jsonl
for rep in 1 2; do
  run min16 $B/lib_prev.so
  run min12 $B/lib_rm12.so
  run min8 $B/lib_rm8.so
  run min4 $B/lib_rm4.so
done
