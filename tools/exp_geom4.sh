#!/bin/bash
# Experiment (GPU box): the default two-tier one-round geometry against uniform workgroups (URGYM_STEP_TIERS=0) over N and envs.
set -u
TAG=${1:-exp_geom4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
run() { name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'variant': '$name', 'args': '$*', 'value_M': round(d['value'] / 1e6, 2), 'ms_per_step': round(d['ms_per_step'], 5), 'kernel_us': round(d['roofline']['kernel_us'], 2)}))" | tee -a $OUT/results.jsonl; }
: > $OUT/results.jsonl
for N in 16384 32768 49152 65536 81920; do
  run tiers --num-envs $N
  URGYM_STEP_TIERS=0 run uniform --num-envs $N
done
for ENV in UR5ObsReach-v1 UR5StaReach-v1; do
  for N in 16384 65536; do
    run tiers --env $ENV --num-envs $N
    URGYM_STEP_TIERS=0 run uniform --env $ENV --num-envs $N
  done
done
URGYM_VERBOSE=1 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>&1 | grep urgym | head -3
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
