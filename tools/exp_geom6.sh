#!/bin/bash
# Experiment (GPU box): one-round geometries again after the verdict bounds of the boolean queries (URGYM_STEP_TIERS / URGYM_STEP_ENVS).
set -u
TAG=${1:-exp_geom6}
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
run default
URGYM_STEP_TIERS=0 run uniform
for T in 94,512,84 96,512,79 98,512,73 100,512,68 102,512,63 104,512,58 108,512,48 96,640,47 100,384,79 104,384,75; do
  URGYM_STEP_TIERS=$T run tiers-$T
done
run default
