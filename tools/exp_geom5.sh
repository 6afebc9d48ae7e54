#!/bin/bash
# Experiment (GPU box): launch geometries for multi-round sizes (N = 131072, 262144): uniform E against two-tier variants.
set -u
TAG=${1:-exp_geom5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
run() { name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'variant': '$name', 'args': '$*', 'value_M': round(d['value'] / 1e6, 2), 'ms_per_step': round(d['ms_per_step'], 5)}))" | tee -a $OUT/results.jsonl; }
: > $OUT/results.jsonl
for N in 131072 262144; do
  run default --num-envs $N --steps 100
  for E in 64 80 100 112 128; do URGYM_STEP_ENVS=$E run E$E --num-envs $N --steps 100; done
  for T in 128,512,64 128,512,90 110,728,70 100,728,60 120,728,90; do URGYM_STEP_TIERS=$T run tiers-$T --num-envs $N --steps 100; done
done
