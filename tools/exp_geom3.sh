#!/bin/bash
# Experiment (GPU box): two-tier launch geometries that still fit ONE round of resident workgroups (URGYM_STEP_TIERS=E1,B,E2).
set -u
TAG=${1:-exp_geom3}
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
for T in 96,512,80 100,512,70 104,512,60 108,512,50 112,512,40 120,512,20 100,256,87 110,256,81 128,256,71 100,640,20 96,640,52 94,640,68 100,384,81 110,384,70 128,384,49; do
  URGYM_STEP_TIERS=$T run tiers-$T
done
run default
