#!/bin/bash
# Two-tier launch geometry sweep (URGYM_STEP_TIERS=E1,B,E2); runs ON THE GPU BOX.  usage: tools/exp_tiers.sh <tag>
set -u
TAG=${1:-tiers}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
: > $OUT/tiers.jsonl
one() {
  env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.readline())
print(json.dumps({'cfg': '$*', 'value': b['value'], 'ms_per_step': b['ms_per_step'], 'kernel_us': b['roofline']['kernel_us']}))" >> $OUT/tiers.jsonl
}
one URGYM_STEP_ENVS=46
one URGYM_STEP_ENVS=48
for T in 64,724,27 64,724,24 64,724,32 64,736,26 64,700,28 64,724,16 56,724,35 56,724,28 48,724,43 48,724,32 64,600,40 64,650,36 60,724,31 52,724,39; do
  one URGYM_STEP_TIERS=$T
done
cat $OUT/tiers.jsonl
