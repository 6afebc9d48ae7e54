#!/bin/bash
# Experiment (GPU box): launch geometry and draw threshold again, after the cheaper hull climb / set-up (r2).  usage: tools/exp_geom2.sh <tag>
set -u
TAG=${1:-exp_geom2}
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
L=$R/ur_gym_amd/csrc/liburgym_hip.so
run default $L
for E in 46 64 80 100 112 128; do URGYM_STEP_ENVS=$E run E$E $L; done
URGYM_STEP_TIERS=100,512,70 run tiers-100-512-70 $L
URGYM_STEP_TIERS=80,600,128 run tiers-80-600-128 $L
run refill8 $B/lib_rm8.so
run refill24 $B/lib_rm24.so
run default $L
