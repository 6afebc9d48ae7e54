#!/bin/bash
# Experiment (GPU box): variants of the hull climb -- the examined vertex's coordinates fetched with its record (no extra round trip
# at the end), and a finer direction map (64 x 64 cells per cube face) -- against the shipped library.  Builds: see profiles/r2/EXPERIMENTS.md.
set -u
TAG=${1:-exp_climb}
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
  run base $R/ur_gym_amd/csrc/liburgym_hip.so
  run eager $B/lib_eager.so
  run g64 $B/lib_g64.so
  run both $B/lib_both.so
done
run base-obs $R/ur_gym_amd/csrc/liburgym_hip.so --env UR5ObsReach-v1 --num-envs 16384
run both-obs $B/lib_both.so --env UR5ObsReach-v1 --num-envs 16384
URGYM_LIB=$B/lib_both.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/parity_both.log 2>&1; tail -2 $OUT/parity_both.log
