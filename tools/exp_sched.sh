#!/bin/bash
# Experiment (GPU box): LLVM AMDGPU scheduler strategies for the whole library (-mllvm -amdgpu-sched-strategy=...), builds in build/.
set -u
TAG=${1:-exp_sched}
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
  run default $B/lib_prev.so
  run max-ilp $B/lib_ilp.so
  run max-memory-clause $B/lib_memc.so
done
run default-obs $B/lib_prev.so --env UR5ObsReach-v1 --num-envs 16384
run max-ilp-obs $B/lib_ilp.so --env UR5ObsReach-v1 --num-envs 16384
