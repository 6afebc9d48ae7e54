#!/bin/bash
# Experiment (GPU box): resolution of the direction map (G cells per cube-face edge) with the tie rule of the climb in place;
# variants built with -DURGYM_DIRMAP_G=<G> into ur_gym_amd/csrc/build/.  Then the whole GPU suite on the shipped library.
set -u
TAG=${1:-exp_climb2}
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
  run g64 $B/lib_g64.so
  run g128 $R/ur_gym_amd/csrc/liburgym_hip.so
  run g192 $B/lib_g192.so
  run g256 $B/lib_g256.so
done
run g128-obs $R/ur_gym_amd/csrc/liburgym_hip.so --env UR5ObsReach-v1 --num-envs 16384
run g256-obs $B/lib_g256.so --env UR5ObsReach-v1 --num-envs 16384
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
