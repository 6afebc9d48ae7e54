#!/bin/bash
# Experiment (GPU box): A/B of the shipped library against build/lib_prev.so on the configs where boolean contact queries matter.
set -u
TAG=${1:-exp_ab_ori}
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
  run prev-ori4k $B/lib_prev.so --env UR5OriReach-v1 --num-envs 4096 --steps 300
  run new-ori4k $R/ur_gym_amd/csrc/liburgym_hip.so --env UR5OriReach-v1 --num-envs 4096 --steps 300
done
run prev-ori64k $B/lib_prev.so --env UR5OriReach-v1 --num-envs 65536
run new-ori64k $R/ur_gym_amd/csrc/liburgym_hip.so --env UR5OriReach-v1 --num-envs 65536
run prev-dyn $B/lib_prev.so
run new-dyn $R/ur_gym_amd/csrc/liburgym_hip.so
run prev-dyn4k $B/lib_prev.so --num-envs 4096
run new-dyn4k $R/ur_gym_amd/csrc/liburgym_hip.so --num-envs 4096
run prev-obs $B/lib_prev.so --env UR5ObsReach-v1 --num-envs 16384
run new-obs $R/ur_gym_amd/csrc/liburgym_hip.so --env UR5ObsReach-v1 --num-envs 16384
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
