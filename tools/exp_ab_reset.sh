#!/bin/bash
# Experiment (GPU box): A/B of the shipped library against build/lib_prev.so: default bench, full-reset time, GPU suite.
set -u
TAG=${1:-exp_ab_reset}
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
  run prev $B/lib_prev.so
  run new $R/ur_gym_amd/csrc/liburgym_hip.so
done
run prev-obs $B/lib_prev.so --env UR5ObsReach-v1 --num-envs 16384
run new-obs $R/ur_gym_amd/csrc/liburgym_hip.so --env UR5ObsReach-v1 --num-envs 16384
for lib in $B/lib_prev.so $R/ur_gym_amd/csrc/liburgym_hip.so; do
URGYM_LIB=$lib python - <<'PY'
import os, time, torch
from ur_gym_amd import make_vec
for env_id in ("UR5DynReach-v1", "UR5ObsReach-v1"):
    env = make_vec(env_id, num_envs=65536, seed=5); env.reset(seed=5); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): env.reset(seed=5)
    torch.cuda.synchronize()
    print(os.path.basename(os.environ["URGYM_LIB"]), env_id, "full reset of 65536 envs: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
    env.close()
PY
done
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
