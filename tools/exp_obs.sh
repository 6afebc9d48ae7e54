#!/bin/bash
# Obs kernel (EPA phase) after a change: parity suite, bench at N = 16384 for several E, stamps; runs ON THE GPU BOX.
set -u
TAG=${1:-obs}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; echo "exit $?" >> $OUT/gpu_tests.log
grep -E "passed|failed" $OUT/gpu_tests.log | tail -2
: > $OUT/obs.jsonl
one() {
  env "$@" timeout -k 10 120 python bench.py --env UR5ObsReach-v1 --num-envs 16384 --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.readline())
print(json.dumps({'cfg': '$*', 'value': b['value'], 'ms_per_step': b['ms_per_step'], 'kernel_us': b['roofline']['kernel_us']}))" >> $OUT/obs.jsonl
}
one URGYM_VERBOSE=0
for E in 8 12 16 24 32 48; do one URGYM_STEP_ENVS=$E; done
cat $OUT/obs.jsonl
make -C ur_gym_amd/csrc stamps > /dev/null 2>&1
timeout -k 10 120 python tools/phase_stamps.py --env UR5ObsReach-v1 --num-envs 16384 --envs-per-block 32 > $OUT/stamps_obs_n16384_e32.txt 2>&1
grep -E "GJK loop|barrier wait|loop trips|per loop trip|block lifetime|concurrent|timeline" $OUT/stamps_obs_n16384_e32.txt
