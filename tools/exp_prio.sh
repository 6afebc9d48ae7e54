#!/bin/bash
set -u
TAG=${1:-prio}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; echo "exit $?" >> $OUT/gpu_tests.log
grep -E "passed|failed" $OUT/gpu_tests.log | tail -2
for S in 2 3 4; do
  (cd ur_gym_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wno-unused-value -DURGYM_STRAGGLER_PRIO=$S -shared -o build/liburgym_prio$S.so urgym_hip.hip 2> build/prio$S.err) &
done
wait
ls -la ur_gym_amd/csrc/build/*prio*
: > $OUT/prio.jsonl
one() {
  env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.readline())
print(json.dumps({'cfg': '$*', 'value': b['value'], 'ms_per_step': b['ms_per_step'], 'kernel_us': b['roofline']['kernel_us']}))" >> $OUT/prio.jsonl
}
one URGYM_VERBOSE=0
for S in 2 3 4; do one URGYM_LIB=$R/ur_gym_amd/csrc/build/liburgym_prio$S.so; done
one URGYM_VERBOSE=0
: > $OUT/ori.jsonl
timeout -k 10 120 python bench.py --env UR5OriReach-v1 --num-envs 4096 --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c1-160 >> $OUT/ori.jsonl
timeout -k 10 120 python bench.py --env UR5OriReach-v1 --num-envs 4096 --steps 300 --warmup 20 --no-cpu-baseline --rollout --no-collision 2>/dev/null | cut -c1-160 >> $OUT/ori.jsonl
cat $OUT/prio.jsonl $OUT/ori.jsonl
