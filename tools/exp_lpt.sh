#!/bin/bash
# LPT ticket order on/off; runs ON THE GPU BOX.  usage: tools/exp_lpt.sh <tag>
set -u
TAG=${1:-lpt}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; echo "exit $?" >> $OUT/gpu_tests.log
grep -E "passed|failed" $OUT/gpu_tests.log | tail -2
: > $OUT/sweep.jsonl
one() {
  env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.readline())
print(json.dumps({'cfg': '$*', 'value': b['value'], 'ms_per_step': b['ms_per_step'], 'kernel_us': b['roofline']['kernel_us']}))" >> $OUT/sweep.jsonl
}
one URGYM_LPT=1
one URGYM_LPT=0
for E in 80 91 100 112 128; do one URGYM_LPT=1 URGYM_STEP_ENVS=$E; done
one URGYM_LPT=1 URGYM_STEP_ENVS=46
cat $OUT/sweep.jsonl
make -C ur_gym_amd/csrc stamps > /dev/null 2>&1
timeout -k 10 120 python tools/phase_stamps.py --num-envs 65536 --envs-per-block 91 > $OUT/stamps_e91_lpt.txt 2>&1
grep -E "first set-up|GJK loop|barrier wait|loop trips|per loop trip|block lifetime|concurrent|timeline|start->P1" $OUT/stamps_e91_lpt.txt
