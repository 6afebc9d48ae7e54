#!/bin/bash
# -ffp-contract=fast on the device (vdot3 kept exactly rounded): parity suite + speed; runs ON THE GPU BOX.
set -u
TAG=${1:-contract}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
(cd ur_gym_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=fast -fPIC --offload-arch=gfx950 -Wno-unused-value -shared -o build/liburgym_fc_fast.so urgym_hip.hip 2>/dev/null)
URGYM_LIB=$R/ur_gym_amd/csrc/build/liburgym_fc_fast.so timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; echo "exit $?" >> $OUT/gpu_tests.log
grep -E "passed|failed|^FAILED" $OUT/gpu_tests.log | tail -8
: > $OUT/speed.jsonl
one() {
  env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.readline())
print(json.dumps({'cfg': '$*', 'value': b['value'], 'ms_per_step': b['ms_per_step'], 'kernel_us': b['roofline']['kernel_us']}))" >> $OUT/speed.jsonl
}
one URGYM_VERBOSE=0
one URGYM_LIB=$R/ur_gym_amd/csrc/build/liburgym_fc_fast.so
one URGYM_VERBOSE=0
one URGYM_LIB=$R/ur_gym_amd/csrc/build/liburgym_fc_fast.so
cat $OUT/speed.jsonl
