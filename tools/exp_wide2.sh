#!/bin/bash
# Stamps of the wide step workgroups + REFILL_MIN variants; runs ON THE GPU BOX.  usage: tools/exp_wide2.sh <tag>
set -u
TAG=${1:-wide2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "workbench or geometry or obs_terminal" > $OUT/gpu_tests.log 2>&1; echo "exit $?" >> $OUT/gpu_tests.log
grep -E "passed|failed" $OUT/gpu_tests.log | tail -2
make -C ur_gym_amd/csrc stamps > /dev/null 2>&1
for E in 91 128; do
  timeout -k 10 120 python tools/phase_stamps.py --num-envs 65536 --envs-per-block $E > $OUT/stamps_e$E.txt 2>&1
  grep -E "first set-up|GJK loop|barrier wait|P4  |loop trips|per loop trip|block lifetime|concurrent|timeline" $OUT/stamps_e$E.txt
done
HIPCC=/opt/rocm/bin/hipcc
for RM in 8 16 48; do
  (cd ur_gym_amd/csrc && $HIPCC -O3 -std=c++17 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wno-unused-value -DURGYM_STAMPS -DURGYM_REFILL_MIN=$RM -shared -o build/liburgym_stamps_rm$RM.so urgym_hip.hip 2>/dev/null)
  timeout -k 10 120 python tools/phase_stamps.py --num-envs 65536 --envs-per-block 91 --lib liburgym_stamps_rm$RM.so > $OUT/stamps_e91_rm$RM.txt 2>&1
  echo "== REFILL_MIN $RM"; grep -E "GJK loop|loop trips|per loop trip|block lifetime|concurrent" $OUT/stamps_e91_rm$RM.txt
done
