#!/bin/bash
# Batch-size sweep of the headline env + stamps of the Obs kernel (EPA phase); runs ON THE GPU BOX.  usage: tools/gpu_final.sh <tag>
set -u
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
: > $OUT/batch_sweep.jsonl
for n in 1024 4096 16384 65536 131072 262144 524288 1048576; do
  steps=200; [ $n -ge 262144 ] && steps=60; [ $n -ge 1048576 ] && steps=30
  timeout -k 10 300 python bench.py --env UR5DynReach-v1 --num-envs $n --steps $steps --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/batch_sweep.jsonl
done
python - <<PY
import json
for l in open("$OUT/batch_sweep.jsonl"):
    d = json.loads(l); print(d["config"]["envs_total"], round(d["value"] / 1e6, 1), round(d["ms_per_step"], 3))
PY
make -C ur_gym_amd/csrc stamps > /dev/null 2>&1
timeout -k 10 120 python tools/phase_stamps.py --env UR5ObsReach-v1 --num-envs 16384 --envs-per-block 32 > $OUT/stamps_obs_n16384_e32.txt 2>&1
grep -E "first set-up|GJK loop|barrier wait|P4  |loop trips|per loop trip|block lifetime|concurrent|timeline" $OUT/stamps_obs_n16384_e32.txt
timeout -k 10 120 python tools/phase_stamps.py --num-envs 65536 --envs-per-block 91 > $OUT/stamps_dyn_n65536_e91.txt 2>&1
timeout -k 10 120 python tools/phase_stamps.py --num-envs 65536 --tiers 100,512,70 > $OUT/stamps_dyn_n65536_tiers.txt 2>&1
timeout -k 10 120 python tools/phase_stamps.py --num-envs 728 --envs-per-block 91 > $OUT/stamps_dyn_n728_e91.txt 2>&1
grep -E "first set-up|GJK loop|barrier wait|P4  |loop trips|per loop trip|block lifetime|concurrent|timeline|%" $OUT/stamps_dyn_n65536_tiers.txt
