#!/bin/bash
# Launch-geometry sweep of the step kernel (URGYM_STEP_ENVS) + phase stamps; runs ON THE GPU BOX.  usage: tools/exp_sweep.sh <tag>
set -u
TAG=${1:-sweep}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
: > $OUT/sweep.jsonl
for E in 32 40 44 46 48 52 56 64; do
  URGYM_STEP_ENVS=$E timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.readline())
print(json.dumps({'E': $E, 'value': b['value'], 'ms_per_step': b['ms_per_step'], 'kernel_us': b['roofline']['kernel_us']}))" >> $OUT/sweep.jsonl
done
cat $OUT/sweep.jsonl
make -C ur_gym_amd/csrc stamps > /dev/null 2>&1
for E in 46 48 64; do
  timeout -k 10 120 python tools/phase_stamps.py --num-envs 65536 --envs-per-block $E > $OUT/stamps_e$E.txt 2>&1
done
timeout -k 10 120 python tools/phase_stamps.py --num-envs 262144 --envs-per-block 64 > $OUT/stamps_n262144_e64.txt 2>&1
cat $OUT/stamps_e46.txt
# HBM traffic at E = 48 (128-byte aligned blocks of float64 state) vs the default geometry
pushd /tmp > /dev/null; export TMPDIR=/tmp
for E in 46 48; do
  for C in FETCH_SIZE WRITE_SIZE; do
    URGYM_STEP_ENVS=$E rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_e${E}_$C -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/pmc_e${E}_$C.log 2>&1
  done
  python3 $R/tools/summarize_pmc.py $OUT/pmc_e${E}_FETCH_SIZE/.. 2>/dev/null | head -0
done
popd > /dev/null
python3 - <<PY
import glob, pandas as pd
for E in (46, 48):
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("$OUT/pmc_e%d_%s/*/*counter_collection.csv" % (E, C)):
            df = pd.read_csv(f); df = df[df.Kernel_Name.str.contains("env_kernel<2, 0")]
            print("E", E, C, "KB per launch", df.Counter_Value.mean())
PY
find $OUT -name "*.csv" -size +5M -delete
