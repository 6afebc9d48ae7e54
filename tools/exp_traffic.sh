#!/bin/bash
# HBM-side traffic (FETCH_SIZE x 2, WRITE_SIZE) and speed of geometry / set-up-cache variants; runs ON THE GPU BOX.
set -u
TAG=${1:-traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; echo "exit $?" >> $OUT/gpu_tests.log
grep -E "passed|failed" $OUT/gpu_tests.log | tail -2
: > $OUT/speed.jsonl
one() {
  env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.readline())
print(json.dumps({'cfg': '$*', 'value': b['value'], 'ms_per_step': b['ms_per_step'], 'kernel_us': b['roofline']['kernel_us']}))" >> $OUT/speed.jsonl
}
pushd /tmp > /dev/null; export TMPDIR=/tmp
for V in "URGYM_SETUP_CACHE=1" "URGYM_SETUP_CACHE=0" "URGYM_STEP_ENVS=96 URGYM_SETUP_CACHE=1" "URGYM_STEP_ENVS=96 URGYM_SETUP_CACHE=0"; do
  tag=$(echo $V | tr ' =' '__')
  (cd $R; one $V)
  for C in FETCH_SIZE WRITE_SIZE; do
    env $V rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${tag}_$C -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/pmc_${tag}_$C.log 2>&1
  done
done
popd > /dev/null
cat $OUT/speed.jsonl
python3 - <<PY
import glob, pandas as pd
for d in sorted(glob.glob("$OUT/pmc_*")):
    if not d.endswith(("FETCH_SIZE", "WRITE_SIZE")): continue
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        df = pd.read_csv(f); df = df[df.Kernel_Name.str.contains("env_kernel<2, 0")]
        mult = 2 if d.endswith("FETCH_SIZE") else 1
        print(d.split("pmc_")[1], "MB per launch", round(df.Counter_Value.mean() * 1024 * mult / 1e6, 1))
PY
find $OUT -name "*.csv" -size +5M -delete
