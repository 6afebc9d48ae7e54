#!/bin/bash
# L2 <-> fabric traffic (FETCH_SIZE x 2, WRITE_SIZE) and speed of the set-up-cache levels (URGYM_SETUP_CACHE = 0 / 1 / 2) and of the
# uniform geometry; runs ON THE GPU BOX.  usage: tools/exp_traffic.sh <tag>
set -u
TAG=${1:-traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
: > $OUT/speed.jsonl
one() {
  env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.readline())
print(json.dumps({'cfg': '$*', 'value_M': round(b['value'] / 1e6, 2), 'ms_per_step': round(b['ms_per_step'], 5), 'kernel_us': round(b['roofline']['kernel_us'], 2)}))" >> $OUT/speed.jsonl
}
pushd /tmp > /dev/null; export TMPDIR=/tmp
for V in "URGYM_SETUP_CACHE=2" "URGYM_SETUP_CACHE=1" "URGYM_SETUP_CACHE=0" "URGYM_SETUP_CACHE=2 URGYM_STEP_TIERS=0"; do
  tag=$(echo $V | tr ' =' '__')
  (cd $R; one $V)
  for C in FETCH_SIZE WRITE_SIZE; do
    env $V rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${tag}_$C -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/pmc_${tag}_$C.log 2>&1
  done
done
popd > /dev/null
cat $OUT/speed.jsonl
python3 - <<PY > $OUT/traffic.txt
import glob, pandas as pd
for d in sorted(glob.glob("$OUT/pmc_*")):
    if not d.endswith(("FETCH_SIZE", "WRITE_SIZE")): continue
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        df = pd.read_csv(f); df = df[df.Kernel_Name.str.contains("env_step_fused<2")]
        mult = 2 if d.endswith("FETCH_SIZE") else 1   # gfx950: FETCH_SIZE counts 64-byte units as if they were 32 (MI355X_MICROARCH.md)
        print(d.split("pmc_")[1], "MB per launch", round(df.Counter_Value.mean() * 1024 * mult / 1e6, 1))
PY
cat $OUT/traffic.txt
find $OUT -name "*.csv" -size +5M -delete
