#!/bin/bash
set -u
TAG=${1:-fused}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; echo "exit $?" >> $OUT/gpu_tests.log
grep -E "passed|failed|^FAILED" $OUT/gpu_tests.log | tail -8
for i in 1 2; do
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.readline()); print('timed(1/8)', b['value']/1e6, b['ms_per_step'], b['roofline']['kernel_us'], b['roofline']['launches_timed'])"
  python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.readline()); print('untimed', b['value']/1e6, b['ms_per_step'])"
done
python bench.py --no-cpu-baseline --env UR5ObsReach-v1 --num-envs 16384 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.readline()); print('obs', b['value']/1e6, b['ms_per_step'], b['roofline']['kernel_us'])"
python bench.py --no-cpu-baseline --num-envs 262144 --steps 60 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.readline()); print('dyn262k', b['value']/1e6, b['ms_per_step'], b['roofline']['kernel_us'])"
