#!/bin/bash
# Round 3: one full GPU-box session -- parity suite, default bench, all BASELINE configs, the two-rank rehearsal of the overlapped
# gather, lane tables of the stamps build.  (profiles: tools/r3_prof.sh, a session of its own)
set -u
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -rA > $OUT/gpu_tests.log 2>&1; rc=$?; echo "exit $rc" >> $OUT/gpu_tests.log
grep -E "passed|failed" $OUT/gpu_tests.log | tail -2
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
head -c 300 $OUT/bench_default.json; echo
bash tools/gpu_round.sh $TAG configs || exit 1
URGYM_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --num-envs 16384 --steps 20 --warmup 5 --gather-obs > $OUT/rehearse2.json 2> $OUT/rehearse2.err; echo "rehearse exit $?"
head -c 600 $OUT/rehearse2.json; echo
bash tools/r3_stamps.sh $TAG
