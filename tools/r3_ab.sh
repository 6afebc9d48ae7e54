#!/bin/bash
# Round 3 A/B session ON THE GPU BOX (through gpurun): the parity suite on the variant library, then interleaved bench runs of
# base (in-tree liburgym_hip.so) and the variant(s).  The variant's source is committed on its experiment branch BEFORE this runs.
# usage: tools/r3_ab.sh <tag> <variant-lib-under-ur_gym_amd/csrc/build> [more variant libs ...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
V=""
for lib in "$@"; do
  name=${lib%.so}; name=${name#liburgym_}
  URGYM_LIB=$R/ur_gym_amd/csrc/build/$lib timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/tests_$name.log 2>&1
  rc=$?; echo "tests $name exit $rc" | tee -a $OUT/tests_$name.log; tail -3 $OUT/tests_$name.log
  [ $rc -ne 0 ] && exit $rc
  V="$V --variant $name:ur_gym_amd/csrc/build/$lib:"
done
timeout -k 10 900 python tools/ab.py --out $OUT/dyn.jsonl --reps 2 --variant base:: $V && \
timeout -k 10 600 python tools/ab.py --out $OUT/obs.jsonl --reps 2 --variant base:: $V -- --env UR5ObsReach-v1 --num-envs 16384 && \
timeout -k 10 600 python tools/ab.py --out $OUT/ori.jsonl --reps 1 --variant base:: $V -- --env UR5OriReach-v1 --num-envs 4096 --steps 300
