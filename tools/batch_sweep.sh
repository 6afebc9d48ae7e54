#!/bin/bash
# Throughput vs batch size of the headline env on one GPU; runs ON THE GPU BOX, one JSON line per size.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/batch_sweep.jsonl
: > $OUT
for n in 1024 4096 16384 65536 131072 262144 524288 1048576; do
  steps=200; [ $n -ge 262144 ] && steps=60; [ $n -ge 1048576 ] && steps=30
  timeout -k 10 300 python $R/bench.py --env UR5DynReach-v1 --num-envs $n --steps $steps --warmup 10 --no-cpu-baseline | grep '^{' >> $OUT
done
cat $OUT
