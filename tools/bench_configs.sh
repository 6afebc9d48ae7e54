#!/bin/bash
# BASELINE.json configs[1..3] + batch-size scaling of the headline config; runs ON THE GPU BOX, one JSON line each.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/bench_configs.jsonl
: > $OUT
run() { timeout -k 10 300 python $R/bench.py --no-cpu-baseline "$@" | grep '^{' >> $OUT; }
run --env UR5OriReach-v1 --num-envs 4096 --steps 300 --warmup 20
run --env UR5OriReach-v1 --num-envs 4096 --steps 300 --warmup 20 --rollout
run --env UR5OriReach-v1 --num-envs 4096 --steps 300 --warmup 20 --rollout --no-collision
run --env UR5OriReach-v1 --num-envs 65536 --steps 300 --warmup 20 --rollout --no-collision
run --env UR5ObsReach-v1 --num-envs 16384 --steps 200 --warmup 20
run --env UR5DynReach-v1 --num-envs 65536 --steps 200 --warmup 20
run --env UR5DynReach-v1 --num-envs 65536 --steps 200 --warmup 20 --rollout
run --env UR5DynReach-v1 --num-envs 262144 --steps 60 --warmup 10
run --env UR5StaReach-v1 --num-envs 65536 --steps 100 --warmup 10
cat $OUT
