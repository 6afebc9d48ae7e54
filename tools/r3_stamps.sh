#!/bin/bash
# Round 3: lane / section tables of the stamps build for the BASELINE configs (runs ON THE GPU BOX through gpurun).
# usage: tools/r3_stamps.sh <tag>
set -u
TAG=${1:-r3_stamps}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 200 python tools/phase_stamps.py --env UR5DynReach-v1 --num-envs 65536 --tiers 99,512,69 > $OUT/lane_table_dyn_n65536.txt 2>&1 && \
timeout -k 10 200 python tools/phase_stamps.py --env UR5ObsReach-v1 --num-envs 16384 --envs-per-block 24 > $OUT/lane_table_obs_n16384.txt 2>&1 && \
timeout -k 10 200 python tools/phase_stamps.py --env UR5DynReach-v1 --num-envs 728 --envs-per-block 91 > $OUT/lane_table_dyn_n728_e91.txt 2>&1
echo "stamps exit $?"
