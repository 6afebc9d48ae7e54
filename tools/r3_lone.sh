#!/bin/bash
# Where the cycles of a LONE wave go (one wave per SIMD: 8 workgroups of 91 envs on the whole chip): issue vs waits, by counter.
set -u
TAG=${1:-r3_lone}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export URGYM_STEP_ENVS=91
P=$OUT/lone; mkdir -p $P
lone() { local name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $P/$name -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --num-envs 728 > $P/$name.log 2>&1 || echo "lone $name failed"; }
lone a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
lone b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
lone c SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_IFETCH
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --num-envs 728 > $P/trace.log 2>&1
python3 $R/tools/summarize_pmc.py "UR5DynReach-v1 N=728 lone wave per SIMD::$P" > $OUT/lone_summary.json 2> $OUT/lone_summary.err
head -c 3000 $OUT/lone_summary.json
rm -rf $P
