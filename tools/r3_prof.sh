#!/bin/bash
# Round 3 profile session ON THE GPU BOX (through gpurun): kernel trace + separate --pmc passes of the bench command of every BASELINE
# configuration, condensed into ONE summary keyed by configuration and real kernel name (tools/summarize_pmc.py).
# usage: tools/r3_prof.sh <tag>           -> gpurun_out/<tag>/{pmc_summary.json, kernel_stats_<cfg>.csv, bench_<cfg>.json}
set -u
TAG=${1:-r3_prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
SPECS=()
prof_cfg() {  # <short> <label> <bench args...>
  local short=$1 label=$2; shift 2
  local P=$OUT/$short; mkdir -p $P
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 $R/bench.py --no-cpu-baseline "$@" > $P/trace.log 2>&1 || { echo "$short trace failed"; return 1; }
  grep '^{' $P/trace.log | tail -1 > $OUT/bench_$short.json
  pmc() { local name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $P/$name -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline "${BARGS[@]}" > $P/$name.log 2>&1 || { echo "$short $name failed"; return 1; }; }
  BARGS=("$@")
  pmc sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY || return 1
  pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT || return 1
  pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum || return 1
  pmc fetch FETCH_SIZE || return 1
  pmc write WRITE_SIZE || return 1
  find $P -name "*.csv" -size +20M -delete
  cp $(find $P/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$short.csv 2>/dev/null
  SPECS+=("$label::$P")
  echo "$short done"
}
rocprofv3 -L > $OUT/counters_avail.txt 2>&1 || true
prof_cfg dyn65536 "UR5DynReach-v1 N=65536" && \
prof_cfg obs16384 "UR5ObsReach-v1 N=16384" --env UR5ObsReach-v1 --num-envs 16384 && \
prof_cfg ori4096 "UR5OriReach-v1 N=4096" --env UR5OriReach-v1 --num-envs 4096 --steps 300 && \
prof_cfg ori4096nc "UR5OriReach-v1 N=4096 no-collision rollout" --env UR5OriReach-v1 --num-envs 4096 --steps 300 --rollout --no-collision
rc=$?
# one wave per SIMD (8 workgroups of 91 envs on the whole chip): where the cycles of a LONE wave go -- issue vs waits
if [ $rc -eq 0 ]; then
  export URGYM_STEP_ENVS=91
  P=$OUT/lone; mkdir -p $P
  lone() { local name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $P/$name -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --num-envs 728 > $P/$name.log 2>&1 || echo "lone $name failed"; }
  lone a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
  lone b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
  lone c SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_IFETCH
  unset URGYM_STEP_ENVS
  SPECS+=("UR5DynReach-v1 N=728 lone wave per SIMD::$P")
fi
python3 $R/tools/summarize_pmc.py --elf $R/ur_gym_amd/csrc/build/resource_usage.txt "${SPECS[@]}" > $OUT/pmc_summary.json 2> $OUT/pmc_summary.err
head -c 400 $OUT/pmc_summary.json; echo
# raw per-pass directories are large: keep the condensed files only (once the summary really holds kernels)
grep -q "env_step_fused" $OUT/pmc_summary.json && rm -rf $OUT/dyn65536 $OUT/obs16384 $OUT/ori4096 $OUT/ori4096nc $OUT/lone
exit $rc
