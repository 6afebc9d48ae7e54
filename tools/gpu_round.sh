#!/bin/bash
# One GPU-box session of a round: parity suite, default bench, the N = 2 rehearsal of the multi-rank path from the plain command
# line, kernel trace + PMC passes of the default bench command.  Runs ON THE GPU BOX through gpurun; outputs under gpurun_out/<tag>/.
# usage: tools/gpu_round.sh <tag> [tests|bench|prof ...]   (default: all)
set -u
TAG=${1:-r2}; shift || true
WHAT=${*:-tests bench rehearse prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for w in $WHAT; do
  case $w in
    tests)
      timeout -k 10 900 python -m pytest tests -m gpu -q -rA > $OUT/gpu_tests.log 2>&1; echo "exit $?" >> $OUT/gpu_tests.log
      grep -E "passed|failed" $OUT/gpu_tests.log | tail -2 ;;
    bench)
      timeout -k 10 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"
      head -c 600 $OUT/bench_default.json; echo ;;
    rehearse)
      URGYM_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --num-envs 16384 --steps 20 --warmup 5 --gather-obs > $OUT/rehearse2.json 2> $OUT/rehearse2.err; echo "rehearse exit $?"
      head -c 400 $OUT/rehearse2.json; echo ;;
    configs)
      : > $OUT/bench_configs.jsonl
      run() { timeout -k 10 300 python $R/bench.py --no-cpu-baseline "$@" | grep '^{' >> $OUT/bench_configs.jsonl; }
      run --env UR5OriReach-v1 --num-envs 4096 --steps 300 --warmup 20
      run --env UR5OriReach-v1 --num-envs 4096 --steps 300 --warmup 20 --rollout --no-collision
      run --env UR5ObsReach-v1 --num-envs 16384 --steps 200 --warmup 20
      run --env UR5DynReach-v1 --num-envs 65536 --steps 200 --warmup 20
      run --env UR5DynReach-v1 --num-envs 262144 --steps 60 --warmup 10
      run --env UR5StaReach-v1 --num-envs 65536 --steps 100 --warmup 10
      cut -c1-200 $OUT/bench_configs.jsonl ;;
    prof)
      P=$OUT/prof; mkdir -p $P
      pushd /tmp > /dev/null; export TMPDIR=/tmp
      rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 $R/bench.py --no-cpu-baseline > $P/trace.log 2>&1; echo "trace exit $?"
      pmc() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $P/$name -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $P/$name.log 2>&1; echo "$name exit $?"; }
      pmc sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
      pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT
      pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
      pmc fetch FETCH_SIZE
      pmc write WRITE_SIZE
      popd > /dev/null
      find $P -name "*.csv" -size +20M -delete
      python3 $R/tools/summarize_pmc.py $P > $OUT/pmc_summary.json 2> $OUT/pmc_summary.err; head -c 300 $OUT/pmc_summary.json; echo
      cp $(find $P/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null ;;
  esac
done
