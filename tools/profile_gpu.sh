#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + PMC passes of bench.py; outputs under gpurun_out/.
# usage: tools/profile_gpu.sh <tag> [extra bench args]
set -u
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
echo "trace exit $?" >> $OUT/trace.log
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $OUT/pmc_sq.log 2>&1
echo "pmc_sq exit $?" >> $OUT/pmc_sq.log
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $OUT/pmc_sq2.log 2>&1
echo "pmc_sq2 exit $?" >> $OUT/pmc_sq2.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1
echo "fetch exit $?" >> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1
echo "write exit $?" >> $OUT/pmc_write.log
# keep only the small summaries (<64 MiB merge limit)
find $OUT -name "*.csv" -size +20M -delete
ls -laR $OUT | head -60
