#!/bin/bash
# second-level counters (cache behaviour) for the step kernel; runs ON THE GPU BOX
set -u
TAG=${1:-x}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/$name.log 2>&1; echo "$name exit $?" >> $OUT/$name.log; }
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT
run fetch FETCH_SIZE
run write WRITE_SIZE
# the kernel trace runs the DEFAULT bench command (same steps / warm-up), so that its average kernel duration is comparable with bench.py's own
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline > $OUT/trace.log 2>&1
find $OUT -name "*.csv" -size +20M -delete
