#!/bin/bash
# Collect rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   tools/profile_pmc.sh <tag> [bench.py args...]
# Writes text summaries gpurun_out/<tag>_kernel_stats.txt and gpurun_out/<tag>_pmc.txt.
# Every rocprofv3 call sits under `timeout`: a TA_* counter set once aborted the profiler and hung
# the box for the whole gpurun limit.  Counter passes are separate runs with --kernel-trace only (never combined with other traces).
set -u
TAG=$1; shift
ARGS="--steps 20 --warmup 3 --passes-per-step 1 --repeats 1 --no-cpu-baseline --quick --inflight 1 $*"  # serial passes: per-kernel durations without overlap
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
python3 tools/rocpd_summary.py $OUT/trace/run_results.db > $R/gpurun_out/${TAG}_kernel_stats.txt 2>&1
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc$i -o run -- python3 bench.py $ARGS > $OUT/pmc$i.log 2>&1
  python3 tools/rocpd_summary.py $OUT/pmc$i/run_results.db 2>&1 | sed -n '/PMC counters/,$p' > $OUT/pmc$i.txt
done
cat $OUT/pmc*.txt > $R/gpurun_out/${TAG}_pmc.txt
rm -rf $OUT/trace $OUT/pmc[0-9]*/  # keep the text, drop the databases
tail -3 $OUT/trace.log | cut -c1-600
