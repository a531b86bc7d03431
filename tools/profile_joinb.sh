#!/bin/bash
# rocprofv3 evidence for Join B on the GPU box (run through gpurun from the repo root):  tools/profile_joinb.sh <tag>
# Writes gpurun_out/<tag>_kernel_stats.txt (kernel trace) and gpurun_out/<tag>_pmc.txt (FETCH_SIZE / WRITE_SIZE and the
# instruction mix per dispatch, separate passes, --kernel-trace only) of `tools/joinb_bench.py --quick 1000000`.
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- python3 tools/joinb_bench.py --quick 1000000 > $OUT/trace.log 2>&1
python3 tools/rocpd_summary.py $OUT/trace/run_results.db > $R/gpurun_out/${TAG}_kernel_stats.txt 2>&1
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc$i -o run -- python3 tools/joinb_bench.py --quick 1000000 > $OUT/pmc$i.log 2>&1
  python3 tools/rocpd_summary.py $OUT/pmc$i/run_results.db 2>&1 | sed -n '/PMC counters/,$p' > $OUT/pmc$i.txt
done
cat $OUT/pmc*.txt > $R/gpurun_out/${TAG}_pmc.txt
rm -rf $OUT/trace $OUT/pmc[0-9]*/
head -40 $R/gpurun_out/${TAG}_pmc.txt | cut -c1-200
