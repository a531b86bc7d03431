#!/bin/bash
# rocprofv3 kernel trace of bench.py's TIMED configuration (two batches in flight) on the GPU box:
#   tools/profile_overlap.sh <tag>   ->  gpurun_out/<tag>_overlap_stats.txt (kernel stats + how the launches of the two
# streams overlap)
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/otrace -o run -- python3 bench.py --steps 2 --passes-per-step 100 --repeats 1 --warmup 5 --no-cpu-baseline --quick --inflight 2 $* > $OUT/otrace.log 2>&1
python3 tools/rocpd_summary.py $OUT/otrace/run_results.db > $R/gpurun_out/${TAG}_overlap_stats.txt 2>&1
rm -rf $OUT/otrace
tail -2 $OUT/otrace.log | cut -c1-400
