#!/bin/bash
# round 6, call 44: rocprofv3 stats + PMC of ONE launch for 8 SORTED batches of 1 M regions, 1024-thread blocks (the serial group launch of
# bench.py's sorted_bed.group_launch leg: launch_us_1024_threads)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
GFFX_HIP_GROUP=1 GFFX_HIP_WIN_THREADS=1024 bash tools/profile_pmc.sh r06_joinA_sorted_group8_1m --inflight 8 --passes-per-step 8 --presort chr_start > $R/gpurun_out/r6_44.txt 2>&1
tail -3 $R/gpurun_out/r6_44.txt | cut -c1-300
grep -m4 "k_join_pairs" $R/gpurun_out/r06_joinA_sorted_group8_1m_kernel_stats.txt | cut -c1-220
