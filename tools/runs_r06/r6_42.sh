#!/bin/bash
# round 6, call 42: bench.py's sorted group leg taken apart (block width, grid), next to kbench on the same data
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python tools/_kb/sorted_group_dbg.py 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/r6_42.txt | cut -c1-250
export KB_DATA=$R/tools/_kb/data/bench_c2.bin
for th in 512 1024 0; do GFFX_HIP_WIN_THREADS=$th KB_GROUP=8 GFFX_HIP_GROUP=1 timeout 120 tools/_kb/kb6d 1000000 5 258 40 2 2>&1 | grep "group launch\|group of" | tail -2 | sed "s/^/kbench WIN_THREADS=$th /" | tee -a $R/gpurun_out/r6_42.txt; done
