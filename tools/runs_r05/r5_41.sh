#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_join_a_gpu.py -x -q -k "launch_grid" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 > $R/gpurun_out/r5_41.txt
cat $R/gpurun_out/r5_41.txt
