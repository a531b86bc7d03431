#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/runs_r05/roots_in_flight.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $R/gpurun_out/r5_38.txt
cat $R/gpurun_out/r5_38.txt
