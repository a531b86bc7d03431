#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/cli_stages.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | head -60 > $R/gpurun_out/r5_25.txt
cat $R/gpurun_out/r5_25.txt
