#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5_21.txt
: > $O
cd $R && python tools/runs_r05/ab_roots.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" >> $O
cd $R/tools/_kb/wt && python $R/tools/runs_r05/ab_roots.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" >> $O
cat $O
