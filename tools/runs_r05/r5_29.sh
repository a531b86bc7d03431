#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_29.txt
: > $O
python -m pytest tests/test_cli_gpu.py tests/test_fullsize_gpu.py -x -q -k "cli" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 >> $O
python tools/cli_stages.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | head -30 >> $O
python tools/fuzz_cli.py 40 21 2>&1 | tail -1 >> $O
cat $O
