#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_14.txt
: > $O
python -m pytest tests/test_cli_gpu.py tests/test_depth_gpu.py tests/test_coverage_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 >> $O

python tools/fuzz_cli.py 60 2>&1 | tail -2 >> $O
cat $O
