#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
(time python -m pytest tests/test_wide_form_gpu.py tests/test_fullsize_gpu.py -x -q -k "full_size_properties or sv_sized" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8) > $R/gpurun_out/r5_26.txt 2>&1
cat $R/gpurun_out/r5_26.txt
