#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_19.txt
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k config2_join_a_10m 2>&1 | tail -60 > $O
cat $O | cut -c1-300
