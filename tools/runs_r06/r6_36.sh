#!/bin/bash
# round 6, call 36: the test written for the continuation lines (an index of clusters of 5 / 6 / 7 roots)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_36.txt
: > $O
timeout 1500 python -m pytest tests/test_continuation_lines_gpu.py -x -q -m gpu 2>&1 | tail -15 >> $O
cat $O | cut -c1-250
