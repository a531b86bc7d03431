#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2; do echo "## process $i"; tools/_kb/alloc_costs; done > $R/gpurun_out/r5_28.txt 2>&1
cat $R/gpurun_out/r5_28.txt
