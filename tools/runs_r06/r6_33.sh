#!/bin/bash
# round 6, call 33: the round's evidence set again (tools/profile_r06.sh), on the kernels with continuation lines
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/profile_r06.sh > $R/gpurun_out/r6_33.txt 2>&1
tail -5 $R/gpurun_out/r6_33.txt | cut -c1-300
cat $R/gpurun_out/r06_inflight_sweep.txt
