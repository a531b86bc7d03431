#!/bin/bash
# round 6, call 11: the round's evidence set (tools/profile_r06.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/profile_r06.sh > $R/gpurun_out/r6_11.txt 2>&1
tail -5 $R/gpurun_out/r6_11.txt | cut -c1-300
cat $R/gpurun_out/r06_inflight_sweep.txt
