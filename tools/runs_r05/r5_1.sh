#!/bin/bash
# round 5, call 1: the gather cost model
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 tools/_kb/gather_ubench model > gpurun_out/r05_gather_ubench.txt 2>&1
echo "--- grid 256" >> gpurun_out/r05_gather_ubench.txt
timeout 300 tools/_kb/gather_ubench model 8388608 256 >> gpurun_out/r05_gather_ubench.txt 2>&1
tail -5 gpurun_out/r05_gather_ubench.txt
