#!/bin/bash
# round 6, call 41: kbench on bench.py's own data (KB_DATA: synth.gencode_like_roots seed 42, synth_bed seed 1001): why its sorted batch is
# slower than kbench's -- lone launches, a launch for 8 batches, and the per-wave stamps of the lone sorted launch
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_41.txt
: > $O
export KB_DATA=$R/tools/_kb/data/bench_c2.bin
for ps in 0 2; do
  echo "== bench data, presort=$ps: lone launch; 8 batches in one launch" >> $O
  for rep in 1 2; do KB_GROUP=8 GFFX_HIP_GROUP=1 timeout 120 tools/_kb/kb6d 1000000 5 258 40 $ps 2>&1 | grep "^pass:\|group launch\|MISMATCH\|exact sweep" >> $O; done
done
echo "== kb6d_ws bench data presort=2" >> $O
timeout 120 tools/_kb/kb6d_ws 1000000 5 258 10 2 2>&1 | grep "wave stamps\|  block\|    wave\|pass:" | head -60 >> $O
cat $O | cut -c1-220
