#!/bin/bash
# round 6, call 40: a group launch over 8 sorted batches: block width 512 / 1024 / the engine's choice (kbench), against shuffled ones
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_40.txt
: > $O
for th in 0 512 1024; do for ps in 0 2; do
  echo "== 8 x 1 M in one launch, presort=$ps WIN_THREADS=$th" >> $O
  for b in kb6h kb6h; do GFFX_HIP_WIN_THREADS=$th KB_GROUP=8 GFFX_HIP_GROUP=1 timeout 120 tools/_kb/$b 1000000 5 258 40 $ps 2>&1 | grep "group launch\|MISMATCH\|group of" | sed "s/^/$b /" >> $O; done
done; done
cat $O | cut -c1-220
