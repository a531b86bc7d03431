#!/bin/bash
# round 6, call 22: phase stamps of EVERY wave of the slowest blocks of a lone sorted 1 M launch: which wave holds the block up, and in which phase
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_22.txt
: > $O
for ps in 2 0; do
  echo "== kb6_ws nq=1000000 presort=$ps" >> $O
  timeout 120 tools/_kb/kb6_ws 1000000 5 258 10 $ps 2>&1 | grep "wave stamps\|  block\|    wave\|pass:" >> $O
done
cat $O | cut -c1-200
