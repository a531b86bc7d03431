#!/bin/bash
# round 6, call 31: per-wave phase stamps with the continuation lines in: what the slowest waves of a lone sorted 1 M launch do now
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_31.txt
: > $O
for ps in 2 0; do
  echo "== kb6h_ws nq=1000000 presort=$ps" >> $O
  timeout 120 tools/_kb/kb6h_ws 1000000 5 258 10 $ps 2>&1 | grep "wave stamps\|  block\|    wave\|pass:" | head -60 >> $O
done
cat $O | cut -c1-230
