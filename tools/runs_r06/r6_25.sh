#!/bin/bash
# round 6, call 25: what the slow waves of a sorted launch carry: per wave, the most sweeps one lane has and the most list entries one lane walks
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_25.txt
: > $O
for ps in 2 0; do
  echo "== kb6t_ws nq=1000000 presort=$ps" >> $O
  timeout 120 tools/_kb/kb6t_ws 1000000 5 258 10 $ps 2>&1 | grep "wave stamps\|  block\|    wave\|pass:" | head -60 >> $O
done
cat $O | cut -c1-220
