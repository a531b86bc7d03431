#!/bin/bash
# round 6, call 21: where the slowest blocks of a lone sorted 1 M launch spend their time (phase stamps per block)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_21.txt
: > $O
for ps in 2 0; do for th in 0 512; do
  echo "== kb6_st nq=1000000 presort=$ps WIN_THREADS=$th" >> $O
  GFFX_HIP_WIN_THREADS=$th timeout 120 tools/_kb/kb6_st 1000000 5 258 10 $ps 2>&1 | grep "blocks:\|block [0-9]*:\|pass:\|wave rounds" >> $O
done; done
cat $O | cut -c1-400
