#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_22.txt
: > $O
export GFFX_HIP_WIN_WIDE=2 KB_MODE=0
for rep in 1 2; do
for v in c0p0 c2p0 c2p1 c2p3 c1p1 c0p1; do
  echo -n "$v mixed pairs: " >> $O; timeout 60 tools/_kb/kb_$v 1000000 5 258 50 0 10000 10 2>&1 | grep "pass:" >> $O
  echo -n "$v mixed roots: " >> $O; timeout 60 tools/_kb/kb_$v 1000000 5 520 50 0 10000 10 2>&1 | grep "pass:" >> $O
  echo -n "$v wide  pairs: " >> $O; timeout 60 tools/_kb/kb_$v 1000000 5 258 50 0 200000 0 2>&1 | grep "pass:" >> $O
  echo -n "$v wide  roots: " >> $O; timeout 60 tools/_kb/kb_$v 1000000 5 520 50 0 200000 0 2>&1 | grep "pass:" >> $O
done
done
cat $O
