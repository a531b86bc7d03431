#!/bin/bash
# round 6, call 17: what the (unused) ticket code costs a launch that serves a group: kb6 against kb6_nt (ticket code compiled out of the DYN instantiations)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_17.txt
: > $O
for rep in 1 2 3; do for b in kb6 kb6_nt; do
  echo "== $b rep $rep: group of 8 GROUP=1; group of 16; group of 4" >> $O
  KB_GROUP=8 GFFX_HIP_GROUP=1 timeout 120 tools/_kb/$b 1000000 5 258 40 0 2>&1 | grep "group launch\|MISMATCH" >> $O
  KB_GROUP=16 timeout 120 tools/_kb/$b 1000000 5 258 40 0 2>&1 | grep "group of\|MISMATCH" | tail -1 >> $O
  KB_GROUP=4 timeout 120 tools/_kb/$b 1000000 5 258 40 0 2>&1 | grep "group of\|MISMATCH" | tail -1 >> $O
done; done
cat $O
