#!/bin/bash
# round 6, call 15: where do the root kernel's +7-10 % against round 5 come from?  kb5 | kb6 | no ticket code in the root kernel | one PairSub record in the kernarg | both
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_15.txt
: > $O
for nq in 1000000 10000000; do for fl in 520 258; do
  echo "== nq=$nq flags=$fl TICKETS=0: kb5 | kb6 | kb6n | kb6s (static record) | kb6ns" >> $O
  for b in kb5 kb6 kb6n kb6s kb6ns; do GFFX_HIP_TICKETS=0 timeout 120 tools/_kb/$b $nq 5 $fl 40 0 2>&1 | grep "pass:" | sed "s/^/$b /" >> $O; done
done; done
cat $O
