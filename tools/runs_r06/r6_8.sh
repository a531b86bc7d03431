#!/bin/bash
# round 6, call 8: words per thread for the kept words of deferred regions at 1024 threads (2 / 4 / 6): a sorted batch's dense stretches walk their list tails twice when they do not fit
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_8.txt
: > $O
for b in kb6 kb6_s4 kb6_s6; do for nq in 1000000 10000000; do for ps in 0 2; do
  echo "== $b nq=$nq presort=$ps" >> $O
  timeout 120 tools/_kb/$b $nq 5 258 40 $ps 2>&1 | grep "pass:" >> $O
done; done; done
cat $O
