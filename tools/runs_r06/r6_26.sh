#!/bin/bash
# round 6, call 26: rows beyond the batch's end no longer take the sweep call (pair_locate): the whole GPU suite, then kb6 against kb6p
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_26.txt
: > $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 >> $O
for nq in 1000000 10000000; do for ps in 2 0; do for fl in 258 520; do
  echo "== nq=$nq presort=$ps flags=$fl: kb6 | kb6p" >> $O
  for b in kb6 kb6p kb6 kb6p; do timeout 120 tools/_kb/$b $nq 5 $fl 40 $ps 2>&1 | grep "pass:\|exact sweep" | sed "s/^/$b /" >> $O; done
done; done; done
echo "== 999 999 / 1 000 001 / 4097 regions (partial last rounds)" >> $O
for nq in 999999 1000001 4097; do timeout 120 tools/_kb/kb6p $nq 5 258 40 0 2>&1 | grep "pass:\|exact sweep\|MISMATCH" | sed "s/^/kb6p $nq /" >> $O; done
cat $O | cut -c1-220
