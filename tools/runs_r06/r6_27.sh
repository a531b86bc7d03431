#!/bin/bash
# round 6, call 27: calls 23/24/26 again with all binaries built the way the library is (tools/kb_build.sh: the atomic optimizer off):
# today's kernels (kb6) | rows beyond the end skip the sweep (kb6p) | + lists touched before the walks (kb6t) | + one walk per list (kb6m)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_27.txt
: > $O
for nq in 1000000 10000000; do for ps in 2 0; do for fl in 258 520; do
  echo "== nq=$nq presort=$ps flags=$fl" >> $O
  for b in kb6 kb6p kb6t kb6m kb6 kb6p kb6t kb6m; do timeout 120 tools/_kb/$b $nq 5 $fl 40 $ps 2>&1 | grep "pass:" | sed "s/^/$b /" >> $O; done
done; done; done
echo "== 999 999 / 1 000 001 / 4097 regions (partial last rounds)" >> $O
for nq in 999999 1000001 4097; do for b in kb6 kb6p; do timeout 120 tools/_kb/$b $nq 5 258 40 0 2>&1 | grep "pass:\|exact sweep\|MISMATCH" | sed "s/^/$b $nq /" >> $O; done; done
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 >> $O
cat $O | cut -c1-220
