#!/bin/bash
# round 6, call 28: the row beyond the batch's end as {n_chr, 0, 1} (kb6q) against today's kernels (kb6), and what the deferred walks
# cost at most (kb6abl: no region is deferred -- timing only, the results are wrong): lone launches and a launch that serves 8 batches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_28.txt
: > $O
for nq in 1000000 10000000; do for ps in 2 0; do for fl in 258 520; do
  echo "== nq=$nq presort=$ps flags=$fl" >> $O
  for b in kb6 kb6q kb6abl kb6 kb6q kb6abl; do timeout 120 tools/_kb/$b $nq 5 $fl 40 $ps 2>&1 | grep "pass:\|exact sweep" | sed "s/^/$b /" >> $O; done
done; done; done
for ps in 0 2; do
  echo "== 8 x 1 M in one launch, presort=$ps" >> $O
  for b in kb6 kb6q kb6abl kb6 kb6q kb6abl; do KB_GROUP=8 GFFX_HIP_GROUP=1 timeout 120 tools/_kb/$b 1000000 5 258 40 $ps 2>&1 | grep "group launch\|MISMATCH" | sed "s/^/$b /" >> $O; done
done
echo "== 999 999 / 1 000 001 / 4097 regions (partial last rounds)" >> $O
for nq in 999999 1000001 4097; do for b in kb6 kb6q; do timeout 120 tools/_kb/$b $nq 5 258 40 0 2>&1 | grep "pass:\|exact sweep\|MISMATCH" | sed "s/^/$b $nq /" >> $O; done; done
timeout 1500 python -m pytest tests/test_join_a_gpu.py tests/test_group_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 >> $O
cat $O | cut -c1-220
