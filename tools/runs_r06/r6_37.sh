#!/bin/bash
# round 6, call 37: long lists (8 .. 32 roots) by the whole wave, one record per lane (pair_long_count / pair_long_place, two calls):
# parity first (with the new continuation-line test), then the committed kernels (kb6h) against it (kb6w), and the per-wave stamps of a lone sorted launch
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_37.txt
: > $O
timeout 1800 python -m pytest tests/test_join_a_gpu.py tests/test_sorted_gpu.py tests/test_group_gpu.py tests/test_fuzz_gpu.py tests/test_continuation_lines_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 >> $O
for nq in 1000000 10000000; do for ps in 2 0; do for fl in 258 520; do
  echo "== nq=$nq presort=$ps flags=$fl" >> $O
  for b in kb6h kb6w kb6h kb6w; do timeout 120 tools/_kb/$b $nq 5 $fl 40 $ps 2>&1 | grep "pass:" | sed "s/^/$b /" >> $O; done
done; done; done
for ps in 0 2; do
  echo "== 8 x 1 M in one launch, presort=$ps" >> $O
  for b in kb6h kb6w kb6h kb6w; do KB_GROUP=8 GFFX_HIP_GROUP=1 timeout 120 tools/_kb/$b 1000000 5 258 40 $ps 2>&1 | grep "group launch\|MISMATCH" | sed "s/^/$b /" >> $O; done
done
for m in 0 1; do
  echo "== KB_MODE=$m (0 contained, 1 contains-region), 1 M random / sorted" >> $O
  for ps in 0 2; do for b in kb6h kb6w; do KB_MODE=$m timeout 120 tools/_kb/$b 1000000 5 258 40 $ps 2>&1 | grep "pass:" | sed "s/^/$b ps=$ps /" >> $O; done; done
done
echo "== kb6w_ws nq=1000000 presort=2" >> $O
timeout 120 tools/_kb/kb6w_ws 1000000 5 258 10 2 2>&1 | grep "wave stamps\|  block\|    wave\|pass:" | head -40 >> $O
cat $O | cut -c1-220
