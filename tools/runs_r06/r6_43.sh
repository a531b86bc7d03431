#!/bin/bash
# round 6, call 43: the block width of a pair pass by the dense-round census (pair_threads): the committed kernels (kb6h) against the
# kernels that count dense wave rounds (kb6n); the bench line's value for shuffled and for sorted batches in flight; parity subsets
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_43.txt
: > $O
for nq in 1000000 10000000; do for ps in 2 0; do
  echo "== nq=$nq presort=$ps flags=258" >> $O
  for b in kb6h kb6n kb6h kb6n; do timeout 120 tools/_kb/$b $nq 5 258 40 $ps 2>&1 | grep "pass:" | sed "s/^/$b /" >> $O; done
done; done
export KB_DATA=$R/tools/_kb/data/bench_c2.bin
for ps in 0 2; do
  echo "== bench data: 8 x 1 M in one launch (wall, launches back to back; then serial launches), presort=$ps" >> $O
  for b in kb6d kb6n; do KB_GROUP=8 GFFX_HIP_GROUP=1 timeout 120 tools/_kb/$b 1000000 5 258 40 $ps 2>&1 | grep "group launch\|group of\|MISMATCH" | tail -2 | sed "s/^/$b /" >> $O; done
  echo "== bench data: 16 batches in flight, two groups on two streams (wall), presort=$ps" >> $O
  for b in kb6d kb6n; do KB_GROUP=16 timeout 120 tools/_kb/$b 1000000 5 258 40 $ps 2>&1 | grep "group of\|MISMATCH" | tail -1 | sed "s/^/$b /" >> $O; done
done
unset KB_DATA
for pre in none chr_start; do
  echo "== bench.py --quick --presort $pre" >> $O
  timeout 600 python bench.py --quick --no-traffic --no-cpu-baseline --presort $pre 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
g = d.get('group_launch') or {}
print('value %.1f G regions/s (%.2f us per pass); launches of %s batches, %.2f us each, frac %.3f, block_threads %s' % (d['value'] / 1e9, d['us_per_pass'], g.get('batches_per_launch'), d['roofline']['pass_kernel_us'], d['roofline']['frac'], d['roofline'].get('block_threads')))" >> $O 2>&1
done
timeout 1500 python -m pytest tests/test_sorted_gpu.py tests/test_group_gpu.py tests/test_continuation_lines_gpu.py tests/test_join_a_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -3 >> $O
cat $O | cut -c1-230
