#!/bin/bash
# round 5, call 42: the mixed form (every tenth row SV-sized) with three batches in flight: the engine's grid (one block per CU) against a block per slot
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_42.txt
: > $O
for blocks in 0 512; do
  echo -n "mixed batch, inflight 3, FUSED_BLOCKS $blocks: " >> $O
  GFFX_HIP_FUSED_BLOCKS=$blocks python bench.py --quick --no-traffic --no-cpu-baseline --repeats 3 --inflight 3 --wide-every 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('value %.2f G/s, %.3f us per pass, kernel alone %.2f us (%d threads, %s blocks)' % (d['value']/1e9, d['us_per_pass'], r['pass_kernel_us'], r['block_threads'], r['blocks']))" >> $O
done
echo -n "mixed batch, inflight 1: " >> $O
python bench.py --quick --no-traffic --no-cpu-baseline --repeats 3 --inflight 1 --wide-every 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('value %.2f G/s, %.3f us per pass, kernel alone %.2f us (%d threads, %s blocks)' % (d['value']/1e9, d['us_per_pass'], r['pass_kernel_us'], r['block_threads'], r['blocks']))" >> $O
cat $O
