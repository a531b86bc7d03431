#!/bin/bash
# round 5, call 32: the headline with 1024-thread blocks forced, 1 - 4 batches in flight (the engine's own choice: 512 when batches share the device)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_32.txt
: > $O
for thr in 1024 512; do
for inf in 1 2 3 4; do
  echo -n "threads $thr inflight $inf: " >> $O
  GFFX_HIP_WIN_THREADS=$thr python bench.py --quick --no-traffic --no-cpu-baseline --repeats 3 --inflight $inf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.2f G/s, %.3f us per pass, kernel %.2f us (%d threads), frac %.4f' % (d['value']/1e9, d['us_per_pass'], d['roofline']['pass_kernel_us'], d['roofline']['block_threads'], d['roofline']['frac']))" >> $O
done
done
cat $O
