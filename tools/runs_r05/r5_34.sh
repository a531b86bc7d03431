#!/bin/bash
# round 5, call 34: finer sweep around 256 blocks x 3 batches in flight
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_34.txt
: > $O
run() {  # threads blocks inflight
  echo -n "threads $1 blocks $2 inflight $3: " >> $O
  GFFX_HIP_WIN_THREADS=$1 GFFX_HIP_FUSED_BLOCKS=$2 python bench.py --quick --no-traffic --no-cpu-baseline --repeats 3 --inflight $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.2f G/s, %.3f us per pass, kernel %.2f us (%d threads), frac %.4f' % (d['value']/1e9, d['us_per_pass'], d['roofline']['pass_kernel_us'], d['roofline']['block_threads'], d['roofline']['frac']))" >> $O
}
for blocks in 224 256 288 320 384; do for inf in 3 5 6; do run 512 $blocks $inf; done; done
for inf in 2 3 4; do run 1024 128 $inf; done
run 512 256 3
run 512 0 3
cat $O
