#!/bin/bash
# round 5, call 33: fewer blocks per launch (GFFX_HIP_FUSED_BLOCKS) so that kernels of different batches are co-resident, 512-thread blocks
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_33.txt
: > $O
for blocks in 256 170 128; do
for inf in 2 3 4; do
  echo -n "threads 512 blocks $blocks inflight $inf: " >> $O
  GFFX_HIP_WIN_THREADS=512 GFFX_HIP_FUSED_BLOCKS=$blocks python bench.py --quick --no-traffic --no-cpu-baseline --repeats 3 --inflight $inf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.2f G/s, %.3f us per pass, kernel %.2f us (%d threads), frac %.4f' % (d['value']/1e9, d['us_per_pass'], d['roofline']['pass_kernel_us'], d['roofline']['block_threads'], d['roofline']['frac']))" >> $O
done
done
cat $O
