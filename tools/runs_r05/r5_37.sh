#!/bin/bash
# round 5, call 37: the grid rule limited to batches of <= 1024 rounds: the headline again, and 10 M-region batches in flight (full grid)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_37.txt
: > $O
for args in "--inflight 3" "--inflight 3 --queries-per-gpu 10000000 --steps 3 --passes-per-step 20" "--inflight 3 --queries-per-gpu 2000000 --steps 10 --passes-per-step 50"; do
  echo -n "defaults, $args: " >> $O
  python bench.py --quick --no-traffic --no-cpu-baseline --repeats 3 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('value %.2f G/s, %.3f us per pass, kernel alone %.2f us (%d threads, %s blocks), frac %.4f' % (d['value']/1e9, d['us_per_pass'], r['pass_kernel_us'], r['block_threads'], r['blocks'], r['frac']))" >> $O
done
for blocks in 256 512; do
  echo -n "forced $blocks blocks, 10 M regions, inflight 3: " >> $O
  GFFX_HIP_FUSED_BLOCKS=$blocks python bench.py --quick --no-traffic --no-cpu-baseline --repeats 3 --inflight 3 --queries-per-gpu 10000000 --steps 3 --passes-per-step 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('value %.2f G/s, %.3f us per pass, kernel alone %.2f us (%d threads, %s blocks)' % (d['value']/1e9, d['us_per_pass'], r['pass_kernel_us'], r['block_threads'], r['blocks']))" >> $O
  echo -n "forced $blocks blocks, 2 M regions, inflight 3: " >> $O
  GFFX_HIP_FUSED_BLOCKS=$blocks python bench.py --quick --no-traffic --no-cpu-baseline --repeats 3 --inflight 3 --queries-per-gpu 2000000 --steps 10 --passes-per-step 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('value %.2f G/s, %.3f us per pass, kernel alone %.2f us (%d threads, %s blocks)' % (d['value']/1e9, d['us_per_pass'], r['pass_kernel_us'], r['block_threads'], r['blocks']))" >> $O
done
python -m pytest tests/test_join_a_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2 >> $O
cat $O
