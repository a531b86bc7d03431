#!/bin/bash
# round 5, the last call: the whole GPU suite and smoke() on the final tree
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_40.txt
: > $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 >> $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> $O
python bench.py --quick --no-traffic --no-cpu-baseline --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench --quick: value %.2f G/s, %.3f us per pass, kernel alone %.2f us (%d threads, %s blocks), frac %.4f' % (d['value']/1e9, d['us_per_pass'], r['pass_kernel_us'], r['block_threads'], r['blocks'], r['frac']))" >> $O
cat $O
