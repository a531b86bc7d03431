#!/bin/bash
# round 6, call 35: fuzz campaigns with fresh seeds and smoke() on the final kernels (continuation lines in)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_35.txt
: > $O
timeout 1500 python tools/fuzz_parity.py 900 6640 2>&1 | tail -2 >> $O
timeout 600 python tools/fuzz_lines.py 300 6641 2>&1 | tail -2 >> $O
timeout 900 python tools/fuzz_cli.py 30 6642 2>&1 | tail -2 >> $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $O
cat $O | cut -c1-250
