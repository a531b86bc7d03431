#!/bin/bash
# round 6, call 20: after the ticket-word fix: the whole GPU suite, a fuzz campaign, smoke
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_20.txt
: > $O
timeout 600 python -m pytest tests/test_group_gpu.py -m gpu -x -q 2>&1 | tail -4 >> $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 >> $O
timeout 1200 python tools/fuzz_parity.py 500 6630 2>&1 | tail -2 >> $O
timeout 600 python tools/fuzz_cli.py 30 6631 2>&1 | tail -1 >> $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O
cat $O | cut -c1-300
