#!/bin/bash
# round 6, call 38: the whole GPU suite on the round's final tree (238 tests), smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_38.txt
: > $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 >> $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $O
cat $O | cut -c1-250
