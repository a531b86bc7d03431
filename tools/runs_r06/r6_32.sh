#!/bin/bash
# round 6, call 32: continuation lines in the library: the whole GPU suite, then the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_32.txt
: > $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 >> $O
timeout 900 python bench.py > $R/gpurun_out/r6_32_bench.json 2> $R/gpurun_out/r6_32_bench.err
tail -c 6000 $R/gpurun_out/r6_32_bench.json >> $O
cat $O | cut -c1-6000
