#!/bin/bash
# round 5, call 27: campaigns with OTHER seeds on the final binaries (every earlier campaign of the round drew seed 6)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_27.txt
: > $O
for seed in 11 12; do
  echo -n "fuzz_parity 1500 seed $seed: " >> $O; python tools/fuzz_parity.py 1500 $seed 2>&1 | tail -1 >> $O
done
echo -n "fuzz_lines 1000 seed 13: " >> $O; python tools/fuzz_lines.py 1000 13 2>&1 | tail -1 >> $O
echo -n "fuzz_cli 60 seed 14: " >> $O; python tools/fuzz_cli.py 60 14 2>&1 | tail -1 >> $O
cat $O
