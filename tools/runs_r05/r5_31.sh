#!/bin/bash
# round 5, call 31: two more seeds of the Join A campaign, one of the wide-form check at full size
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_31.txt
: > $O
for seed in 21 22; do
  echo -n "fuzz_parity 1200 seed $seed: " >> $O; python tools/fuzz_parity.py 1200 $seed 2>&1 | tail -1 >> $O
done
cat $O
