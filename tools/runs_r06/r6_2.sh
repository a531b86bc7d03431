#!/bin/bash
# round 6, call 2: block lives by XCD (block % 8): is the 13-20 % spread of block lives at 10 M regions a property of the XCDs?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_2.txt
: > $O
for rep in 1 2; do for ps in 0 2; do
  echo "== kb6_st nq=10000000 presort=$ps rep=$rep" >> $O
  timeout 120 tools/_kb/kb6_st 10000000 5 258 10 $ps 2>&1 | grep "blocks:\|by block\|slowest\|fastest\|pass:" >> $O
done; done
echo "== kb6_st nq=4000000" >> $O
timeout 120 tools/_kb/kb6_st 4000000 5 258 10 0 2>&1 | grep "blocks:\|by block\|slowest\|fastest\|pass:" >> $O
cat $O | cut -c1-600
