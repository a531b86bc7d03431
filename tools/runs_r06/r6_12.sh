#!/bin/bash
# round 6, call 12: 64-byte records of the wide form's table: parity (wide-form tests, the forced-form fuzz leg), then kbench wide / mixed
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_12.txt
: > $O
timeout 1500 python -m pytest tests/test_wide_form_gpu.py tests/test_join_a_gpu.py tests/test_group_gpu.py -m gpu -x -q 2>&1 | tail -6 >> $O
timeout 900 python tools/fuzz_parity.py 250 6061 2>&1 | tail -3 >> $O
for mode in 2 0 1; do
  echo "== KB_MODE=$mode all wide (U[100, 200000]) pairs / roots; every tenth row SV-sized pairs / roots" >> $O
  for b in kb6_base kb6; do KB_MODE=$mode timeout 120 tools/_kb/$b 1000000 0 18 40 0 200000 0 2>&1 | grep "pass:" | sed "s/^/$b /" >> $O; done
  for b in kb6_base kb6; do KB_MODE=$mode timeout 120 tools/_kb/$b 1000000 0 520 40 0 200000 0 2>&1 | grep "pass:" | sed "s/^/$b /" >> $O; done
  for b in kb6_base kb6; do KB_MODE=$mode timeout 120 tools/_kb/$b 1000000 0 258 40 0 10000 10 2>&1 | grep "pass:" | sed "s/^/$b /" >> $O; done
  for b in kb6_base kb6; do KB_MODE=$mode timeout 120 tools/_kb/$b 1000000 0 520 40 0 10000 10 2>&1 | grep "pass:" | sed "s/^/$b /" >> $O; done
done
echo "== 10 M: all wide / every tenth row SV-sized (overlap, pairs)" >> $O
for b in kb6_base kb6; do timeout 120 tools/_kb/$b 10000000 0 18 20 0 200000 0 2>&1 | grep "pass:" | sed "s/^/$b /" >> $O; done
for b in kb6_base kb6; do timeout 120 tools/_kb/$b 10000000 0 258 20 0 10000 10 2>&1 | grep "pass:" | sed "s/^/$b /" >> $O; done
cat $O | cut -c1-300
