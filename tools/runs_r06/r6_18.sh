#!/bin/bash
# round 6, call 18: three kinds of launch (plain / group / tickets), each a translation unit: the whole GPU suite, fuzz, kbench against round 5, the quick bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_18.txt
: > $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 >> $O
timeout 1200 python tools/fuzz_parity.py 400 6620 2>&1 | tail -2 >> $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O
if [ -x tools/_kb/kb6 ]; then
for nq in 1000000 10000000; do for fl in 258 520; do
  echo "== nq=$nq flags=$fl: kb5 | kb6" >> $O
  for b in kb5 kb6; do timeout 120 tools/_kb/$b $nq 5 $fl 40 0 2>&1 | grep "pass:" | sed "s/^/$b /" >> $O; done
done; done
echo "== kb6 group of 8, GROUP=1; group of 16; group of 4" >> $O
KB_GROUP=8 GFFX_HIP_GROUP=1 timeout 120 tools/_kb/kb6 1000000 5 258 40 0 2>&1 | grep "group launch\|MISMATCH" >> $O
KB_GROUP=16 timeout 120 tools/_kb/kb6 1000000 5 258 40 0 2>&1 | grep "group of\|MISMATCH" | tail -1 >> $O
KB_GROUP=4 timeout 120 tools/_kb/kb6 1000000 5 258 40 0 2>&1 | grep "group of\|MISMATCH" | tail -1 >> $O
fi
python bench.py --quick --cpu-seconds 2 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench quick: value %.1f G/s, launch %.2f us, frac %.3f, serial %.2f us frac %.3f' % (d['value']/1e9, d['roofline']['pass_kernel_us'], d['roofline']['frac'], d['serial']['roofline']['pass_kernel_us'], d['serial']['roofline']['frac']))" >> $O
cat $O | cut -c1-300
