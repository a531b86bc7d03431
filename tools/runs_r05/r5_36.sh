#!/bin/bash
# round 5, call 36: after the in-flight grid rule -- rocprofv3 + PMC of the timed region's kernel variant (512 threads x 256 blocks), the whole GPU suite, a campaign
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
GFFX_HIP_WIN_THREADS=512 GFFX_HIP_FUSED_BLOCKS=256 bash tools/profile_pmc.sh r05_joinA_pairs_1m_512x256 > gpurun_out/r5_36_profile.log 2>&1
O=$R/gpurun_out/r5_36.txt
: > $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 >> $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> $O
python tools/fuzz_parity.py 400 31 2>&1 | tail -1 >> $O
python tools/fuzz_cli.py 30 32 2>&1 | tail -1 >> $O
head -5 gpurun_out/r05_joinA_pairs_1m_512x256_kernel_stats.txt | cut -c1-150 >> $O
cat $O
