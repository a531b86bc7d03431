#!/bin/bash
# Round 4's evidence, made on the GPU box in one gpurun call (from the repo root):
#   tools/profile_r04.sh            -> gpurun_out/r04_*  (copy what is to be judged into profiles/)
# kernel-trace stats + PMC counters (separate passes, tools/profile_pmc.sh) of the pair kernel at 1 M regions in the timed
# region's 512-thread form and alone (1024 threads), at 10 M regions, of the root pass, and the bench line itself.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
GFFX_HIP_WIN_THREADS=512 bash tools/profile_pmc.sh r04_joinA_pairs_1m_512
bash tools/profile_pmc.sh r04_joinA_pairs_1m
bash tools/profile_pmc.sh r04_joinA_pairs_10m --queries-per-gpu 10000000
python3 bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_stderr.txt
tail -c 600 gpurun_out/r04_bench_line.json
