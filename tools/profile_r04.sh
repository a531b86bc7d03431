#!/bin/bash
# Round 4's evidence, made on the GPU box in one gpurun call (from the repo root):
#   tools/profile_r04.sh            -> gpurun_out/r04_*  (copy what is to be judged into profiles/)
# kernel-trace stats + PMC counters (separate passes, tools/profile_pmc.sh) of the pair kernel at 1 M regions in the timed
# region's 512-thread form and alone (1024 threads), at 10 M regions, of the wide form, every kernel of a full bench.py run,
# and the bench line itself.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
GFFX_HIP_WIN_THREADS=512 bash tools/profile_pmc.sh r04_joinA_pairs_1m_512
bash tools/profile_pmc.sh r04_joinA_pairs_1m
bash tools/profile_pmc.sh r04_joinA_pairs_10m --queries-per-gpu 10000000
# the wide form: 1 M regions of width U[100, 200000] (bench.py's wide_regions leg as a run of its own), u64 offsets
bash tools/profile_pmc.sh r04_joinA_wide_1m --region-width 100 200000 --offsets u64
# every kernel of one full bench.py run
(cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" && timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r04_all/trace -o run -- python3 bench.py --no-cpu-baseline --no-traffic > gpurun_out/r04_all_bench.log 2>&1; python3 tools/rocpd_summary.py gpurun_out/r04_all/trace/run_results.db > gpurun_out/r04_all_kernels_stats.txt 2>&1; rm -rf gpurun_out/r04_all)
python3 bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_stderr.txt
tail -c 600 gpurun_out/r04_bench_line.json
