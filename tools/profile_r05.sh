#!/bin/bash
# Round 5's evidence, made on the GPU box in one gpurun call (from the repo root):
#   tools/profile_r05.sh            -> gpurun_out/r05_*  (copy what is to be judged into profiles/)
# kernel-trace stats + PMC counters (separate passes, tools/profile_pmc.sh; one kernel variant and one batch size per file) of the
# pair kernel at 1 M regions in the timed region's form (512 threads x 256 blocks; and x 489, two batches in flight) and alone (1024 threads), at 10 M regions, of the mixed form on
# wide regions and on a batch with every tenth row SV-sized, every kernel of a full bench.py run, Join B's kernels, and the bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
GFFX_HIP_WIN_THREADS=512 bash tools/profile_pmc.sh r05_joinA_pairs_1m_512
GFFX_HIP_WIN_THREADS=512 GFFX_HIP_FUSED_BLOCKS=256 bash tools/profile_pmc.sh r05_joinA_pairs_1m_512x256  # the timed region's launches: one 512-thread block per CU (three batches in flight)
bash tools/profile_pmc.sh r05_joinA_pairs_1m
bash tools/profile_pmc.sh r05_joinA_pairs_10m --queries-per-gpu 10000000
bash tools/profile_pmc.sh r05_joinA_wide_1m --region-width 100 200000 --offsets u64
bash tools/profile_pmc.sh r05_joinA_mixed_1m --wide-every 10
(cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" && timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r05_all/trace -o run -- python3 bench.py --no-cpu-baseline --no-traffic > gpurun_out/r05_all_bench.log 2>&1; python3 tools/rocpd_summary.py gpurun_out/r05_all/trace/run_results.db > gpurun_out/r05_all_kernels_stats.txt 2>&1; rm -rf gpurun_out/r05_all)
(cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" && timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r05_jb/trace -o run -- python3 tools/joinb_bench.py --quick 1000000 > gpurun_out/r05_joinB_bench_quick.log 2>&1; python3 tools/rocpd_summary.py gpurun_out/r05_jb/trace/run_results.db > gpurun_out/r05_joinB_kernel_stats.txt 2>&1; rm -rf gpurun_out/r05_jb)
python3 bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_stderr.txt
tail -c 600 gpurun_out/r05_bench_line.json
