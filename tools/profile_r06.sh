#!/bin/bash
# Round 6's evidence, made on the GPU box in one gpurun call (from the repo root):
#   tools/profile_r06.sh            -> gpurun_out/r06_*  (copy what is to be judged into profiles/)
# kernel-trace stats + PMC counters (separate passes, tools/profile_pmc.sh; one kernel variant and one batch size per file) of
#   * the launch the timed region runs: ONE launch for a group of 8 batches of 1 M regions (512-thread blocks: two groups alternate),
#   * a lone 1 M-region launch, a 10 M-region launch, the same two sorted by (seqid, start), the mixed form on wide regions and on a
#     batch with every tenth row SV-sized,
# the kernel trace of the DEFAULT bench.py timed region (16 batches in flight: two groups of 8 on two streams) with its concurrency line,
# the sweep over batches in flight, and the bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
GFFX_HIP_GROUP=1 GFFX_HIP_WIN_THREADS=512 bash tools/profile_pmc.sh r06_joinA_group8_1m --inflight 8 --passes-per-step 8
bash tools/profile_pmc.sh r06_joinA_pairs_1m
bash tools/profile_pmc.sh r06_joinA_pairs_10m --queries-per-gpu 10000000
bash tools/profile_pmc.sh r06_joinA_sorted_1m --presort chr_start
bash tools/profile_pmc.sh r06_joinA_sorted_10m --presort chr_start --queries-per-gpu 10000000
bash tools/profile_pmc.sh r06_joinA_wide_1m --region-width 100 200000 --offsets u64
bash tools/profile_pmc.sh r06_joinA_mixed_1m --wide-every 10
# the timed configuration: the program itself after `--`, default arguments but the legs that are not the timed region
(cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" && timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r06_timed/trace -o run -- python3 bench.py --quick --no-traffic --no-cpu-baseline > gpurun_out/r06_timed_bench.log 2>&1; python3 tools/rocpd_summary.py gpurun_out/r06_timed/trace/run_results.db > gpurun_out/r06_timed_config_trace.txt 2>&1; rm -rf gpurun_out/r06_timed)
(cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" && timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r06_all/trace -o run -- python3 bench.py --no-cpu-baseline --no-traffic > gpurun_out/r06_all_bench.log 2>&1; python3 tools/rocpd_summary.py gpurun_out/r06_all/trace/run_results.db > gpurun_out/r06_all_kernels_stats.txt 2>&1; rm -rf gpurun_out/r06_all)
: > gpurun_out/r06_inflight_sweep.txt
for n in 1 2 3 4 5 6 7 8 10 12 16 24 32; do
  python3 bench.py --quick --no-traffic --no-cpu-baseline --inflight $n 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
g = d.get('group_launch') or {}
print('inflight %2d: %.1f G regions/s (%.2f us per pass; repeats %.1f .. %.1f); launches of %s batches on %s streams, %.2f us each; roofline.frac of that launch %.3f' % (d['config']['batches_in_flight'], d['value'] / 1e9, d['us_per_pass'], d['repeats']['value_min'] / 1e9, d['repeats']['value_max'] / 1e9, g.get('batches_per_launch', 1), g.get('streams', 'their own'), d['roofline']['pass_kernel_us'], d['roofline']['frac']))" >> gpurun_out/r06_inflight_sweep.txt
done
python3 bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_stderr.txt
tail -c 600 gpurun_out/r06_bench_line.json
