"""bench.py --gpus N, N > 1, on the hardware that is there: the ranks share the box's one GPU and talk over gloo
(GFFX_BENCH_BACKEND=gloo) -- everything but the RCCL transport of the driver's 8-GPU run is exercised: the spawn of the ranks,
the chromosome-bucket sharding (BASELINE configs[3], commands/intersect.rs:114-120), the max-over-ranks timing, the hit-count
all-gather, and the JSON line the driver parses.  The line's totals are checked against the oracle on the UNSHARDED batch."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run_bench(extra):
    env = dict(os.environ, GFFX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--quick", "--steps", "2", "--warmup", "1",
                        "--passes-per-step", "10", "--repeats", "2", "--no-traffic", "--cpu-seconds", "1"] + extra,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    # stdout is the ONE JSON line and nothing else (gloo's / RCCL's banners go to stderr): a driver may json.loads() it whole
    line = json.loads(r.stdout)
    assert r.stdout.count("\n") == 1, r.stdout[-2000:]
    return line


def _oracle_pairs(n, seed):
    from gffx_amd import synth
    from oracle import binding as ob

    roots = synth.gencode_like_roots(63000, seed=42)
    regions = synth.synth_bed(n, seed=seed)
    oix = ob.OracleIndex.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
    trip, _ = oix.query_features(regions, 2, False)
    return len(trip)


def test_bench_two_ranks_weak_scaling_line():
    line = _run_bench([])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["unit"] == "queries/s" and line["value"] > 0
    cfg = line["config"]
    assert cfg["regions_total"] == 2_000_000
    assert cfg["kept_pairs_total"] == _oracle_pairs(2_000_000, 1001)  # the two shards' pairs add up to the unsharded batch's
    assert cfg["exchange_ms"] is not None
    assert line["roofline"]["frac"] > 0 and line["roofline"]["traffic"]["hbm_bytes_per_launch"] is None  # (not copied from a file)
    cb = line["cpu_baseline"]
    assert cb and cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0  # a real baseline on the N > 1 line as well


def test_committed_two_rank_line_is_valid_json():
    """profiles/r05_bench_gpus2_gloo.json is what `bench.py --gpus 2` printed on stdout, as it was printed."""
    with open(os.path.join(ROOT, "profiles", "r05_bench_gpus2_gloo.json")) as fh:
        line = json.load(fh)
    assert line["n_gpus"] == 2 and line["config"]["regions_total"] == 2_000_000


def test_bench_two_ranks_strong_scaling_line():
    line = _run_bench(["--scaling", "strong", "--strong-total", "4000000"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    cfg = line["config"]
    assert cfg["regions_total"] == 4_000_000
    assert cfg["kept_pairs_total"] == _oracle_pairs(4_000_000, 1003)
    assert "inside the timed region" in cfg["sharding"]
