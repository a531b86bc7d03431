#!/usr/bin/env python3
"""bench.py -- BED overlap queries/sec against a GRCh38-scale GFF index on N MI355X.

One PASS of the intersect hot path = Join A: regions x root intervals -> per-region kept counts (input order), the root_fids
of every kept pair, and one segment base per group of 256 regions (GFFX_OUT_SEGBASE: a region's segment starts at its group's
base + the counts before it) over one batch of synthetic BED regions that is already resident in HBM.  One "step" =
--passes-per-step (default 100) such passes over the SAME resident batch, so that the timed region of K steps lasts tens of
milliseconds whatever K the driver passes (a 1 M-region pass is ~10 us: 20 single passes would be a 0.2 ms measurement).  Workload at N=1 = BASELINE.json configs[1]: 1 M synthetic BED regions (seed 1001, chr ~ length,
width U[100,10000], unsorted) x a GENCODE/GRCh38-shaped index (25 seqids, ~63 k root genes of a ~3.4 M-line annotation,
seed 42), --overlap mode.

The JSON line carries, next to the contract's fields:
  value / ms_per_step   K steps = K x passes-per-step passes issued round-robin to --inflight (default 3) QueryBatch objects (own HIP
                        stream and result buffers each, the same resident regions) by ONE call into the C-ABI
                        (gffx_hip_batches_run_n: the launch loop runs in C): the launch ramp / drain of one pass overlaps the next.
                        The pipeline is warmed (2 x inflight passes, synced) immediately before the timed region.  The timed
                        region is repeated --repeats times (default 5), each bracketed by barrier + synchronize; `value` is the
                        MEDIAN repeat, `repeats` lists them all.
  serial                the same K steps with ONE batch: strictly serial passes (what profiles/*kernel_stats* shows); the engine
                        takes 1024-thread blocks for such a pass (roofline inside), 512-thread blocks while several batches are in flight
  roofline              dominant kernel, HIP-event durations of serial back-to-back launches on the engine's stream
  roofline_10m          the same for a 10 M-region batch (seed 1002); roofline_10m_contained: --contained (configs[2]'s mode);
                        wide_regions: 1 M regions of width U[100, 200000] (AUTO runs the mixed form of k_join_pairs, every lane the
                        wide way); mixed_widths: the headline's batch with every tenth row SV-sized (the mixed form, lane by lane);
                        sorted_bed: the 1 M batch sorted by (seqid, start) as BED files usually are
  cli_pass              the pass the CLI runs (root bitmap only) at 1 M and 10 M regions, next to the root_fid pass
  t_xfer                host regions in (pinned), counts + root_fids back on the host: two batches double-buffered
  t_e2e                 the `gffx` CLI on a 3.5 M-line synthetic GFF3 x the 1 M-row BED: wall clock + its stage timers; and x a
                        100 M-row BED (2.4 GB of text) with 64 host threads
  join_b                Join B (k_lines_exists over a 3.4 M-line table) with the device-built region tables
  depth                 `gffx depth`'s kernel (k_depth_regions) on the same regions against a 3.4 M-line table
  cpu_baseline          the oracle's Join A on 1 thread (the reference is serial there), + all cores, + Join B on all cores

For N>1 (launched by torch.distributed.run, one rank per GPU) --scaling weak (default) gives every rank 1 M regions of an
N x 1 M batch sharded by chromosome bucket (gffx_amd.shard, LPT with splitting; index replicated); --scaling strong shards
BASELINE configs[3]'s 100 M regions over the N ranks.  The only collective is the RCCL all-gather of per-rank hit counts,
once per job: --exchange final-timed puts it inside the timed region (default for --scaling strong).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--passes-per-step", type=int, default=100,
                    help="a step = this many passes over the resident batch (the timed region then lasts tens of ms)")
    ap.add_argument("--repeats", type=int, default=5, help="the K-step timed region is measured this many times; value = median")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--queries-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = --queries-per-gpu regions per rank; strong = --strong-total regions (configs[3]: 100 M) "
                         "sharded over the ranks, the hit-count all-gather inside the timed region")
    ap.add_argument("--strong-total", type=int, default=100_000_000)
    ap.add_argument("--region-width", type=int, nargs=2, default=None, metavar=("LO", "HI"),
                    help="profiling aid: regions of width U[LO, HI] instead of the configuration's U[100, 10000] (the line says so)")
    ap.add_argument("--wide-every", type=int, default=0, metavar="N",
                    help="profiling aid: every N-th region SV-sized (width U[20000, 2000000]): the mixed_widths leg as a run of its own")
    ap.add_argument("--mode", default="overlap", choices=["overlap", "contained", "contains_region"])
    ap.add_argument("--strategy", default="auto", choices=["auto", "direct", "sorted", "fused", "windows"])
    ap.add_argument("--out", default="fids", choices=["counts", "fids", "triples"])
    ap.add_argument("--offsets", default="seg", choices=["seg", "u32", "u64", "none"],
                    help="where the pairs of a region are: seg = one u64 base per group of 256 regions (GFFX_OUT_SEGBASE, windows "
                         "strategy / auto); u32 / u64 = a segment start per region")
    ap.add_argument("--presort", default="none", choices=["none", "chr_end", "chr_start"],
                    help="EXPERIMENT ONLY: reorder the synthetic regions on the host before upload")
    ap.add_argument("--exchange", default=None, choices=["final", "final-timed", "per-step"])
    ap.add_argument("--inflight", type=int, default=16,
                    help="batches in flight in the timed region (own result buffers each).  The engine serves four or more with ONE launch "
                         "per group of up to 8 (gffx_hip_batches_run_n: 16 = two groups of 8 alternating between two streams); up to "
                         "three run pass by pass as in round 5")
    ap.add_argument("--quick", action="store_true", help="headline + roofline + cpu_baseline only (skip the extra legs)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic live (two child runs of this script under rocprofv3 --pmc, ~20 s)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    return ap.parse_args()


def rank_placement(env, n_devices):
    """(rank, world, local_rank, device index) of this process: the driver launches one rank per GPU through
    torch.distributed.run, so LOCAL_RANK r takes device r; with fewer devices than ranks (the 1-GPU box of the gloo tests) the
    ranks wrap around the devices that exist."""
    world = int(env.get("WORLD_SIZE", "1"))
    rank = int(env.get("RANK", "0"))
    local_rank = int(env.get("LOCAL_RANK", "0"))
    if n_devices < 1:
        raise RuntimeError("no HIP device")
    return rank, world, local_rank, local_rank % n_devices


def widen_every(regions, every, seed=1005):
    """Every `every`-th row (a random tenth for every = 10, ...) becomes SV-sized: width U[20 000, 2 000 000] (the mixed_widths leg)."""
    out = regions.copy()
    rng = np.random.default_rng(seed)
    pick = rng.choice(len(out), len(out) // every, replace=False)
    wid = rng.integers(20_000, 2_000_000, len(pick), dtype=np.int64)
    out[pick, 2] = np.minimum(out[pick, 1].astype(np.int64) + wid, 0xFFFFFFF0).astype(np.uint32)
    return out


def bench_regions(synth, shard, n_chr, world, rank, scaling="weak", queries_per_gpu=1_000_000, strong_total=100_000_000, region_width=None,
                  wide_every=0):
    """This rank's regions: weak scaling = its chromosome-bucket shard of an N x --queries-per-gpu batch of configs[1]'s seed,
    strong scaling = its shard of configs[3]'s batch (seed 1003; commands/intersect.rs:114-120 buckets by seqid, the shards are
    LPT-placed bucket slices).  Returns (regions, global region count, configuration name)."""
    strong = scaling == "strong" and world > 1
    if strong:
        nq_global, seed, cfg = strong_total, 1003, "configs[3]"
    else:
        nq_global, seed, cfg = queries_per_gpu * world, 1001, "configs[1]"
    if region_width:
        regions_all = synth.synth_bed(nq_global, seed=seed, width=tuple(region_width))
        cfg += " with region widths U[%d, %d] (NOT the configuration's: --region-width)" % tuple(region_width)
    else:
        regions_all = synth.synth_bed(nq_global, seed=seed)
    if wide_every:
        regions_all = widen_every(regions_all, wide_every)
        cfg += " with every %d-th row widened to U[20000, 2000000] (NOT the configuration's: --wide-every)" % wide_every
    if world > 1:
        regions = np.ascontiguousarray(regions_all[shard.shard_rows(regions_all, n_chr, world, rank)])
    else:
        regions = regions_all
    return regions, nq_global, cfg


def host_info():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu": model, "nproc": os.cpu_count()}


def reference_toolchain():
    """the harness probes for a Rust toolchain; with one (and the reference's crates vendored) the real `gffx`
    could be built and timed -- tools/pin_against_reference.py does that.  Here it only records what it found."""
    cargo = shutil.which("cargo")
    out = {"cargo": None, "reference_binary_timed": False,
           "note": "no Rust toolchain on this box: the baseline below is the C restatement (oracle/), not the Rust binary"}
    if cargo:
        try:
            v = subprocess.run([cargo, "--version"], capture_output=True, text=True, timeout=20).stdout.strip()
        except Exception as exc:  # noqa: BLE001
            v = "cargo found but not runnable: %r" % (exc,)
        out["cargo"] = v
        out["note"] = ("cargo is present; the real binary is built and compared by tools/pin_against_reference.py --cargo (it needs "
                       "the reference tree and its crates, which are not on the GPU box): the baseline below is still the C restatement")
    return out


def cpu_baseline(roots, regions, mode, budget_s):
    """Oracle (C restatement of the reference's serial tree walk, intersect.rs:124-166) timed on this host: 1 thread,
    like the reference.  Bounded sample of the same workload."""
    from oracle import binding as ob

    oix = ob.OracleIndex.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
    n = len(regions)
    oix.query_features(regions[: min(n, 20000)], mode, False)  # warm
    done, t_used, hits, reps = 0, 0.0, 0, 0
    while t_used < budget_s and reps < 400:
        t0 = time.perf_counter()
        t, _ = oix.query_features(regions, mode, False)
        t_used += time.perf_counter() - t0
        done += n
        hits = len(t)
        reps += 1
    return {"value": done / t_used, "unit": "queries/s", "cores": 1, "kind": "port", "host": host_info(),
            "reference_toolchain": reference_toolchain(),
            "sample": "%d x the full %d-region batch of this workload, Join A only (pointer-based centered interval tree, "
                      "serial, as commands/intersect.rs:124-166); C restatement, not the Rust binary" % (reps, n),
            "pairs_per_batch": hits}


def child_json(argv):
    """CPU legs that fork worker pools run as child processes (nothing forks from a process that initialised the GPU)."""
    r = subprocess.run([sys.executable] + argv, capture_output=True, text=True)
    try:
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        return {"error": (r.stderr or r.stdout)[-300:]}


class Pass:
    """K passes over resident regions, round-robin over `inflight` batches."""

    def __init__(self, engine, ix, dev_cols, nq, inflight, mode, out_flags, strategy):
        self.engine, self.mode, self.flags, self.strategy = engine, mode, out_flags, strategy
        self.batches = []
        for i in range(max(1, inflight)):
            bb = engine.QueryBatch(ix, max(nq, 1))
            # every batch in flight reads its OWN copy of the regions (round 6: a launch that serves eight batches would otherwise find
            # seven of its eight region streams in the L2 -- a caller's batches in flight hold different regions)
            cols = dev_cols if i == 0 else tuple(c.clone() for c in dev_cols)
            bb.set_regions_device(cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), nq, keep=cols)
            self.batches.append(bb)
        self.issued = 0

    def step(self):
        self.batches[self.issued % len(self.batches)].run(self.mode, False, self.flags, self.strategy)
        self.issued += 1

    def run_n(self, n_passes):
        """n passes round-robin over the batches, issued by ONE call into the C-ABI (the launch loop runs in C)."""
        arr = (ctypes.c_void_p * len(self.batches))(*[bb._h for bb in self.batches])
        self.engine.check(self.engine.lib().gffx_hip_batches_run_n(arr, len(self.batches), self.mode, 0, self.flags, self.strategy,
                                                                   n_passes))

    def sync(self):
        for bb in self.batches:
            bb.sync()

    def size_and_warm(self, warmup):
        """First pass sizes the pair buffers (a capacity replay can only happen here); returns the pass's kept pairs."""
        b0 = self.batches[0]
        b0.run(self.mode, False, self.flags, self.strategy)
        b0.wait()
        pairs = b0.total_hits
        for bb in self.batches[1:]:
            bb.reserve_hits(pairs + pairs // 8 + 1024)
            bb.run(self.mode, False, self.flags, self.strategy)
            bb.wait()
        for _ in range(warmup * len(self.batches)):
            self.step()
        self.sync()
        return pairs

    def timed(self, n_passes, barrier, torch):
        """One timed region: exactly n_passes passes between barrier + synchronize on both sides."""
        # warm the pipeline immediately before the timed region: >= 2 x inflight passes, drained
        self.run_n(2 * len(self.batches))
        self.sync()
        barrier()
        t0 = time.perf_counter()
        self.run_n(n_passes)
        self.sync()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def check(self, pairs):
        """Every batch's sticky error flag and pair total, once, outside the timed region."""
        for bb in self.batches:
            bb.wait()
            assert bb.total_hits == pairs, (bb.total_hits, pairs)

    def kernel_us(self, n_prof, block_threads=None, blocks=None):
        """HIP-event durations per kernel: serial launches back to back on batch 0's stream.  block_threads: the block width
        the measured passes must use (the engine takes 1024-thread blocks for a pass of 500 000 regions or more that runs ALONE and
        512-thread blocks while another batch is in flight: serial passes measured for a timed region with two batches in
        flight have to be told which kernel that region ran).  blocks: likewise the grid (one 512-thread block per CU from the
        third batch in flight on)."""
        b0 = self.batches[0]
        before = b0.options().get("GFFX_HIP_WIN_THREADS", 0)
        before_blocks = b0.options().get("GFFX_HIP_FUSED_BLOCKS", 0)
        if block_threads:
            b0.set_option("WIN_THREADS", block_threads)
        if blocks:
            b0.set_option("FUSED_BLOCKS", blocks)
        try:
            return self._kernel_us(n_prof)
        finally:
            b0.set_option("WIN_THREADS", before)
            b0.set_option("FUSED_BLOCKS", before_blocks)

    def _kernel_us(self, n_prof):
        b0 = self.batches[0]
        b0.set_profiling(True)
        b0.reset_profile()
        for _ in range(n_prof):
            b0.run(self.mode, False, self.flags, self.strategy)
        b0.sync()
        b0.set_profiling(False)
        kern = {}
        for kid, name in self.engine.KERNEL_NAMES.items():
            ms, n = b0.kernel_ms(kid)
            if n:
                kern[name] = {"avg_us": 1e3 * ms / n, "launches_per_step": n / n_prof}
        # the same passes between ONE event pair: launch-to-launch average without the ~3 us an event pair per launch adds
        self.pass_us_one_event_pair = b0.timed_runs(self.mode, False, self.flags, self.strategy, max(n_prof, 20))
        self.block_threads = b0.block_threads
        self.block_count = b0.block_count
        return kern

    def close(self):
        for bb in self.batches:
            bb.close()


def roofline_obj(kern, nq, pairs, out_b, note, traffic=None, one_pair_us=None, block_threads=None, blocks=None):
    h_bar = pairs / max(nq, 1)
    bytes_per_query = 12.0 + 4.0 + out_b * h_bar  # regions in, count out, pairs out
    per_launch_us = sum(k["avg_us"] * k["launches_per_step"] for k in kern.values())
    # the pass's duration: launches back to back between one pair of HIP events on the engine's stream (it includes the gap
    # between launches and no per-launch event cost: within a few % of the rocprofv3 kernel average in profiles/)
    pass_us = one_pair_us if one_pair_us else per_launch_us
    dominant = max(kern.items(), key=lambda kv: kv[1]["avg_us"] * kv[1]["launches_per_step"])[0] if kern else None
    achieved = (bytes_per_query * nq) / (pass_us * 1e-6) / 1e9 if pass_us > 0 else 0.0
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "frac_of_copy_ceiling": achieved / 6300.0, "traffic": traffic, "dominant_kernel": dominant,
            "regions_per_launch": nq, "pairs_per_region": h_bar, "algorithmic_bytes_per_pass": bytes_per_query * nq,
            "pass_kernel_us": pass_us, "pass_kernel_us_event_pair_per_launch": per_launch_us, "kernels": kern,
            "block_threads": block_threads, "blocks": blocks, "note": note}


def measure_traffic(args, block_threads=None, blocks=None, group=1):
    """roofline.traffic, measured by THIS run: HBM bytes per launch of the dominant kernel from the L2's fabric counters,
    collected as MI355X_MICROARCH.md prescribes -- FETCH_SIZE and WRITE_SIZE in SEPARATE `rocprofv3 --kernel-trace --pmc`
    passes (never combined with other trace domains), per-dispatch averages, FETCH_SIZE doubled (gfx950 reports half the
    bytes of a wide read; for 16-byte gathers that makes the figure an upper bound).  Each pass is a child process running
    this script with --traffic-child (the same workload, 12 serial passes)."""
    import shutil
    import sqlite3

    if shutil.which("rocprofv3") is None:
        return None
    kib = {}
    with tempfile.TemporaryDirectory(prefix="gffx_pmc_", dir="/tmp") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", out, "-o", "run", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--traffic-child", "--mode", args.mode, "--strategy", args.strategy,
                   "--out", args.out, "--offsets", args.offsets, "--queries-per-gpu", str(args.queries_per_gpu), "--inflight", str(group)]
            if args.region_width:
                cmd += ["--region-width", str(args.region_width[0]), str(args.region_width[1])]
            try:
                env = dict(os.environ, TMPDIR="/tmp", GFFX_HIP_GROUP="1")  # (the child's launches: one group of `group` batches, serial)
                if block_threads:  # (the child's serial passes must run the kernel variant of the timed region)
                    env["GFFX_HIP_WIN_THREADS"] = str(block_threads)
                if blocks:
                    env["GFFX_HIP_FUSED_BLOCKS"] = str(blocks)
                r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=240)
            except Exception as exc:
                return {"error": repr(exc)[:200]}
            db = None
            for dirpath, _, files in os.walk(out):
                for f in files:
                    if f.endswith("_results.db"):
                        db = os.path.join(dirpath, f)
            if r.returncode != 0 or db is None:
                return {"error": "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, (r.stderr or "")[-200:])}
            # per kernel AND grid: the child's first launches serve one batch each (they size the buffers), the measured ones a group
            # of `group` batches -- a different grid of the same kernel; the variant that moves the most bytes in all is the measured one
            rows = sqlite3.connect(db).execute(
                "select kernel_name, count(*), avg(value), grid_size_x from counters_collection where counter_name = ? "
                "group by kernel_name, grid_size_x", (counter,)).fetchall()
            rows = [x for x in rows if "k_join_" in x[0] or "k_tile_join" in x[0] or "k_partition" in x[0]]
            if not rows:
                return {"error": "no dispatch of a join kernel in the %s pass" % counter}
            name, n, avg, grid_x = max(rows, key=lambda x: x[1] * x[2])
            kib[counter] = {"kernel": name.split("<")[0].replace("void ", ""), "dispatches": n, "avg_KiB": avg, "grid_threads": grid_x}
    b = 2.0 * kib["FETCH_SIZE"]["avg_KiB"] * 1024 + kib["WRITE_SIZE"]["avg_KiB"] * 1024
    return {"hbm_bytes_per_launch": b, "kernel": kib["FETCH_SIZE"]["kernel"], "FETCH_SIZE_KiB": kib["FETCH_SIZE"]["avg_KiB"],
            "WRITE_SIZE_KiB": kib["WRITE_SIZE"]["avg_KiB"], "dispatches": kib["FETCH_SIZE"]["dispatches"], "grid_threads": kib["FETCH_SIZE"]["grid_threads"],
            "batches_per_launch": group,
            "source": "measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate child "
                      "passes; bytes = 2 x FETCH_SIZE + WRITE_SIZE (the doubling is calibrated for wide streaming reads: an "
                      "upper bound for this kernel's 16-byte index gathers)"}


def to_dev(torch, regions, dev):
    t = torch.from_numpy(np.ascontiguousarray(regions).view(np.int32))  # u32 bit patterns
    return tuple(t[:, k].contiguous().to(dev) for k in range(3))


def xfer_leg(engine, torch, ix, regions, mode, pairs, reps=8):
    """T_xfer: regions start in pinned host memory, counts + root_fids end in pinned host memory.  Two batches
    double-buffered: the copies of one overlap the kernel of the other (they run on the batches' own streams)."""
    nq = len(regions)
    L = engine.lib()
    pin_in = torch.from_numpy(np.ascontiguousarray(regions).view(np.int32)).pin_memory()
    cap = pairs + pairs // 8 + 1024
    pin_counts = [torch.empty(nq, dtype=torch.int32).pin_memory() for _ in range(2)]
    pin_fids = [torch.empty(cap, dtype=torch.int32).pin_memory() for _ in range(2)]
    u32p = ctypes.POINTER(ctypes.c_uint32)
    bs = [engine.QueryBatch(ix, nq) for _ in range(2)]
    flags = engine.OUT_FIDS | engine.OUT_SEGBASE

    def one(k):
        b = bs[k]
        engine.check(L.gffx_hip_batch_set_regions_host(b._h, ctypes.cast(pin_in.data_ptr(), u32p), nq))
        b.run(mode, False, flags, 0)

    def collect(k):
        b = bs[k]
        b.wait()
        engine.check(L.gffx_hip_batch_copy_counts(b._h, ctypes.cast(pin_counts[k].data_ptr(), u32p)))
        engine.check(L.gffx_hip_batch_copy_fids(b._h, ctypes.cast(pin_fids[k].data_ptr(), u32p)))
        return b.total_hits

    for k in range(2):
        bs[k].reserve_hits(cap)
        one(k)
        collect(k)
    t0 = time.perf_counter()
    one(0)
    for i in range(1, reps):
        one(i & 1)
        got = collect((i - 1) & 1)
    got = collect((reps - 1) & 1)
    dt = (time.perf_counter() - t0) / reps
    assert got == pairs
    for b in bs:
        b.close()
    moved = 12.0 * nq + 4.0 * nq + 4.0 * pairs
    return {"ms_per_batch": 1e3 * dt, "value": nq / dt, "unit": "queries/s", "regions": nq,
            "pcie_GBps": moved / dt / 1e9,
            "note": "pinned host regions -> HBM, pass, counts + root_fids -> pinned host; two batches double-buffered; "
                    "never the headline value (PCIe Gen5 x16 bounds it at ~63 GB/s)"}


def _same_file(a, b):
    """byte equality of two outputs (b None: just 'a exists and is not empty')"""
    try:
        if b is None:
            return os.path.getsize(a) > 0
        return os.path.getsize(a) == os.path.getsize(b) and open(a, 'rb').read() == open(b, 'rb').read()
    except OSError:
        return False


def e2e_leg(synth, roots, regions, tmp):
    """T_e2e: the product CLI, end to end, on text inputs: GENCODE-shaped GFF3 (~3.5 M lines) x the 1 M-row BED."""
    gff, bed = os.path.join(tmp, "anno.gff"), os.path.join(tmp, "q.bed")
    n_lines = synth.write_gff3_fast(gff, roots)
    synth.write_bed_fast(bed, regions, roots["names"])
    G = os.path.join(ROOT, "gffx_amd", "bin", "gffx")
    t0 = time.perf_counter()
    subprocess.run([G, "index", "-i", gff], check=True, capture_output=True)
    t_index = time.perf_counter() - t0
    out = {"gff_lines": n_lines, "gff_MB": os.path.getsize(gff) / 1e6, "bed_rows": len(regions), "index_s": t_index, "runs": {}}
    for name, extra in (("intersect", []), ("intersect -e", ["-e"])):
        best, stages, size = None, "", 0
        for _ in range(2):  # (second run: page cache warm)
            t0 = time.perf_counter()
            sj = os.path.join(tmp, "stats.json")
            r = subprocess.run([G, "intersect", "--stats-json", sj, "-i", gff, "-b", bed, "-o", os.path.join(tmp, "out.gff")] + extra,
                               capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": r.stderr[-300:]}
            if best is None or dt < best:
                best = dt
                try:  # the run's own stage timers and counts (gffx intersect --stats-json), not scraped from stderr
                    stages = json.load(open(sj))
                except Exception as exc:
                    stages = {"error": repr(exc)[:200]}
                size = os.path.getsize(os.path.join(tmp, "out.gff"))
        out["runs"][name] = {"wall_s": best, "regions_per_s": len(regions) / best, "output_MB": size / 1e6, "stages": stages}
    # ---- the CPU path on the SAME files, next to it: the oracle's intersect_run (commands/intersect.rs:541-655 restated in C:
    # load the index, parse the BED, Join A serial, then merged blocks with -e, or the per-line scan of every line of the hit
    # blocks against every region of its seqid).  One thread, as the restatement is written (the reference runs the per-line
    # scan under rayon: cpu_baseline.join_b_allcore has the all-core figure of that stage).  `-e`: the full 1 M-row BED.
    # Per-line mode is O(lines x regions per seqid): on the full BED it would run for hours, so it is timed -- GPU CLI and
    # oracle -- on the first `small` rows of the same BED.
    small = 20000
    bed_small = os.path.join(tmp, "q_small.bed")
    with open(bed) as f, open(bed_small, "w") as g:
        for i, ln in enumerate(f):
            if i >= small:
                break
            g.write(ln)
    t0 = time.perf_counter()
    r = subprocess.run([G, "intersect", "-i", gff, "-b", bed_small, "-o", os.path.join(tmp, "out_small.gff")], capture_output=True, text=True)
    gpu_small = time.perf_counter() - t0
    if r.returncode != 0:
        return {"error": r.stderr[-300:]}
    code = ("import sys, time, json; sys.path.insert(0, %r); from oracle import binding as ob; t0 = time.perf_counter(); "
            "rc, err = ob.intersect_run(sys.argv[1], sys.argv[3], bed=sys.argv[2], mode=2, entire_group=sys.argv[4] == '1'); "
            "print(json.dumps({'rc': rc, 'err': err[:200], 'wall_s': time.perf_counter() - t0}))" % ROOT)
    for name, b_, eg, key in (("intersect -e", bed, "1", "out_o.gff"), ("intersect", bed_small, "0", "out_o_small.gff")):
        try:
            rr = subprocess.run([sys.executable, "-c", code, gff, b_, os.path.join(tmp, key), eg], capture_output=True, text=True, timeout=900)
            oj = json.loads(rr.stdout.strip().splitlines()[-1])
        except Exception as exc:
            oj = {"error": repr(exc)[:200]}
        if name == "intersect -e":
            out["runs"][name]["cpu_oracle_wall_s"] = oj.get("wall_s")
            out["runs"][name]["cpu_oracle"] = dict(oj, cores=1, kind="port", bed_rows=len(regions),
                                                   outputs_identical=_same_file(os.path.join(tmp, key), os.path.join(tmp, "out.gff")))
        else:
            same = _same_file(os.path.join(tmp, key), os.path.join(tmp, "out_small.gff"))
            out["runs"]["intersect, first %d rows" % small] = {
                "wall_s": gpu_small, "bed_rows": small, "cpu_oracle_wall_s": oj.get("wall_s"), "cpu_oracle": dict(oj, cores=1, kind="port"),
                "outputs_identical": same,
                "note": "per-line mode: the oracle's scan is O(lines of the hit blocks x regions of the seqid), so both sides run "
                        "the first %d rows of the BED; the full-size GPU run is runs['intersect']" % small}
    # BASELINE configs[3]'s size through the streaming CLI: 100 M rows (2.4 GB of BED text), --entire-group and per-line mode,
    # 64 host threads (the reference's default of 12 is the first table column of DESIGN 5.2); best of 2, page cache warm
    try:
        big_n = 100_000_000
        if shutil.disk_usage(tmp).free > (6 << 30):
            big = os.path.join(tmp, "q100m.bed")
            base_rows = synth.synth_bed(big_n // 10, seed=1003)  # (ten shifted copies of 10 M seeded rows: numpy needs ~1 min
            shifted = []                                          #  for 100 M fresh ones, which would double the run)
            for k in range(10):
                part = base_rows.copy()
                part[:, 1:] += np.uint32(13 * k)
                shifted.append(part)
            synth.write_bed_fast(big, np.concatenate(shifted), roots["names"])
            del shifted, base_rows
            for name, extra in (("intersect -e, 100 M rows, -t 64", ["-e", "-t", "64"]), ("intersect, 100 M rows, -t 64", ["-t", "64"])):
                best = None
                for _ in range(2):
                    t0 = time.perf_counter()
                    r = subprocess.run([G, "intersect", "-i", gff, "-b", big, "-o", os.path.join(tmp, "out.gff")] + extra,
                                       capture_output=True, text=True)
                    dt = time.perf_counter() - t0
                    if r.returncode != 0:
                        raise RuntimeError(r.stderr[-300:])
                    best = dt if best is None else min(best, dt)
                out["runs"][name] = {"wall_s": best, "regions_per_s": big_n / best, "bed_rows": big_n,
                                     "bed_GB": os.path.getsize(big) / 1e9,
                                     "output_MB": os.path.getsize(os.path.join(tmp, "out.gff")) / 1e6}
            os.remove(big)
    except Exception as exc:  # (the small runs above stand on their own)
        out["runs"]["100 M rows"] = {"error": repr(exc)[:300]}
    return out


def main():
    args = parse_args()
    rank, world, local_rank, _ = rank_placement(os.environ, 1)
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # invoked like the 1-GPU run: start the ranks as fresh child processes (this process has not touched the GPU and
            # never will -- nothing is exec'ed or forked from a process that initialised HIP) and pass their output through
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            sys.exit(subprocess.call(cmd, env=env))
        args.gpus = world
    if args.exchange is None:
        args.exchange = "final-timed" if (args.scaling == "strong" and world > 1) else "final"
    # stdout carries ONE thing: rank 0's JSON line.  Everything else a rank prints -- gloo's and RCCL's C++ banners included,
    # which go to file descriptor 1 behind Python's back -- is sent to stderr: fd 1 becomes a copy of fd 2 from here on, and the
    # line is written to the descriptor that was stdout when the process started.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from gffx_amd import engine, shard, synth

    if not torch.cuda.is_available() or engine.device_count() < 1:
        print("bench.py: no MI355X visible; the engine has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    # GFFX_BENCH_BACKEND=gloo lets the N>1 plumbing be exercised on a 1-GPU box (all ranks share device 0); the driver's
    # runs use nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("GFFX_BENCH_BACKEND", "nccl")
    dev_index = rank_placement(os.environ, torch.cuda.device_count())[3]
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    coll_dev = dev if backend == "nccl" else None  # gloo gathers CPU tensors

    mode = {"contained": 0, "contains_region": 1, "overlap": 2}[args.mode]
    strategy = {"auto": 0, "direct": 1, "sorted": 2, "fused": 3, "windows": 5}[args.strategy]
    out_flags = {"counts": engine.OUT_COUNTS, "fids": engine.OUT_FIDS, "triples": engine.OUT_TRIPLES}[args.out]
    if args.strategy != "direct" and args.out != "counts":
        offs = args.offsets if args.strategy in ("auto", "windows") or args.offsets != "seg" else "u64"
        out_flags |= {"seg": engine.OUT_SEGBASE, "u32": engine.OUT_OFFSETS32, "u64": engine.OUT_OFFSETS, "none": 0}[offs]
    if args.strategy == "sorted":
        out_flags |= engine.OUT_EMIT_ORDER
    out_b = {"counts": 0.0, "fids": 4.0, "triples": 12.0}[args.out]

    # ---- synthetic inputs (identical on every rank; each rank keeps its shard)
    roots = synth.gencode_like_roots(63000, seed=42)
    n_chr = len(roots["chr_offsets"]) - 1
    strong = args.scaling == "strong" and world > 1
    regions, nq_global, cfg = bench_regions(synth, shard, n_chr, world, rank, args.scaling, args.queries_per_gpu, args.strong_total,
                                            args.region_width, args.wide_every)
    if args.presort == "chr_end":
        regions = np.ascontiguousarray(regions[np.lexsort((regions[:, 2], regions[:, 0]))])
    elif args.presort == "chr_start":  # (as BED files usually are; the `sorted_bed` leg's order)
        regions = np.ascontiguousarray(regions[np.lexsort((regions[:, 1], regions[:, 0]))])
    nq = len(regions)

    ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"],
                                         roots["names"], device=dev_index)
    cols = to_dev(torch, regions, dev)  # regions resident in HBM as SoA u32 (torch owns the memory; the engine borrows the pointers)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.traffic_child:  # (under rocprofv3 --pmc: the workload, 12 serial launches, nothing else)
        # --inflight here = the batches ONE launch of the timed region serves (the parent passes the group's size and GFFX_HIP_GROUP=1:
        # one group, one stream, launch after launch)
        one = Pass(engine, ix, cols, nq, args.inflight, mode, out_flags, strategy)
        one.size_and_warm(0)
        one.run_n(12 * len(one.batches))
        one.sync()
        one.close()
        return
    run = Pass(engine, ix, cols, nq, args.inflight, mode, out_flags, strategy)
    pairs = run.size_and_warm(args.warmup)
    if world > 1:  # the collective's first call sets up its channels: not part of the job's steady state
        shard.allgather_hit_counts(nq, pairs, device=coll_dev)

    # ---- the timed region: exactly K steps (= K x passes-per-step passes) between barrier + synchronize, MAX over ranks;
    # measured --repeats times, the MEDIAN repeat is the line's value
    n_passes = args.steps * args.passes_per_step

    def timed_once():
        if world > 1 and args.exchange != "final":
            run.run_n(2 * len(run.batches))
            run.sync()
            barrier()
            t0 = time.perf_counter()
            if args.exchange == "per-step":
                for _ in range(args.steps):
                    run.run_n(args.passes_per_step)
                    run.batches[0].wait()
                    shard.allgather_hit_counts(nq, pairs, device=coll_dev)
            else:
                run.run_n(n_passes)
            run.sync()
            if args.exchange == "final-timed":  # the job's one exchange step, inside the timed region
                run.batches[0].wait()
                shard.allgather_hit_counts(nq, run.batches[0].total_hits, device=coll_dev)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        else:
            el = run.timed(n_passes, barrier, torch)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    all_elapsed = [timed_once() for _ in range(max(1, args.repeats))]
    elapsed = float(np.median(all_elapsed))
    barrier()
    run.check(pairs)
    exchange_ms = None
    if world > 1:
        tx = time.perf_counter()
        counts = shard.allgather_hit_counts(nq, pairs, device=coll_dev)
        torch.cuda.synchronize()
        exchange_ms = 1e3 * (time.perf_counter() - tx)
        nq_total, pairs_total = int(counts[:, 0].sum()), int(counts[:, 1].sum())
    else:
        nq_total, pairs_total = nq, pairs

    result = None
    if rank == 0:
        # the kernel variant the timed region ran (the last pass of batch 0 in it): the serial passes measured for `roofline`
        # and the PMC child are forced to the same block width
        timed_threads = run.batches[0].block_threads or None
        timed_blocks = run.batches[0].block_count or None
        plan_groups, plan_largest, plan_streams = engine.batches_plan(run.batches)
        group_launch = None
        if plan_groups:
            # the timed region's launches serve `plan_largest` batches each: THAT launch is what the roofline is quoted for -- serial
            # launches of one group, back to back between one pair of HIP events on the stream they run on
            gb = run.batches[:plan_largest]
            for bb in gb:
                bb.set_option("WIN_THREADS", timed_threads or 0)
            us_launch, grouped = engine.timed_group_runs(gb, mode, False, out_flags, strategy, max(5, min(args.steps, 30)))
            for bb in gb:
                bb.set_option("WIN_THREADS", 0)
            kern = {"k_join_pairs": {"avg_us": us_launch, "launches_per_step": args.passes_per_step / plan_largest}}
            run.pass_us_one_event_pair = us_launch
            group_launch = {"batches_per_launch": plan_largest, "groups": plan_groups, "streams": plan_streams, "grouped": grouped,
                            "launch_us": us_launch, "us_per_pass_inside_the_launch": us_launch / plan_largest,
                            "launches_per_step": args.passes_per_step / plan_largest,
                            "launch_us_x_launches_per_step_ms": 1e-3 * us_launch * args.passes_per_step / plan_largest,
                            "note": "ms_per_step below launch_us x launches_per_step = what two streams buy: one group's drain under the next one's ramp"}
        else:
            kern = run.kernel_us(max(5, min(args.steps, 30)), timed_threads, timed_blocks)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if world > 1:
            traffic = {"hbm_bytes_per_launch": None, "note": "not measured in this run (the PMC passes are child runs of the 1-GPU "
                       "workload; a multi-GPU line does not copy them)"}
        elif os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = {"hbm_bytes_per_launch": tj.get("hbm_bytes_per_pass"), "kernel": tj.get("kernel"),
                           "measured_at_commit": tj.get("commit"), "source": "profiles/traffic_latest.json (rocprofv3 --pmc "
                           "FETCH_SIZE / WRITE_SIZE passes; NOT measured by this run)"}
            except Exception:
                traffic = None
        if world == 1 and not args.no_traffic and not args.quick:
            live = measure_traffic(args, timed_threads, None if plan_groups else timed_blocks, plan_largest if plan_groups else 1)
            if live and "error" not in live:
                traffic = live
            elif live and traffic is not None:
                traffic["live_measurement_error"] = live["error"]
        result = {
            "metric": "BED overlap queries/sec vs GRCh38-scale GFF index at 1/2/4/8 MI355X",  # BASELINE.json
            "value": nq_total * n_passes / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "passes_per_step": args.passes_per_step,
            "us_per_pass": 1e6 * elapsed / n_passes,
            "repeats": {"n": len(all_elapsed), "timed_region_ms": [1e3 * x for x in all_elapsed],
                        "value_min": nq_total * n_passes / max(all_elapsed), "value_max": nq_total * n_passes / min(all_elapsed),
                        "note": "every repeat = exactly K steps between barrier + synchronize; value is the median repeat"},
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE %s: %d synthetic BED regions %s (seed %d) x GENCODE/GRCh38-shaped index (25 seqids, %d root "
                            "genes, seed 42), --%s, regions resident in HBM as u32 SoA; one step = %d passes over the resident batch "
                            "(a pass = the kernel's work over the batch's regions into one of the result buffers in flight; from four "
                            "buffers on ONE launch serves a group of them: group_launch; every buffer in flight has its own copy of the regions)"
                            % (cfg, nq_global if strong else args.queries_per_gpu, "in total" if strong else "per GPU", 1003 if strong else 1001,
                               ix.n_roots, args.mode, args.passes_per_step),
                "regions_total": nq_total,
                "kept_pairs_total": pairs_total,
                "pairs_per_region": pairs_total / max(nq_total, 1),
                "outputs": "per-region counts%s (input order) + %s, segments in round order"
                           % ({"seg": " and one u64 segment base per group of 256 regions", "u32": " and u32 segment offsets",
                               "u64": " and u64 segment offsets", "none": ""}[args.offsets]
                              if args.out != "counts" else "", args.out),
                "strategy": args.strategy,
                "presort": args.presort,
                "batches_in_flight": len(run.batches),
                "sharding": ("chromosome buckets, LPT with splitting; index replicated; all-gather of hit counts "
                             + {"per-step": "after every step (inside the timed region)",
                                "final-timed": "once per job, inside the timed region",
                                "final": "once per job, after the K timed steps (exchange_ms)"}[args.exchange])
                            if world > 1 else "none (1 GPU)",
                "exchange_ms": exchange_ms,
                # GFFX_HIP_* tuning knobs that were NOT at their defaults for this run ({}: all defaults), as the index / the timed
                # batches read them when they were created
                "knobs": {**ix.options(), **run.batches[0].options()},
            },
            "roofline": roofline_obj(kern, nq * (plan_largest if plan_groups else 1), pairs * (plan_largest if plan_groups else 1), out_b,
                                     "achieved = (12 B region + 4 B count + 4 B x pairs/region) x the regions of ONE LAUNCH of the timed region / "
                                     "its duration: serial launches back to back between one pair of HIP events on the stream they run on "
                                     "(rank 0), at the block width of the timed region's launches.  With four batches or more in flight a launch "
                                     "serves a group of batches (group_launch: regions_per_launch = batches_per_launch x the batch; traffic "
                                     "is per launch as well); with fewer, one batch (round 5's launches in flight: forced to that region's grid).  "
                                     "The resident batch is re-read by every pass: at 1 M regions (12 MB) the input stream is served by the 256 MB "
                                     "Infinity Cache, not by HBM (immaterial at this fraction of the roofline, but it is not a cold-HBM figure)", traffic,
                                     run.pass_us_one_event_pair, timed_threads, timed_blocks),
            "group_launch": group_launch,
        }
    if world == 1:
        # ---- strictly serial passes (one batch, one stream): what the committed rocprofv3 kernel stats show
        ser = Pass(engine, ix, cols, nq, 1, mode, out_flags, strategy)
        ser.size_and_warm(1)
        el = ser.timed(n_passes, barrier, torch)
        ser.check(pairs)
        ks = ser.kernel_us(max(5, min(args.steps, 30)))  # (nothing forced: what the engine picks for a pass that runs alone)
        result["serial"] = {"ms_per_step": 1e3 * el / args.steps, "us_per_pass": 1e6 * el / n_passes, "value": nq * n_passes / el, "unit": "queries/s",
                            "batches_in_flight": 1,
                            "roofline": roofline_obj(ks, nq, pairs, out_b, "one batch, passes strictly one after the other: the engine "
                                                     "takes 1024-thread blocks (one per CU) for a pass of 500 000 regions or more that runs alone",
                                                     None, ser.pass_us_one_event_pair, ser.block_threads)}
        ser.close()
    if world == 1 and not args.quick:
        # ---- the pass the CLI runs: root bitmap only (commands/intersect.rs:598-615 needs the unique roots, nothing else)
        cli = {}
        bm = Pass(engine, ix, cols, nq, 1, mode, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, strategy)  # (the CLI's flags)
        bm.size_and_warm(1)
        kb = bm.kernel_us(10)
        cli["1m"] = {"bitmap_pass_us": bm.pass_us_one_event_pair, "kernels": kb, "fids_pass_us": result["roofline"]["pass_kernel_us"],
                     "note": "both: serial passes back to back between one pair of HIP events; kernels{} with a pair per launch"}
        bm.close()
        # ---- 10 M regions (seed 1002): roofline of the same pass, and the CLI's pass
        reg10 = synth.synth_bed(10_000_000, seed=1002)
        cols10 = to_dev(torch, reg10, dev)
        p10 = Pass(engine, ix, cols10, len(reg10), 1, mode, out_flags, strategy)
        pairs10 = p10.size_and_warm(1)
        k10 = p10.kernel_us(10)
        result["roofline_10m"] = roofline_obj(k10, len(reg10), pairs10, out_b, "10 M synthetic BED regions (seed 1002), same pass",
                                              None, p10.pass_us_one_event_pair, p10.block_threads)
        p10.close()
        # ... and in --contained mode (BASELINE configs[2]'s mode)
        p10c = Pass(engine, ix, cols10, len(reg10), 1, 0, out_flags, strategy)
        pairs10c = p10c.size_and_warm(1)
        k10c = p10c.kernel_us(10)
        result["roofline_10m_contained"] = roofline_obj(k10c, len(reg10), pairs10c, out_b, "the same 10 M regions, --contained "
                                                        "(BASELINE configs[2]'s mode)", None, p10c.pass_us_one_event_pair, p10c.block_threads)
        p10c.close()
        bm10 = Pass(engine, ix, cols10, len(reg10), 1, mode, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, strategy)
        bm10.size_and_warm(1)
        kb10 = bm10.kernel_us(5)
        cli["10m"] = {"bitmap_pass_us": bm10.pass_us_one_event_pair, "kernels": kb10,
                      "fids_pass_us": result["roofline_10m"]["pass_kernel_us"]}
        bm10.close()
        result["cli_pass"] = cli
        del cols10, reg10
        # ---- cold input: three distinct resident 12 M-region batches (432 MB of regions, more than the 256 MB Infinity Cache),
        # one QueryBatch pointed at them in turn, pass after pass on one stream: no pass finds its regions in any cache
        result["roofline_cold"] = cold_leg(engine, synth, torch, ix, dev, mode, out_flags, strategy, out_b)
        # ---- regions wider than one window line answers (wmax = 16 Ki): after the first waited pass AUTO runs the batch's pair
        # passes on the WIDE form of k_join_pairs (two lines and two rank records per region; round 3: the sweep kernel)
        regw = synth.synth_bed(nq, seed=1004, width=(100, 200000))
        colsw = to_dev(torch, regw, dev)
        pw = Pass(engine, ix, colsw, nq, 1, mode, out_flags & ~engine.OUT_SEGBASE | engine.OUT_OFFSETS, 0)
        pairsw = pw.size_and_warm(2)  # (the second waited pass is the one AUTO re-routes)
        kw = pw.kernel_us(10)
        result["wide_regions"] = roofline_obj(kw, nq, pairsw, out_b, "%d regions of width U[100, 200000] (seed 1004), AUTO, u64 offsets: "
                                              "most regions are wider than one window line answers, the batch's passes run on the "
                                              "wide form of k_join_pairs" % nq, None, pw.pass_us_one_event_pair, pw.block_threads)
        result["wide_regions"]["wide_form"] = bool(pw.batches[0].wide_form)
        pw.close()
        # ... and the root pass over them (what the CLI runs per chunk): the wide form of k_join_roots
        bmw = Pass(engine, ix, colsw, nq, 1, mode, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, 0)
        bmw.size_and_warm(2)
        kbw = bmw.kernel_us(10)
        result["wide_regions"]["root_pass"] = {"bitmap_pass_us": bmw.pass_us_one_event_pair, "kernels": kbw,
                                               "wide_form": bool(bmw.batches[0].wide_form)}
        bmw.close()
        # ... and --contained over the same wide regions (round 5: a wide lane's run of roots filtered by their ends, no sweep kernel)
        pwc = Pass(engine, ix, colsw, nq, 1, 0, out_flags & ~engine.OUT_SEGBASE | engine.OUT_OFFSETS, 0)
        pairswc = pwc.size_and_warm(2)
        kwc = pwc.kernel_us(10)
        result["wide_regions_contained"] = roofline_obj(kwc, nq, pairswc, out_b, "the same regions, --contained, AUTO: the mixed form of k_join_pairs "
                                                        "(round 4: the sweep kernel k_join_fused)", None, pwc.pass_us_one_event_pair, pwc.block_threads)
        result["wide_regions_contained"]["mixed_form"] = bool(pwc.batches[0].wide_form)
        pwc.batches[0].set_option("WIN_WIDE", 0)  # round 4's answer: the sweep kernel
        pwc.kernel_us(10)
        result["wide_regions_contained"]["sweep_kernel_pass_us"] = pwc.pass_us_one_event_pair
        pwc.close()
        del colsw, regw
        # ---- a MIXED batch: the headline's regions with every tenth row replaced by an SV-sized one (width U[20 000, 2 000 000])
        # -- below the eighth at which AUTO used to leave the narrow form: the mixed form serves every region its own way
        regm = widen_every(regions, 10)
        colsm = to_dev(torch, regm, dev)
        pm = Pass(engine, ix, colsm, nq, 1, mode, out_flags, 0)
        pairsm = pm.size_and_warm(2)  # (regions on the device carry no width sample: the first waited pass tells AUTO)
        km = pm.kernel_us(10)
        result["mixed_widths"] = roofline_obj(km, nq, pairsm, out_b, "the headline's %d regions, every tenth one widened to U[20000, 2000000] bases "
                                              "(seed 1005), AUTO: the mixed form of k_join_pairs -- a lane per region, narrow regions from one "
                                              "line, wide ones from two lines and two ranks" % nq, None, pm.pass_us_one_event_pair, pm.block_threads)
        result["mixed_widths"]["mixed_form"] = bool(pm.batches[0].wide_form)
        result["mixed_widths"]["narrow_pass_us"] = result["serial"]["roofline"]["pass_kernel_us"] if "serial" in result else None
        pm.batches[0].set_option("WIN_WIDE", 0)  # the same batch on the narrow form (every wide row an out-of-line sweep): round 4's answer
        pm.kernel_us(10)
        result["mixed_widths"]["narrow_form_pass_us"] = pm.pass_us_one_event_pair
        pm.close()
        pmc = Pass(engine, ix, colsm, nq, 1, 0, out_flags, 0)  # --contained over the mixed batch
        pairsmc = pmc.size_and_warm(2)
        kmc = pmc.kernel_us(10)
        result["mixed_widths"]["contained"] = roofline_obj(kmc, nq, pairsmc, out_b, "the mixed batch, --contained, AUTO", None, pmc.pass_us_one_event_pair,
                                                           pmc.block_threads)
        result["mixed_widths"]["contained"]["mixed_form"] = bool(pmc.batches[0].wide_form)
        pmc.batches[0].set_option("WIN_WIDE", 0)
        pmc.kernel_us(10)
        result["mixed_widths"]["contained"]["narrow_form_pass_us"] = pmc.pass_us_one_event_pair
        pmc.close()
        pmr = Pass(engine, ix, colsm, nq, 1, 1, out_flags, 0)  # --contains-region over the mixed batch (a wide row: the roots over its first base that reach its end)
        pairsmr = pmr.size_and_warm(2)
        kmr = pmr.kernel_us(10)
        result["mixed_widths"]["contains_region"] = roofline_obj(kmr, nq, pairsmr, out_b, "the mixed batch, --contains-region, AUTO", None,
                                                                 pmr.pass_us_one_event_pair, pmr.block_threads)
        result["mixed_widths"]["contains_region"]["mixed_form"] = bool(pmr.batches[0].wide_form)
        pmr.batches[0].set_option("WIN_WIDE", 0)
        pmr.kernel_us(10)
        result["mixed_widths"]["contains_region"]["narrow_form_pass_us"] = pmr.pass_us_one_event_pair
        pmr.close()
        bmm = Pass(engine, ix, colsm, nq, 1, mode, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, 0)
        bmm.size_and_warm(2)
        kbm = bmm.kernel_us(10)
        result["mixed_widths"]["root_pass"] = {"bitmap_pass_us": bmm.pass_us_one_event_pair, "kernels": kbm, "mixed_form": bool(bmm.batches[0].wide_form)}
        bmm.close()
        del colsm, regm
        # ---- a BED file sorted by (seqid, start), as most are
        regs = np.ascontiguousarray(regions[np.lexsort((regions[:, 1], regions[:, 0]))])
        colss = to_dev(torch, regs, dev)
        ps = Pass(engine, ix, colss, nq, 1, mode, out_flags, strategy)
        pairss = ps.size_and_warm(1)
        ks_ = ps.kernel_us(10)
        result["sorted_bed"] = roofline_obj(ks_, nq, pairss, out_b, "the headline's regions sorted by (seqid, start)", None,
                                            ps.pass_us_one_event_pair, ps.block_threads)
        ps.close()
        if group_launch and group_launch.get("grouped"):
            # ... and the headline's configuration on sorted batches (every batch its own copy): the launch that serves a group, alone, at both
            # block widths; then as many batches in flight as the timed region has, one timed region
            try:
                n_sg = group_launch["batches_per_launch"]
                pg = Pass(engine, ix, colss, nq, len(run.batches), mode, out_flags, strategy)
                pg.size_and_warm(1)
                alg_sg = n_sg * (12.0 * nq + 4.0 * nq + out_b * pairss)
                sg = {"batches_per_launch": n_sg, "algorithmic_bytes_per_launch": alg_sg}
                for th in (512, 1024):
                    for bb in pg.batches[:n_sg]:
                        bb.set_option("WIN_THREADS", th)
                    us_sg, grouped_sg = engine.timed_group_runs(pg.batches[:n_sg], mode, False, out_flags, strategy, 20)
                    sg["launch_us_%d_threads" % th] = us_sg
                    sg["frac_%d_threads" % th] = alg_sg / (us_sg * 1e-6) / 8e12
                    sg["grouped"] = bool(grouped_sg)
                for bb in pg.batches:
                    bb.set_option("WIN_THREADS", 0)
                n_passes = args.steps * args.passes_per_step
                dt = pg.timed(n_passes, lambda: None, torch)
                pg.check(pairss)
                sg["in_flight"] = {"batches_in_flight": len(pg.batches), "passes": n_passes, "timed_region_ms": 1e3 * dt,
                                   "value": nq * n_passes / dt, "unit": "queries/s", "block_threads": pg.batches[0].block_threads}
                sg["note"] = ("serial launches, each serving this many sorted batches, back to back between one pair of HIP events, at 512-thread blocks "
                              "(what the timed region's launches use: two groups co-resident) and at 1024; in_flight: the headline's timed region on sorted "
                              "batches, one repeat")
                result["sorted_bed"]["group_launch"] = sg
                pg.close()
            except Exception as exc:  # (an auxiliary leg)
                result["sorted_bed"]["group_launch"] = {"error": repr(exc)[:300]}
        del colss, regs
        # ---- transfers included, and the product CLI end to end
        result["t_xfer"] = xfer_leg(engine, torch, ix, regions, mode, pairs)
        try:
            with tempfile.TemporaryDirectory(prefix="gffx_bench_") as tmp:
                result["t_e2e"] = e2e_leg(synth, roots, regions, tmp)
        except Exception as exc:  # the headline must not die with an auxiliary leg
            result["t_e2e"] = {"error": repr(exc)[:300]}
        result["join_b"] = join_b_leg(engine, synth, roots, regions, mode)
        result["depth"] = depth_leg(engine, synth, roots, ix, cols, nq)
    if rank == 0 and not args.no_cpu_baseline:
        # (N > 1: rank 0 times the same 1-thread leg on ITS shard, after the timed region and the exchange)
        cb = cpu_baseline(roots, regions, mode, args.cpu_seconds)
        if world > 1:
            cb["sample"] += "; rank 0's shard of the %d-GPU batch" % world
        if not args.quick and world == 1:
            cb["join_a_allcore"] = child_json([os.path.join(ROOT, "tools", "cpu_allcore.py"), str(nq), str(mode), "4"])
            cb["join_b_allcore"] = child_json([os.path.join(ROOT, "tools", "cpu_joinb_allcore.py"), str(nq), str(mode), "6"])
        result["cpu_baseline"] = cb
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cold_leg(engine, synth, torch, ix, dev, mode, out_flags, strategy, out_b, n_each=12_000_000, n_sets=3, rounds=3):
    sets, pairs = [], []
    for k in range(n_sets):
        sets.append(to_dev(torch, synth.synth_bed(n_each, seed=1005 + k), dev))
    torch.cuda.synchronize()
    b = engine.QueryBatch(ix, n_each)
    for c in sets:  # size the pair buffers, learn every set's pair count
        b.set_regions_device(c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), n_each, keep=c)
        b.run(mode, False, out_flags, strategy)
        b.wait()
        pairs.append(b.total_hits)
    b.reserve_hits(max(pairs) + max(pairs) // 8 + 1024)
    b.set_profiling(True)
    b.reset_profile()
    for _ in range(rounds):
        for c in sets:
            b.set_regions_device(c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), n_each, keep=c)
            b.run(mode, False, out_flags, strategy)
    b.sync()
    b.set_profiling(False)
    kern = {}
    for kid, name in engine.KERNEL_NAMES.items():
        ms, n = b.kernel_ms(kid)
        if n:
            kern[name] = {"avg_us": 1e3 * ms / n, "launches_per_step": n / (rounds * n_sets)}
    threads = b.block_threads
    b.wait()
    b.close()
    obj = roofline_obj(kern, n_each, sum(pairs) / len(pairs), out_b,
                       "%d passes over %d distinct resident batches of %d regions (seeds 1005..), %d MB of regions in rotation -- more "
                       "than the 256 MB Infinity Cache -- on ONE stream: every pass streams its regions from HBM.  Durations: a HIP "
                       "event pair per launch (~2.5 us of event cost in each)" % (rounds * n_sets, n_sets, n_each, 12 * n_each * n_sets // 1000000),
                       None, None, threads)
    obj["input_MB_in_rotation"] = 12.0 * n_each * n_sets / 1e6
    return obj


def depth_leg(engine, synth, roots, ix, cols, nq):
    """`gffx depth` (BASELINE configs[4]'s command, reference semantics: per-feature-ID region counts): k_depth_regions over
    the pairs of an Overlap pass, HIP-event time per 1 M-region batch."""
    tab = synth.gencode_like_block_table(roots)
    table = engine.DepthTable(tab["n_groups"], tab["block_line_off"], tab["line_start"], tab["line_end"],
                              tab["line_group"], tab["block_of_fid"])
    batch = engine.QueryBatch(ix, nq)
    batch.set_regions_device(cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), nq, keep=cols)
    batch.run(2, False, engine.OUT_FIDS | engine.OUT_OFFSETS, 0)
    batch.wait()
    pairs = batch.total_hits
    table.accumulate(batch)  # warm
    batch.set_profiling(True)
    batch.reset_profile()
    reps = 10
    for _ in range(reps):
        table.accumulate(batch)
    batch.set_profiling(False)
    ms, n = batch.kernel_ms(engine.K_DEPTH)
    d, _, _ = table.results()
    pair_lines = float(np.diff(tab["block_line_off"]).mean()) * pairs  # (pair, block line) tests per batch
    us = 1e3 * ms / max(n, 1)
    batch.close()
    return {"kernel": "k_depth_regions", "avg_us": us, "regions_per_s": nq / (us * 1e-6),
            "line_table": {"lines": int(tab["block_line_off"][-1]), "groups": tab["n_groups"], "blocks": len(tab["block_line_off"]) - 1},
            "pair_line_tests_per_batch": pair_lines, "achieved_GBps": 12.0 * pair_lines / (us * 1e-6) / 1e9,
            "depth_sum_check": int(d.sum() // (reps + 1)),
            "note": "12 B per (pair, block line) read; reference: commands/depth.rs:121-217 (it re-parses a root's block text "
                    "for every batch that touches it)"}


def join_b_leg(engine, synth, roots, regions, mode):
    """Join B (commands/intersect.rs:500-521 for every line of the annotation): device preparation of the region tables
    (radix sorts, running max / min, directories) + k_lines_exists, HIP-event times."""
    tab = synth.gencode_like_block_table(roots)
    per_block = np.diff(tab["block_line_off"]).astype(np.int64)
    chr_of_root = np.repeat(np.arange(len(roots["chr_offsets"]) - 1), np.diff(roots["chr_offsets"]))
    seq = np.repeat(chr_of_root, per_block).astype(np.uint32)
    lt = engine.LineTable(seq, tab["line_start"] + 1, tab["line_end"])  # raw 1-based closed columns 4/5
    n_seq = len(roots["chr_offsets"]) - 1
    kept = lt.test(regions, n_seq, mode)  # warm
    us, prep = [], []
    for _ in range(5):
        lt.test(regions, n_seq, mode)
        us.append(1e3 * lt.last_kernel_ms)
        prep.append(1e3 * lt.last_prep_ms)
    avg = float(np.mean(us))
    out = {"kernel": "k_lines_exists2", "avg_us": avg, "prep_us": float(np.mean(prep)), "sort_passes": lt.last_sort_passes, "lines": int(lt.n),
           "regions": int(len(regions)),
           "lines_per_s": lt.n / (avg * 1e-6), "achieved_GBps": 13.0 * lt.n / (avg * 1e-6) / 1e9,
           "frac_of_hbm_peak": 13.0 * lt.n / (avg * 1e-6) / 1e9 / HBM_PEAK_GBS, "kept_lines": int(kept.sum()),
           "note": "13 B per line (seq, start, end in; keep flag out); prep_us = the region tables built on the device per call "
                   "(one stable radix sort by (seqid, start): k_radix_hist + sort_passes x k_radix_pass -- 4 at GRCh38 scale since round 5: "
                   "the last pass sorts by the mixed-radix digit of (seqid, top byte of start) --, then k_b_local / _carry / _finish)"}
    lt.close()
    return out


if __name__ == "__main__":
    main()
