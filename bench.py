#!/usr/bin/env python3
"""bench.py -- BED overlap queries/sec against a GRCh38-scale GFF index on N MI355X.

One "step" = one pass of the intersect hot path (Join A: regions x root intervals -> per-region
kept counts + segment offsets in input order, and the root_fids of every kept pair) over one batch
of synthetic BED regions that is already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]: 1 M synthetic BED regions
(seed 1001, chr ~ length, width U[100,10000], unsorted) x a GENCODE/GRCh38-shaped index
(25 seqids, ~63 k root genes of a ~3.4 M-line annotation, seed 42), --overlap mode.
For N>1 the global batch is N x 1 M regions, sharded by chromosome bucket over the ranks
(gffx_amd.shard, LPT with splitting; index replicated); the only collective is the RCCL
all-gather of per-rank hit counts, once per job: it is timed on its own (config.exchange_ms) after the K
timed steps; --exchange final-timed puts it inside the timed region, per-step issues it after every step.

Steps are issued round-robin to --inflight (default 2) QueryBatch objects -- each with its own HIP
stream and result buffers, all reading the same resident regions -- so the launch ramp / drain of one
pass overlaps the next pass (34 -> 47 G regions/s on one MI355X); --inflight 1 gives strictly serial
passes.  The roofline object is computed from single-pass kernel durations either way.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--queries-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--mode", default="overlap", choices=["overlap", "contained", "contains_region"])
    ap.add_argument("--strategy", default="auto", choices=["auto", "direct", "sorted", "fused", "slots"])
    ap.add_argument("--out", default="fids", choices=["counts", "fids", "triples"])
    ap.add_argument("--no-offsets", action="store_true",
                    help="fused / partitioned strategy: do not write every region's segment offset")
    ap.add_argument("--input-order", action="store_true",
                    help="partitioned strategy: also scatter the per-region records into input-order arrays "
                         "inside the pass (k_unpermute); default leaves them as {row, count, offset} records")
    ap.add_argument("--presort", default="none", choices=["none", "chr_end", "bucket"],
                    help="EXPERIMENT ONLY: reorder the synthetic regions on the host before upload")
    ap.add_argument("--exchange", default="final", choices=["final", "final-timed", "per-step"],
                    help="N>1: all-gather the per-rank hit counts once at the end of the timed region (default, "
                         "north_star's 'final hit-count all-gather') or after every step (latency-bound)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches in flight per GPU: steps are issued round-robin to this many QueryBatch objects "
                         "(own HIP stream and buffers each, same resident regions), so one pass's ramp/drain overlaps "
                         "the next pass; 1 = strictly serial passes")
    ap.add_argument("--depth", action="store_true",
                    help="additionally time the `gffx depth` join (k_depth_regions) on the same regions against a "
                         "GENCODE-shaped line table (~3.4 M lines) and report it as an extra \"depth\" object")
    ap.add_argument("--join-b", action="store_true",
                    help="additionally time Join B (k_lines_exists: every line of a GENCODE-shaped line table against "
                         "all regions of its seqid) and report it as an extra \"join_b\" object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--cpu-allcore", action="store_true",
                    help="also report the oracle's Join A with the batch split over all host cores (tools/cpu_allcore.py, "
                         "run as a child process) as \"cpu_allcore\" -- a fairer CPU upper bound, not the reference's behaviour")
    return ap.parse_args()


def cpu_baseline(roots, regions, mode, budget_s):
    """Oracle (C restatement of the reference's serial tree walk, intersect.rs:124-166) timed on
    this host: 1 thread, like the reference.  Bounded sample of the same workload."""
    from oracle import binding as ob

    oix = ob.OracleIndex.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
    n = len(regions)
    oix.query_features(regions[: min(n, 20000)], mode, False)  # warm
    done, t_used, hits = 0, 0.0, 0
    reps = 0
    while t_used < budget_s and reps < 400:
        t0 = time.perf_counter()
        t, _ = oix.query_features(regions, mode, False)
        t_used += time.perf_counter() - t0
        done += n
        hits = len(t)
        reps += 1
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": done / t_used, "unit": "queries/s", "cores": 1, "kind": "port",
            "host": {"cpu": model, "nproc": os.cpu_count()},
            "sample": "%d x the full %d-region batch of this workload, Join A only (pointer-based centered "
                      "interval tree, serial, as commands/intersect.rs:124-166); C restatement, not the Rust binary"
                      % (reps, n),
            "pairs_per_batch": hits}


def depth_leg(engine, synth, roots, batch, mode, nq, pairs):
    """`gffx depth` (BASELINE configs[4]'s command, reference semantics: per-feature-ID region counts):
    k_depth_regions over the pairs of an Overlap pass, HIP-event time per 1 M-region batch."""
    tab = synth.gencode_like_block_table(roots)
    table = engine.DepthTable(tab["n_groups"], tab["block_line_off"], tab["line_start"], tab["line_end"],
                              tab["line_group"], tab["block_of_fid"])
    flags = engine.OUT_FIDS | engine.OUT_OFFSETS
    batch.run(2, False, flags, 0)
    batch.wait()
    table.accumulate(batch)  # warm
    batch.set_profiling(True)
    batch.reset_profile()
    reps = 10
    for _ in range(reps):
        table.accumulate(batch)
    batch.set_profiling(False)
    ms, n = batch.kernel_ms(engine.K_DEPTH)
    d, _, _ = table.results()
    lines = int(tab["block_line_off"][-1])
    pair_lines = float(np.diff(tab["block_line_off"]).mean()) * pairs  # (pair, block line) tests per batch
    us = 1e3 * ms / max(n, 1)
    return {"kernel": "k_depth_regions", "avg_us": us, "regions_per_s": nq / (us * 1e-6),
            "line_table": {"lines": lines, "groups": tab["n_groups"], "blocks": len(tab["block_line_off"]) - 1},
            "pair_line_tests_per_batch": pair_lines,
            "achieved_GBps": 12.0 * pair_lines / (us * 1e-6) / 1e9,
            "depth_sum_check": int(d.sum() // (reps + 1)),
            "note": "12 B per (pair, block line) read; reference: commands/depth.rs:121-217 (it re-parses a root's "
                    "block text for every batch that touches it)"}


def join_b_leg(engine, synth, roots, regions, mode):
    """Join B (commands/intersect.rs:500-521 for every line of the annotation): k_lines_exists, HIP-event time."""
    tab = synth.gencode_like_block_table(roots)
    per_block = np.diff(tab["block_line_off"]).astype(np.int64)
    chr_of_root = np.repeat(np.arange(len(roots["chr_offsets"]) - 1), np.diff(roots["chr_offsets"]))
    seq = np.repeat(chr_of_root, per_block).astype(np.uint32)
    lt = engine.LineTable(seq, tab["line_start"] + 1, tab["line_end"])  # raw 1-based closed columns 4/5
    n_seq = len(roots["chr_offsets"]) - 1
    kept = lt.test(regions, n_seq, mode)  # warm
    us = []
    for _ in range(5):
        lt.test(regions, n_seq, mode)
        us.append(1e3 * lt.last_kernel_ms)
    avg = float(np.mean(us))
    # CPU beside it: the reference scans ALL regions of the line's seqid per line (intersect.rs:500-521);
    # the oracle's literal scan on a bounded sample of lines, 1 thread
    from oracle import binding as ob
    order = np.argsort(regions[:, 0], kind="stable")
    r = regions[order]
    off = np.concatenate([[0], np.cumsum(np.bincount(r[:, 0], minlength=n_seq))])
    pick = np.random.default_rng(5).choice(lt.n, size=400, replace=False)
    raw_s, raw_e = tab["line_start"] + 1, tab["line_end"]
    t0 = time.perf_counter()
    for i in pick.tolist():
        c = int(seq[i])
        ob.line_predicate(int(raw_s[i]), int(raw_e[i]), r[off[c]:off[c + 1], 1], r[off[c]:off[c + 1], 2], mode)
    cpu_s = time.perf_counter() - t0
    return {"kernel": "k_lines_exists", "avg_us": avg,
            "cpu_baseline": {"value": len(pick) / cpu_s, "unit": "lines/s", "cores": 1, "kind": "port",
                             "sample": "%d random lines, literal scan of all regions of the line's seqid" % len(pick)}, "lines": int(lt.n), "regions": int(len(regions)),
            "lines_per_s": lt.n / (avg * 1e-6), "achieved_GBps": 13.0 * lt.n / (avg * 1e-6) / 1e9,
            "kept_lines": int(kept.sum()),
            "note": "13 B per line (seq, start, end in; keep flag out); the regions' sort / prefix-max tables are "
                    "prepared on the host per call and are not in this time"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus),
                  file=sys.stderr)
            sys.exit(2)
        args.gpus = world

    import torch
    import torch.distributed as dist

    from gffx_amd import engine, shard, synth

    if not torch.cuda.is_available() or engine.device_count() < 1:
        print("bench.py: no MI355X visible; the engine has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    # GFFX_BENCH_BACKEND=gloo lets the N>1 plumbing be exercised on a 1-GPU box (all ranks share
    # device 0); the driver's runs use nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("GFFX_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    mode = {"contained": 0, "contains_region": 1, "overlap": 2}[args.mode]
    strategy = {"auto": 0, "direct": 1, "sorted": 2, "fused": 3, "slots": 4}[args.strategy]
    out_flags = {"counts": engine.OUT_COUNTS, "fids": engine.OUT_FIDS, "triples": engine.OUT_TRIPLES}[args.out]
    if args.strategy != "direct" and args.out != "counts" and not args.no_offsets:
        out_flags |= engine.OUT_OFFSETS  # segments follow the order rounds / tiles were served: offsets are explicit
    if args.strategy == "sorted" and not args.input_order:
        out_flags |= engine.OUT_EMIT_ORDER

    # ---- synthetic inputs (identical on every rank; each rank keeps its shard)
    roots = synth.gencode_like_roots(63000, seed=42)
    n_chr = len(roots["chr_offsets"]) - 1
    nq_global = args.queries_per_gpu * world
    regions_all = synth.synth_bed(nq_global, seed=1001)
    if world > 1:
        rows = shard.shard_rows(regions_all, n_chr, world, rank)
        regions = np.ascontiguousarray(regions_all[rows])
    else:
        regions = regions_all
    if args.presort == "chr_end":
        regions = regions[np.lexsort((regions[:, 2], regions[:, 0]))]
    elif args.presort == "bucket":  # (chr, end >> 21) buckets, input order inside a bucket
        key = (regions[:, 0].astype(np.int64) << 11) | (regions[:, 2].astype(np.int64) >> 21)
        regions = regions[np.argsort(key, kind="stable")]
    regions = np.ascontiguousarray(regions)
    nq = len(regions)

    ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"],
                                         roots["names"], device=dev_index)
    # regions resident in HBM as SoA u32 (torch owns the memory; the engine borrows the pointers)
    t_regions = torch.from_numpy(np.ascontiguousarray(regions).view(np.int32))  # u32 bit patterns
    d_chr = t_regions[:, 0].contiguous().to(dev)
    d_start = t_regions[:, 1].contiguous().to(dev)
    d_end = t_regions[:, 2].contiguous().to(dev)
    torch.cuda.synchronize()
    batches = []
    for _ in range(max(1, args.inflight)):
        bb = engine.QueryBatch(ix, max(nq, 1))
        bb.set_regions_device(d_chr.data_ptr(), d_start.data_ptr(), d_end.data_ptr(), nq,
                              keep=(d_chr, d_start, d_end))
        batches.append(bb)
    batch = batches[0]
    issued = [0]

    coll_dev = dev if backend == "nccl" else None  # gloo gathers CPU tensors

    def step():
        batches[issued[0] % len(batches)].run(mode, False, out_flags, strategy)
        issued[0] += 1

    def sync_all():
        for bb in batches:
            bb.sync()

    def exchange():
        # the path's one exchange step: all-gather of per-rank (queries, kept pairs)
        if world > 1 and args.exchange == "per-step":
            batch.wait()
            return shard.allgather_hit_counts(nq, batch.total_hits, device=coll_dev)
        return None

    # sizing pass (also the parity-relevant total), then warmup
    step()
    batch.wait()
    pairs = batch.total_hits
    for _ in range(args.warmup * len(batches)):
        step()
        exchange()
    for bb in batches:  # (sizes every batch's pair buffers: a capacity replay can only happen here)
        if bb is batch or args.warmup > 0:
            bb.wait()
    issued[0] = 0
    if world > 1:  # the collective's first call sets up its channels: not part of the job's steady state
        shard.allgather_hit_counts(nq, pairs, device=coll_dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        exchange()
    if world > 1 and args.exchange == "final-timed":
        sync_all()
        batch.wait()  # the job's one exchange step, inside the timed region
        shard.allgather_hit_counts(nq, batch.total_hits, device=coll_dev)
    else:
        sync_all()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0  # this rank's K steps (+ the exchange); MAX over ranks below
    barrier()
    exchange_ms = None
    if world > 1:
        # the job's ONE exchange step (north_star: the final hit-count all-gather): once per job, not per step, so it is
        # timed on its own (--exchange final-timed puts it inside the K-step region, per-step runs one per step)
        batch.wait()
        tx = time.perf_counter()
        shard.allgather_hit_counts(nq, batch.total_hits, device=coll_dev)
        torch.cuda.synchronize()
        exchange_ms = 1e3 * (time.perf_counter() - tx)
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        counts = shard.allgather_hit_counts(nq, pairs, device=coll_dev)
        nq_total, pairs_total = int(counts[:, 0].sum()), int(counts[:, 1].sum())
    else:
        nq_total, pairs_total = nq, pairs
    sync_all()
    batch.wait()
    issued[0] = 0  # the profiled loop below runs on batch 0 only, one pass at a time

    # ---- per-kernel durations, HIP events on the engine's own stream (separate profiled loop)
    batch.set_profiling(True)
    batch.reset_profile()
    n_prof = max(5, min(args.steps, 30))
    for _ in range(n_prof):  # back to back on the engine's stream, one event pair per launch, one sync at the end
        batch.run(mode, False, out_flags, strategy)
    batch.sync()
    batch.set_profiling(False)
    kern = {}
    for kid, name in engine.KERNEL_NAMES.items():
        ms, n = batch.kernel_ms(kid)
        if n:
            kern[name] = {"avg_us": 1e3 * ms / n, "launches_per_step": n / n_prof}

    result = None
    if rank == 0:
        h_bar = pairs / max(nq, 1)
        out_b = {"counts": 0.0, "fids": 4.0, "triples": 12.0}[args.out]
        bytes_per_query = 12.0 + 4.0 + out_b * h_bar  # regions in, count out, pairs out
        pass_us = sum(k["avg_us"] * k["launches_per_step"] for k in kern.values())
        dominant = max(kern.items(), key=lambda kv: kv[1]["avg_us"] * kv[1]["launches_per_step"])[0] if kern else None
        achieved = (bytes_per_query * nq) / (pass_us * 1e-6) / 1e9 if pass_us > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_pass")
            except Exception:
                traffic = None
        result = {
            "metric": "BED overlap queries/sec vs GRCh38-scale GFF index at 1/2/4/8 MI355X",  # BASELINE.json
            "value": nq_total * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: %d synthetic BED regions per GPU (seed 1001) x GENCODE/GRCh38-shaped "
                            "index (25 seqids, %d root genes, seed 42), --%s, regions resident in HBM as u32 SoA"
                            % (args.queries_per_gpu, ix.n_roots, args.mode),
                "regions_total": nq_total,
                "kept_pairs_total": pairs_total,
                "pairs_per_region": h_bar,
                "outputs": {"direct": "per-region counts (input order) + %s in CSR order" % args.out,
                            "sorted": "per-region {row, count, offset} records in tile order%s + %s"
                                      % (" and input-order counts/offsets" if args.input_order else "", args.out),
                            }.get(args.strategy,
                                  "per-region counts%s (input order) + %s, segments in round order"
                                  % ("" if args.no_offsets or args.out == "counts" else " and segment offsets", args.out)),
                "strategy": args.strategy,
                "presort": args.presort,
                "batches_in_flight": len(batches),
                "sharding": ("chromosome buckets, LPT with splitting; index replicated; all-gather of hit counts "
                             + {"per-step": "after every step (inside the timed region)",
                                "final-timed": "once per job, inside the timed region",
                                "final": "once per job, after the K timed steps (exchange_ms)"}[args.exchange])
                            if world > 1 else "none (1 GPU)",
                "exchange_ms": exchange_ms,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_of_copy_ceiling": achieved / 6300.0,  # also against the measured ~6.3 TB/s copy ceiling
                "traffic": traffic,
                "kernel": "+".join(sorted(kern)) if kern else None,
                "dominant_kernel": dominant,
                "algorithmic_bytes_per_pass": bytes_per_query * nq,
                "pass_kernel_us": pass_us,
                "kernels": kern,
                "note": "achieved = (12 B region + 4 B count + 4 B x pairs/region) x regions / summed HIP-event "
                        "durations of the pass's kernels on the engine's stream (rank 0)",
            },
        }
        if args.depth:
            result["depth"] = depth_leg(engine, synth, roots, batch, mode, nq, pairs)
        if args.join_b:
            result["join_b"] = join_b_leg(engine, synth, roots, regions, mode)
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(roots, regions, mode, args.cpu_seconds)
        elif not args.no_cpu_baseline:
            result["cpu_baseline"] = None
        if args.cpu_allcore and world == 1:
            import subprocess
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_allcore.py"), str(nq), str(mode), "5"],
                               capture_output=True, text=True)
            try:
                result["cpu_allcore"] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception:
                result["cpu_allcore"] = {"error": (r.stderr or r.stdout)[-300:]}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
