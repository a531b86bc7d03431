"""Chromosome-bucket sharding + the hit-count all-gather, on CPU (gloo, world size 2).

The device join is replaced by the oracle here (tests may use it): what is under test is the
N>1 plumbing -- the plan is a partition of the batch, every rank derives the same plan, and the
all-gathered per-rank (regions, pairs) add up to the single-process answer."""
import os
import socket

import numpy as np
import pytest

from gffx_amd import shard, synth


def test_plan_is_a_balanced_partition():
    rng = np.random.default_rng(0)
    for n_ranks in (1, 2, 3, 4, 8):
        for trial in range(20):
            sizes = rng.integers(0, 5000, size=int(rng.integers(1, 40))).tolist()
            if trial == 0:
                sizes = [100000] + [10] * 5  # one dominant bucket must be split
            plan = shard.plan_shards(sizes, n_ranks)
            seen = [np.zeros(s, dtype=np.int32) for s in sizes]
            loads = []
            for r in range(n_ranks):
                load = 0
                for c, lo, hi in plan[r]:
                    assert 0 <= lo < hi <= sizes[c]
                    seen[c][lo:hi] += 1
                    load += hi - lo
                loads.append(load)
            assert all((s == 1).all() for s in seen), "every row exactly once"
            total = sum(sizes)
            if total:
                ideal = -(-total // n_ranks)
                assert max(loads) <= ideal + max(1, int(0.02 * ideal)) + max(sizes) * (n_ranks == 1) or \
                    max(loads) <= 1.25 * ideal + 64, (sizes, loads)
            assert shard.plan_shards(sizes, n_ranks) == plan  # deterministic


def test_grch38_like_batch_is_near_perfectly_balanced():
    regions = synth.synth_bed(200_000, seed=5)
    for n in (2, 4, 8):
        rows = [shard.shard_rows(regions, 25, n, r) for r in range(n)]
        allrows = np.concatenate(rows)
        assert len(allrows) == len(regions) and len(np.unique(allrows)) == len(regions)
        sizes = [len(r) for r in rows]
        assert max(sizes) <= 1.03 * len(regions) / n
    with pytest.raises(IndexError):
        shard.bucket_regions(np.array([[30, 1, 2]], np.uint32), 25)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist

    from oracle import binding as ob

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    roots = synth.gencode_like_roots(3000, seed=4)
    regions = synth.synth_bed(40_000, seed=6, edge_frac=0.01, roots=roots)
    rows = shard.shard_rows(regions, 25, world, rank)
    oix = ob.OracleIndex.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
    t, c = oix.query_features(regions[rows], 2, False)
    got = shard.allgather_hit_counts(len(rows), len(t))
    # the CLI's exchange: packed root-hit bitmaps (roots in (seqid, start) order) OR-ed over ranks; per-seqid hits
    hit = np.zeros(len(roots["fid"]), bool)
    hit[np.searchsorted(roots["fid"], t[:, 0])] = True  # (synth fids ascend with the sorted order)
    words = np.packbits(np.concatenate([hit, np.zeros(-len(hit) % 64, bool)]), bitorder="little").view(np.uint64)
    union = shard.allgather_root_bitmap(words)
    per_seq = np.zeros(25, np.int64)
    np.add.at(per_seq, regions[rows][:, 0].astype(np.int64), c.astype(np.int64))
    seq_tab = shard.allgather_seqid_hits(per_seq)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, got.tolist(), len(rows), len(t), union.tolist(), seq_tab.sum(axis=0).tolist()))


def test_two_rank_gloo_allgather_matches_single_process():
    import torch.multiprocessing as mp

    from oracle import binding as ob

    ob.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == res[1][1]  # every rank sees the same gathered table
    table = np.array(res[0][1])
    assert table[0].tolist() == [res[0][2], res[0][3]] and table[1].tolist() == [res[1][2], res[1][3]]
    roots = synth.gencode_like_roots(3000, seed=4)
    regions = synth.synth_bed(40_000, seed=6, edge_frac=0.01, roots=roots)
    oix = ob.OracleIndex.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
    t, c = oix.query_features(regions, 2, False)
    assert table[:, 0].sum() == len(regions) and table[:, 1].sum() == len(t)
    # bitmap union over the ranks == the single-process hit set; per-seqid hits add up
    assert res[0][4] == res[1][4] and res[0][5] == res[1][5]
    bits = np.unpackbits(np.array(res[0][4], np.uint64).view(np.uint8), bitorder="little")[: len(roots["fid"])].astype(bool)
    assert np.array_equal(roots["fid"][bits], np.unique(t[:, 0]))
    per_seq = np.zeros(25, np.int64)
    np.add.at(per_seq, regions[:, 0].astype(np.int64), c.astype(np.int64))
    assert res[0][5] == per_seq.tolist()
    assert np.array_equal(shard.allgather_root_bitmap(np.array([5, 9], np.uint64)), np.array([5, 9], np.uint64))


def test_cpp_plan_of_the_cli_equals_the_python_plan():
    """`gffx intersect --gpus N` shards every BED chunk with the C++ port of plan_shards (host/intersect.cpp)."""
    import ctypes as C
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    L = C.CDLL(os.path.join(root, "gffx_amd", "lib", "libgffx_host.so"))
    u64p = C.POINTER(C.c_uint64)
    L.gffx_host_plan_shards.argtypes = [u64p, C.c_uint32, C.c_uint32, C.POINTER(u64p), u64p]
    L.gffx_host_free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(3)
    for trial in range(200):
        n_chr = int(rng.integers(1, 40))
        sizes = rng.integers(0, 5000, n_chr).astype(np.uint64)
        if trial % 5 == 0:
            sizes[int(rng.integers(0, n_chr))] = int(rng.integers(10_000, 1_000_000))  # one dominant bucket: it is split
        if trial % 11 == 0:
            sizes[:] = 0
        n_ranks = int(rng.integers(1, 9))
        out, n = u64p(), C.c_uint64()
        assert L.gffx_host_plan_shards(sizes.ctypes.data_as(u64p), n_chr, n_ranks, C.byref(out), C.byref(n)) == 0
        got = [[] for _ in range(n_ranks)]
        for i in range(n.value):
            r, c, lo, hi = (int(out[4 * i + j]) for j in range(4))
            got[r].append((c, lo, hi))
        L.gffx_host_free(out)
        assert got == shard.plan_shards(sizes.tolist(), n_ranks), (sizes, n_ranks)


def test_bench_rank_placement_and_strong_scaling_shards():
    """What `bench.py --gpus 8` does with its environment, without a GPU: LOCAL_RANK r takes device r (wrapping only when the
    box has fewer devices than ranks), and --scaling strong gives rank r exactly shard_rows(configs[3]'s batch, 8, r) -- the
    partition tests/test_fullsize_gpu.py::test_config3_100m_regions_sharded_over_8_ranks runs through one GPU (here at 1/50 of
    its size: the plan is a function of the bucket sizes only)."""
    import bench

    for r in range(8):
        env = {"WORLD_SIZE": "8", "RANK": str(r), "LOCAL_RANK": str(r)}
        assert bench.rank_placement(env, 8) == (r, 8, r, r)
        assert bench.rank_placement(env, 1) == (r, 8, r, 0)
        assert bench.rank_placement(env, 2)[3] == r % 2
    assert bench.rank_placement({}, 4) == (0, 1, 0, 0)
    with pytest.raises(RuntimeError):
        bench.rank_placement({}, 0)
    n_chr, total = 25, 2_000_000
    ref = synth.synth_bed(total, seed=1003)
    seen = np.zeros(total, dtype=np.int32)
    sizes = []
    for r in range(8):
        part, nq_global, cfg = bench.bench_regions(synth, shard, n_chr, 8, r, "strong", strong_total=total)
        rows = shard.shard_rows(ref, n_chr, 8, r)
        assert nq_global == total and cfg == "configs[3]" and np.array_equal(part, ref[rows])
        seen[rows] += 1
        sizes.append(len(rows))
    assert (seen == 1).all() and max(sizes) <= 1.03 * total / 8
    # weak scaling: N x 1 M of configs[1]'s seed, sharded the same way; one rank: the batch itself
    part, nq_global, cfg = bench.bench_regions(synth, shard, n_chr, 2, 1, "weak", queries_per_gpu=50_000)
    ref = synth.synth_bed(100_000, seed=1001)
    assert nq_global == 100_000 and cfg == "configs[1]" and np.array_equal(part, ref[shard.shard_rows(ref, n_chr, 2, 1)])
    part, nq_global, _ = bench.bench_regions(synth, shard, n_chr, 1, 0, "strong", queries_per_gpu=30_000)
    assert nq_global == 30_000 and np.array_equal(part, synth.synth_bed(30_000, seed=1001))
