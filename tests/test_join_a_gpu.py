"""Join A parity on the GPU: HIP engine (through the C-ABI) == oracle, bit for bit.

Compared per stage (SURVEY.md section 8a "hit set"): per-query kept counts, the multiset of
(root_fid, start, end) triples, the CSR grouping of pairs per query, and the unique-root set.
"""
import os

import numpy as np
import pytest

from gffx_amd import engine, synth
from gffx_amd.engine import OverlapMode
from oracle import binding as ob

pytestmark = pytest.mark.gpu

FLAGS = engine.OUT_FIDS | engine.OUT_TRIPLES | engine.OUT_ROOT_BITMAP | engine.OUT_OFFSETS
STRATEGIES = [engine.STRATEGY_DIRECT, engine.STRATEGY_SORTED, engine.STRATEGY_FUSED, engine.STRATEGY_WINDOWS]


def _sorted_rows(t):
    t = np.asarray(t, dtype=np.uint32).reshape(-1, 3)
    return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]


def _check(roots, regions, mode, invert, soa=False, strategy=engine.STRATEGY_AUTO):
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    want_t, want_c = oix.query_features(regions, int(mode), invert)
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    b = engine.QueryBatch(ix, max(len(regions), 1))
    if soa:
        b.set_regions_soa(regions[:, 0], regions[:, 1], regions[:, 2])
    else:
        b.set_regions(regions)
    b.run(mode, invert, FLAGS, strategy)
    b.wait()
    assert b.total_hits == len(want_t)
    got_c = b.counts()
    assert np.array_equal(got_c, want_c)
    got_t = b.triples()
    assert np.array_equal(_sorted_rows(got_t), _sorted_rows(want_t))
    assert np.array_equal(b.fids(), got_t[:, 0])
    off = b.offsets()
    assert int(off[-1]) == len(want_t)
    if strategy == engine.STRATEGY_DIRECT:  # direct: CSR in input order
        assert np.array_equal(off, np.concatenate([[0], np.cumsum(want_c.astype(np.uint64))]).astype(np.uint64))
    else:  # partitioned (or AUTO): every query's segment is given explicitly; the segments tile [0, pairs)
        nz = want_c > 0
        seg_lo, seg_hi = off[:-1][nz], off[:-1][nz] + want_c[nz]
        order = np.argsort(seg_lo)
        assert len(seg_lo) == 0 or (seg_lo[order][0] == 0 and seg_hi[order][-1] == len(want_t)
                                    and np.array_equal(seg_hi[order][:-1], seg_lo[order][1:]))
    # pairs of query i are exactly the oracle's pairs of query i (their order inside a query is free)
    for qi in np.random.default_rng(0).choice(len(regions), size=min(200, len(regions)), replace=False):
        seg = got_t[int(off[qi]):int(off[qi]) + int(got_c[qi])]
        one_t, _ = oix.query_features(regions[qi:qi + 1], int(mode), invert)
        assert np.array_equal(_sorted_rows(seg), _sorted_rows(one_t))
    assert np.array_equal(b.unique_roots(), np.unique(want_t[:, 0]))
    # a root_fid-only pass (what bench.py and `depth` run: its own emit path in the one-kernel strategies):
    # the (query, root_fid) pairs of EVERY query
    b.run(mode, invert, engine.OUT_FIDS | engine.OUT_OFFSETS, strategy)
    b.wait()
    c2, off2, f2 = b.counts(), b.offsets(), b.fids()
    assert np.array_equal(c2, want_c) and len(f2) == len(want_t)
    wc = want_c.astype(np.int64)
    qid = np.repeat(np.arange(len(regions), dtype=np.int64), wc)
    within = np.arange(len(qid), dtype=np.int64) - np.repeat(np.cumsum(wc) - wc, wc)
    got_pairs = np.stack([qid, f2[off2[:-1].astype(np.int64)[qid] + within].astype(np.int64)], axis=1)
    by_chr = np.argsort(regions[:, 0], kind="stable")  # the oracle walks seqid after seqid, regions in input order
    want_pairs = np.stack([np.repeat(by_chr, wc[by_chr]), want_t[:, 0].astype(np.int64)], axis=1)
    order = lambda a: a[np.lexsort((a[:, 1], a[:, 0]))]  # noqa: E731
    assert np.array_equal(order(got_pairs), order(want_pairs))
    if strategy in (engine.STRATEGY_AUTO, engine.STRATEGY_WINDOWS):
        # the pass bench.py times: counts + root_fids + ONE segment base per group of 256 regions (no per-region offsets);
        # a consumer derives a region's segment from the group's base and the counts before it
        b.run(mode, invert, engine.OUT_FIDS | engine.OUT_SEGBASE, strategy)
        b.wait()
        c3, f3, sb = b.counts(), b.fids(), b.segbase()
        assert np.array_equal(c3, want_c) and len(f3) == len(want_t) and len(sb) == (len(regions) + 255) // 256
        off3 = b.offsets_from_segbase(c3).astype(np.int64)
        got3 = np.stack([qid, f3[off3[qid] + within].astype(np.int64)], axis=1)
        assert np.array_equal(order(got3), order(want_pairs))
        gtot = np.add.reduceat(wc, np.arange(0, len(wc), 256)) if len(wc) else np.zeros(0, np.int64)
        nzg = gtot > 0  # the groups' runs tile [0, pairs)
        lo, hi = sb.astype(np.int64)[nzg], sb.astype(np.int64)[nzg] + gtot[nzg]
        o3 = np.argsort(lo)
        assert len(lo) == 0 or (lo[o3][0] == 0 and hi[o3][-1] == len(want_t) and np.array_equal(hi[o3][:-1], lo[o3][1:]))
    b.run(mode, invert, FLAGS, strategy)  # (back to the full pass for the records below; segment order is per pass)
    b.wait()
    off = b.offsets()
    # the same results as {input row, count, offset} records in emission order
    rows, rc, ro = b.query_records()
    assert np.array_equal(np.sort(rows), np.arange(len(regions), dtype=np.uint32))
    assert np.array_equal(rc, want_c[rows]) and np.array_equal(ro, off[:-1][rows])
    b.close()
    ix.close()
    return len(want_t)


@pytest.mark.parametrize("strategy", STRATEGIES)
@pytest.mark.parametrize("mode", list(OverlapMode))
@pytest.mark.parametrize("invert", [False, True])
def test_small_two_chromosomes_all_modes(mode, invert, strategy):
    roots = synth.gencode_like_roots(300, seed=3, chroms=synth.SMALL2)
    regions = synth.synth_bed(5000, seed=11, chroms=synth.SMALL2, width=(10, 200000), edge_frac=0.2, roots=roots)
    n = _check(roots, regions, mode, invert, strategy=strategy)
    if mode == OverlapMode.Overlap and invert:
        assert n == 0


@pytest.mark.parametrize("strategy", STRATEGIES)
@pytest.mark.parametrize("mode", list(OverlapMode))
def test_gencode_scale_100k_queries(mode, strategy):
    roots = synth.gencode_like_roots(63000, seed=42)
    regions = synth.synth_bed(100_000, seed=1001, edge_frac=0.001, roots=roots)
    n = _check(roots, regions, mode, False, soa=(mode == OverlapMode.Overlap), strategy=strategy)
    assert n > 0 or mode != OverlapMode.Overlap


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_dense_small_coordinates_all_strategies(seed):
    """Random intervals on a tiny coordinate range: deep nesting, hundreds of hits per region (the
    fused kernel's per-wave hit queues overflow and replay; pair buffers are regrown), ties on
    start and end, degenerate regions -- every strategy, mode and invert against the oracle."""
    rng = np.random.default_rng(100 + seed)
    n_chr = int(rng.integers(1, 4))
    span = int(rng.choice([40, 300, 5000]))
    per = rng.integers(0, 1500, n_chr)
    co = np.concatenate([[0], np.cumsum(per)]).astype(np.uint32)
    n = int(co[-1])
    start = rng.integers(0, span, n).astype(np.uint32)
    end = (start + rng.integers(1, max(2, span // int(rng.choice([1, 3, 20]))), n)).astype(np.uint32)
    roots = {"chr_offsets": co, "start": start, "end": end, "fid": rng.permutation(n).astype(np.uint32) * 7}
    nq = int(rng.integers(1, 6000))
    qs = rng.integers(0, span + 5, nq)
    qe = qs + rng.integers(-3, max(2, span // 2), nq)  # includes qs >= qe rows
    regions = np.stack([rng.integers(0, n_chr, nq), qs, np.maximum(qe, 0)], axis=1).astype(np.uint32)
    for strategy in STRATEGIES:
        for mode in OverlapMode:
            _check(roots, regions, mode, bool(rng.integers(0, 2)), soa=bool(rng.integers(0, 2)), strategy=strategy)


def test_appendix_e_table_on_device():
    # SURVEY.md App. E roots: seq0 {[100,200)->0, [150,400)->4}, seq1 {[0,50)->6}
    ix = engine.TreeIndexData.from_roots([0, 2, 3], [100, 150, 0], [200, 400, 50], [0, 4, 6], ["chr1", "chr2"])

    def q(c, s, e, m, inv=False):
        return sorted(engine.query_features(ix, [[c, s, e]], m, inv)[:, 0].tolist())

    assert q(0, 200, 250, OverlapMode.Overlap) == [4]
    assert q(0, 200, 250, OverlapMode.Overlap, True) == []
    assert q(0, 100, 200, OverlapMode.Contained) == [0]
    assert q(0, 100, 200, OverlapMode.Contained, True) == [4]
    assert q(0, 160, 170, OverlapMode.ContainsRegion) == [0, 4]
    assert q(0, 200, 201, OverlapMode.Overlap) == [4]
    assert q(1, 0, 1, OverlapMode.Overlap) == [6]
    assert engine.query_features(ix, [[0, 100, 200]], OverlapMode.Contained).tolist() == [[0, 100, 200]]


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_edge_inputs(strategy):
    roots = synth.gencode_like_roots(50, seed=5, chroms=synth.SMALL2)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    # empty batch
    assert engine.query_features(ix, np.zeros((0, 3), np.uint32)).shape == (0, 3)
    # out-of-range chr is an error (the reference panics: intersect.rs:117)
    with pytest.raises(engine._ffi.GffxHipError) as ei:
        engine.query_features(ix, [[2, 0, 10]])
    assert ei.value.code == -5
    # u32 extremes and degenerate rows
    regions = np.array([[0, 0, 0], [0, 0, 0xFFFFFFFF], [1, 0xFFFFFFFF, 0], [0, 0xFFFFFFFF, 0xFFFFFFFF],
                        [1, 5, 5], [0, 3_000_000, 10]], dtype=np.uint32)
    for mode in OverlapMode:
        for inv in (False, True):
            _check(roots, regions, mode, inv, strategy=strategy)
    # an index with an empty seqid and a seqid of identical intervals
    roots2 = {"chr_offsets": np.array([0, 0, 4, 5], np.uint32), "start": np.array([7, 7, 7, 7, 0], np.uint32),
              "end": np.array([9, 9, 9, 9, 1], np.uint32), "fid": np.array([1, 2, 3, 4, 5], np.uint32)}
    regions2 = np.array([[0, 0, 100], [1, 8, 9], [1, 9, 10], [1, 6, 7], [1, 6, 8], [2, 0, 1], [2, 1, 2]], np.uint32)
    for mode in OverlapMode:
        _check(roots2, regions2, mode, False, strategy=strategy)
    # u32-wide coordinates: starts near 2^32 exercise the bin directory's shift
    roots3 = {"chr_offsets": np.array([0, 3], np.uint32),
              "start": np.array([0, 0xFFFFFF00, 0x80000000], np.uint32),
              "end": np.array([0xFFFFFFFF, 0xFFFFFFFF, 0x80000001], np.uint32),
              "fid": np.array([10, 20, 30], np.uint32)}
    regions3 = np.array([[0, 0, 1], [0, 0x80000000, 0x80000001], [0, 0xFFFFFF00, 0xFFFFFFFF],
                         [0, 0xFFFFFFFE, 0xFFFFFFFF], [0, 0x7FFFFFFF, 0x80000000]], np.uint32)
    for mode in OverlapMode:
        _check(roots3, regions3, mode, False, strategy=strategy)
    # empty intervals (end == start) are kept whenever qs < start < qe (tree.rs:110 has no other condition) -- the coverage
    # filter of the window index must not hide them.  (end < start is outside the domain: IntervalTree::build, tree.rs:48-50,
    # never terminates on such an interval.)
    roots4 = {"chr_offsets": np.array([0, 5], np.uint32),
              "start": np.array([100, 5000, 70_000, 900_000, 2_000_000], np.uint32),
              "end": np.array([100, 5000, 70_010, 900_000, 2_000_000], np.uint32),
              "fid": np.array([1, 2, 4, 5, 6], np.uint32)}
    regions4 = np.array([[0, 0, 200], [0, 100, 101], [0, 99, 100], [0, 99, 101], [0, 4980, 5010], [0, 4995, 5001], [0, 5000, 5001],
                         [0, 890_000, 910_000], [0, 899_500, 900_001], [0, 1_999_999, 2_000_001], [0, 2_000_000, 2_000_001],
                         [0, 0, 3_000_000]], np.uint32)
    for mode in OverlapMode:
        for inv in (False, True):
            _check(roots4, regions4, mode, inv, strategy=strategy)
    # empty index
    ix0 = engine.TreeIndexData.from_roots([0], [], [], [])
    assert engine.query_features(ix0, np.zeros((0, 3), np.uint32)).shape == (0, 3)


def test_many_seqids_metadata_outside_lds():
    """4000 scaffolds: the seqid table no longer fits the LDS staging budget (global path); the
    partitioned strategy still works with one genome cell per scaffold."""
    chroms = [("scaf%d" % i, 50_000 + 13 * i) for i in range(4000)]
    roots = synth.gencode_like_roots(12000, seed=8, chroms=chroms)
    regions = synth.synth_bed(30000, seed=9, chroms=chroms, width=(10, 5000), edge_frac=0.05, roots=roots)
    for mode in OverlapMode:
        for strategy in STRATEGIES:
            _check(roots, regions, mode, False, strategy=strategy)
    # ... but 5000 seqids are more genome cells than the partitioned strategy supports: loud error,
    # while AUTO quietly takes the direct strategy
    chroms = [("scaf%d" % i, 50_000 + 13 * i) for i in range(5000)]
    roots = synth.gencode_like_roots(12000, seed=8, chroms=chroms)
    regions = synth.synth_bed(40000, seed=9, chroms=chroms, width=(10, 5000), edge_frac=0.05, roots=roots)
    _check(roots, regions, OverlapMode.Overlap, False)
    ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
    b = engine.QueryBatch(ix, 10)
    b.set_regions(regions[:10])
    with pytest.raises(engine._ffi.GffxHipError):
        b.run(OverlapMode.Overlap, False, engine.OUT_FIDS, engine.STRATEGY_SORTED)


def test_many_seqids_lds_budget(monkeypatch):
    """1 200 scaffolds: 19 KB of seqid records on top of filter, split bitmap, strips and -- with per-region offsets, what
    `gffx depth` asks for -- the parked offsets: at both block widths the pass must shed what does not fit the block's LDS
    (split bitmap, then the seqid records, then the filter) instead of launching over the limit (round 3's ADVICE), and stay
    exact whatever it shed."""
    chroms = [("scaf%d" % i, 80_000 + 17 * i) for i in range(1200)]
    roots = synth.gencode_like_roots(9000, seed=18, chroms=chroms)
    regions = synth.synth_bed(30000, seed=19, chroms=chroms, width=(10, 6000), edge_frac=0.05, roots=roots)
    for threads in ("512", "1024"):
        monkeypatch.setenv("GFFX_HIP_WIN_THREADS", threads)
        for mode in (OverlapMode.Overlap, OverlapMode.Contained):
            _check(roots, regions, mode, False, strategy=engine.STRATEGY_WINDOWS)
        _check(roots, regions, OverlapMode.ContainsRegion, True, strategy=engine.STRATEGY_WINDOWS)


def test_sorted_strategy_long_queries_and_dense_windows():
    """Queries far longer than a genome window (the sweep leaves the LDS tile and continues in
    global memory) and a window with more intervals than fit the LDS tile (gather fallback)."""
    rng = np.random.default_rng(5)
    n = 6000
    start = np.sort(rng.integers(0, 200_000_000, n)).astype(np.uint32)
    start[1000:4000] = 50_000_000 + np.arange(3000, dtype=np.uint32) * 7  # 3000 intervals inside one window
    start.sort()
    end = start + rng.integers(1, 3_000_000, n).astype(np.uint32)
    roots = {"chr_offsets": np.array([0, n], np.uint32), "start": start, "end": end,
             "fid": np.arange(n, dtype=np.uint32)}
    nq = 4000
    qs = rng.integers(0, 200_000_000, nq).astype(np.int64)
    w = np.where(rng.random(nq) < 0.3, rng.integers(5_000_000, 150_000_000, nq), rng.integers(1, 20000, nq))
    regions = np.stack([np.zeros(nq, np.int64), qs, qs + w], axis=1).astype(np.uint32)
    regions[:50, 1] = 0  # start at 0, end anywhere
    for mode in OverlapMode:
        for strategy in STRATEGIES:
            _check(roots, regions, mode, False, strategy=strategy)


def test_dense_bins_saturate_the_bin_counter():
    """> 31 intervals starting inside one directory bin (the record's 5-bit count saturates)."""
    k = 200
    rng = np.random.default_rng(3)
    start = np.concatenate([np.full(k, 1000, np.uint32), rng.integers(0, 4_000_000, 50).astype(np.uint32)])
    end = start + rng.integers(1, 500, len(start)).astype(np.uint32)
    roots = {"chr_offsets": np.array([0, len(start)], np.uint32), "start": start, "end": end,
             "fid": np.arange(len(start), dtype=np.uint32)}
    regions = np.array([[0, 900, 1001], [0, 1000, 1001], [0, 999, 1000], [0, 1001, 1200], [0, 0, 4_100_000],
                        [0, 1400, 1500]] * 50, np.uint32)
    for mode in OverlapMode:
        for strategy in STRATEGIES:
            _check(roots, regions, mode, False, strategy=strategy)


@pytest.mark.parametrize("mode", list(OverlapMode))
def test_emit_order_records_without_the_input_order_scatter(mode):
    """GFFX_OUT_EMIT_ORDER: the pass leaves {row, count, offset} records; the input-order arrays
    are still available on demand and agree."""
    roots = synth.gencode_like_roots(63000, seed=42)
    regions = synth.synth_bed(200_000, seed=77, edge_frac=0.002, roots=roots)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    want_t, want_c = oix.query_features(regions, int(mode), False)
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    b = engine.QueryBatch(ix, len(regions))
    b.set_regions(regions)
    for _ in range(2):  # twice: the cursor sets alternate between passes
        b.run(mode, False, engine.OUT_FIDS | engine.OUT_OFFSETS | engine.OUT_EMIT_ORDER, engine.STRATEGY_SORTED)
        b.wait()
        assert b.total_hits == len(want_t)
        rows, cnt, off = b.query_records()
        assert np.array_equal(np.sort(rows), np.arange(len(regions), dtype=np.uint32))
        assert np.array_equal(cnt, want_c[rows])
        fids = b.fids()
        assert np.array_equal(np.sort(fids), np.sort(want_t[:, 0]))
        # every region's segment holds exactly its oracle root_fids
        order = np.argsort(rows)
        seg_of_row = off[order]
        for qi in np.random.default_rng(1).choice(len(regions), size=300, replace=False):
            one_t, _ = oix.query_features(regions[qi:qi + 1], int(mode), False)
            seg = fids[int(seg_of_row[qi]):int(seg_of_row[qi]) + int(want_c[qi])]
            assert np.array_equal(np.sort(seg), np.sort(one_t[:, 0]))
        assert np.array_equal(b.counts(), want_c)  # scatter on demand
        assert np.array_equal(b.offsets()[:-1][rows], off)


def test_one_batch_object_alternating_strategies_and_modes():
    """The cursor words / cursor sets of the fused and partitioned strategies alternate between passes;
    mixing strategies, modes and batch sizes on ONE batch object must not leak state between passes."""
    roots = synth.gencode_like_roots(5000, seed=21)
    regions = synth.synth_bed(150_000, seed=22, edge_frac=0.01, roots=roots)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    b = engine.QueryBatch(ix, len(regions))
    rng = np.random.default_rng(3)
    want = {}
    for step in range(14):
        strategy = STRATEGIES[int(rng.integers(0, len(STRATEGIES)))]
        mode = int(rng.integers(0, 3))
        n = int(rng.choice([len(regions), 70_000, 33, 0]))
        if (mode, n) not in want:
            want[(mode, n)] = oix.query_features(regions[:n], mode, False)
        wt, wc = want[(mode, n)]
        b.set_regions(regions[:n])
        b.run(mode, False, engine.OUT_FIDS | engine.OUT_OFFSETS, strategy)
        b.wait()
        assert b.total_hits == len(wt), (step, strategy, mode, n)
        assert np.array_equal(b.counts(), wc)
        assert np.array_equal(np.sort(b.fids()), np.sort(wt[:, 0]))


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_more_than_2pow20_pairs_for_one_region(strategy):
    """1.1 M identical intervals: a region's rank inside the fused kernel's hit queue no longer fits 20 bits
    (chain replay), the whole seqid is one over-full genome cell (partitioned: gather path)."""
    k = 1_100_000
    roots = {"chr_offsets": np.array([0, k, k + 2], np.uint32),
             "start": np.concatenate([np.full(k, 5, np.uint32), [1, 50]]).astype(np.uint32),
             "end": np.concatenate([np.full(k, 10, np.uint32), [3, 60]]).astype(np.uint32),
             "fid": np.arange(k + 2, dtype=np.uint32)}
    regions = np.array([[0, 9, 20], [1, 0, 100], [0, 10, 11], [0, 0, 6]], np.uint32)
    ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
    b = engine.QueryBatch(ix, len(regions))
    b.set_regions(regions)
    b.run(OverlapMode.Overlap, False, engine.OUT_FIDS | engine.OUT_OFFSETS, strategy)
    b.wait()
    c, off, fids = b.counts(), b.offsets(), b.fids()
    assert c.tolist() == [k, 2, 0, k] and b.total_hits == 2 * k + 2
    for i in (0, 3):
        seg = fids[int(off[i]):int(off[i]) + k]
        assert np.array_equal(np.sort(seg), np.arange(k, dtype=np.uint32))
    assert sorted(fids[int(off[1]):int(off[1]) + 2].tolist()) == [k, k + 1]


def test_capacity_replay_and_reuse():
    """More pairs than the initial buffer guess -> the emit step is replayed, results unchanged."""
    k = 400  # 400 nested intervals, every query hits all of them
    roots = {"chr_offsets": np.array([0, k], np.uint32), "start": np.arange(k, dtype=np.uint32),
             "end": (10_000 - np.arange(k)).astype(np.uint32), "fid": np.arange(k, dtype=np.uint32) * 3}
    regions = np.tile(np.array([[0, 1000, 2000]], np.uint32), (3000, 1))
    for strategy in STRATEGIES:
        _check(roots, regions, OverlapMode.Overlap, False, strategy=strategy)
    ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
    b = engine.QueryBatch(ix, 3000)
    for n in (3000, 10, 0, 2999):  # the same batch object, different sizes
        b.set_regions(regions[:n])
        b.run(OverlapMode.Overlap, False, engine.OUT_FIDS)
        b.wait()
        assert b.total_hits == n * k
        assert np.array_equal(b.counts(), np.full(n, k, np.uint32))


@pytest.mark.parametrize("nq,strategy", [(1_000_000, st) for st in STRATEGIES] +
                         [(10_000_000, engine.STRATEGY_FUSED), (10_000_000, engine.STRATEGY_WINDOWS), (12_500_000, engine.STRATEGY_SORTED)])
def test_full_size_c2_properties(nq, strategy):
    """BASELINE config sizes (configs[1]: 1 M regions x 63 k roots; configs[2]: 10 M; configs[3]: one GPU's 12.5 M
    share of 100 M): sampled oracle parity + size-independent properties (sum of counts == pairs; invert
    complements the mode predicate inside the hit set)."""
    roots = synth.gencode_like_roots(63000, seed=42)
    regions = synth.synth_bed(nq, seed=1001)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    b = engine.QueryBatch(ix, nq)
    b.set_regions(regions)
    res = {}
    for mode in OverlapMode:
        for inv in (False, True):
            b.run(mode, inv, engine.OUT_FIDS | engine.OUT_OFFSETS, strategy)
            b.wait()
            c = b.counts()
            assert int(c.sum(dtype=np.uint64)) == b.total_hits == int(b.offsets()[-1])
            res[(int(mode), inv)] = c
    ov = res[(2, False)]
    assert np.array_equal(res[(0, False)] + res[(0, True)], ov)
    assert np.array_equal(res[(1, False)] + res[(1, True)], ov)
    assert res[(2, True)].sum() == 0
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    sel = np.random.default_rng(7).choice(nq, size=20000, replace=False)
    for mode in OverlapMode:
        _, want_c = oix.query_features(regions[sel], int(mode), False)
        assert np.array_equal(res[(int(mode), False)][sel], want_c)


@pytest.mark.parametrize("strategy", [engine.STRATEGY_WINDOWS])
@pytest.mark.parametrize("mode", list(OverlapMode))
@pytest.mark.parametrize("invert", [False, True])
def test_slots_wide_empty_and_dense_regions_take_the_exact_lane(mode, invert, strategy):
    """The slot / window index answers regions of width <= wmax from a window's candidate list; everything else -- wide
    regions, start >= end, windows with more than kSlotMaxList candidates -- walks the sweep in its lane.  A mix
    of all of them in one batch, odd batch sizes (partial last thread / round), against the oracle."""
    rng = np.random.default_rng(11)
    roots = synth.gencode_like_roots(3000, seed=21, chroms=synth.SMALL2, fid_stride=5)
    # a dense cluster: 60 short roots inside 20 kb of the first seqid, and 30 nested long ones over it
    co, s, e, f = (roots[k].copy() for k in ("chr_offsets", "start", "end", "fid"))
    n0 = int(co[1])
    cs = np.sort(rng.integers(500_000, 520_000, 60)).astype(np.uint32)
    ce = (cs + rng.integers(50, 3000, 60)).astype(np.uint32)
    ls = np.sort(rng.integers(100_000, 500_000, 30)).astype(np.uint32)
    le = (ls + rng.integers(400_000, 3_000_000, 30)).astype(np.uint32)
    add_s, add_e = np.concatenate([cs, ls]), np.concatenate([ce, le])
    s = np.concatenate([s[:n0], add_s, s[n0:]])
    e = np.concatenate([e[:n0], add_e, e[n0:]])
    f = np.concatenate([f[:n0], f.max() + 5 * np.arange(1, 91, dtype=np.uint32), f[n0:]])
    co = co.copy()
    co[1:] += 90
    roots2 = dict(roots, chr_offsets=co, start=s, end=e, fid=f)
    for nq in (1, 3, 2047, 2049, 7001):
        regions = synth.synth_bed(nq, seed=nq, chroms=synth.SMALL2, width=(1, 9000), edge_frac=0.05, roots=roots2)
        k = rng.random(nq)
        wide = k < 0.15  # wider than any wmax
        regions[wide, 2] = regions[wide, 1] + rng.integers(20_000, 5_000_000, int(wide.sum()))
        flip = (k >= 0.15) & (k < 0.2)  # start >= end
        regions[flip, 2] = regions[flip, 1] - np.minimum(regions[flip, 1], rng.integers(0, 3, int(flip.sum())))
        dense = (k >= 0.2) & (k < 0.4)  # inside the cluster
        regions[dense, 0] = 0
        regions[dense, 1] = rng.integers(495_000, 525_000, int(dense.sum()))
        regions[dense, 2] = regions[dense, 1] + rng.integers(1, 4000, int(dense.sum()))
        _check(roots2, regions, mode, invert, soa=bool(nq & 1), strategy=strategy)


def test_launch_grid_follows_the_batches_in_flight():
    """(round 5) Block width and grid of a pair pass follow what else is in flight on the index: alone 1024-thread blocks, one per
    round; with ONE other batch in flight 512-thread blocks, a block per slot; from the third batch on ONE 512-thread block per CU
    (256), so that two batches' kernels are resident side by side (include/gffx_hip.h, gffx_hip_batch_block_count).  The results do
    not depend on any of it."""
    roots = synth.gencode_like_roots(20000, seed=6)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    regions = synth.synth_bed(600_000, seed=21, edge_frac=0.01, roots=roots)
    _, want_c = oix.query_features(regions, 2, False)
    bs = [engine.QueryBatch(ix, len(regions)) for _ in range(3)]
    for b in bs:
        b.set_regions(regions)
    flags = engine.OUT_FIDS | engine.OUT_SEGBASE
    bs[0].run(OverlapMode.Overlap, False, flags)
    bs[0].wait()
    assert (bs[0].block_threads, bs[0].block_count) == (1024, (len(regions) + 4095) // 4096)
    for b in bs:  # nobody waits in between: the second launch sees one busy batch, the third two
        b.run(OverlapMode.Overlap, False, flags)
    assert (bs[0].block_threads, bs[1].block_threads, bs[2].block_threads) == (1024, 512, 512)
    assert (bs[1].block_count, bs[2].block_count) == ((len(regions) + 2047) // 2048, 256)
    for b in bs:
        b.wait()
        assert np.array_equal(b.counts(), want_c)
    bs[1].run(OverlapMode.Overlap, False, flags)  # everything was waited for: alone again
    bs[1].wait()
    assert bs[1].block_threads == 1024 and np.array_equal(bs[1].counts(), want_c)
    for b in bs:
        b.close()
    ix.close()


def test_auto_moves_a_batch_of_wide_regions_off_the_narrow_form():
    """AUTO: a batch of mostly wide regions -- found by a sample of the rows the host hands over, or by a first waited pass that
    sent most regions to the exact sweep (tests/test_wide_form_gpu.py) -- runs on the mixed form of the window kernel (Overlap: round
    4's wide form; the other modes and the inverted passes: round 5); a batch of narrow regions stays where it is."""
    roots = synth.gencode_like_roots(150, seed=5, chroms=synth.SMALL2)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    wide = synth.synth_bed(5000, seed=1, chroms=synth.SMALL2, width=(200_000, 900_000))
    narrow = synth.synth_bed(5000, seed=2, chroms=synth.SMALL2, width=(100, 5000))
    b = engine.QueryBatch(ix, 5000)

    def one_pass(mode, want_c, inv=False):
        b.set_profiling(True)
        b.reset_profile()
        b.run(mode, inv, engine.OUT_FIDS | engine.OUT_OFFSETS)  # AUTO
        b.wait()
        b.set_profiling(False)
        assert np.array_equal(b.counts(), want_c)
        return [name for kid, name in engine.KERNEL_NAMES.items() if b.kernel_ms(kid)[1]], b.wide_form

    for regions, is_wide in ((wide, True), (narrow, False)):
        b.set_regions(regions)
        _, want_c = oix.query_features(regions, 2, False)
        assert one_pass(OverlapMode.Overlap, want_c) == (["k_join_pairs"], is_wide)  # (host regions: a sample of the widths decides at once)
        assert one_pass(OverlapMode.Overlap, want_c) == (["k_join_pairs"], is_wide)
        _, want_cc = oix.query_features(regions, 0, False)
        assert one_pass(OverlapMode.Contained, want_cc) == (["k_join_pairs"], is_wide)
        _, want_cr = oix.query_features(regions, 1, False)
        assert one_pass(OverlapMode.ContainsRegion, want_cr) == (["k_join_pairs"], is_wide)
        for mode in (0, 1):
            _, want_ci = oix.query_features(regions, mode, True)
            assert one_pass(OverlapMode(mode), want_ci, True) == (["k_join_pairs"], is_wide)
        assert one_pass(OverlapMode.Overlap, want_c) == (["k_join_pairs"], is_wide)
    b.close()
    ix.close()


def test_offsets32_and_bitmap_accumulation():
    """GFFX_OUT_OFFSETS32: the segment starts as u32 equal the u64 ones.  GFFX_OUT_BITMAP_KEEP: a caller that streams a
    BED file chunk by chunk through one batch ends with the unique roots of the whole file (intersect.rs:598-615)."""
    roots = synth.gencode_like_roots(5000, seed=3)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    regions = synth.synth_bed(30011, seed=8, edge_frac=0.02, roots=roots)
    b = engine.QueryBatch(ix, len(regions))
    b.set_regions(regions)
    for mode in OverlapMode:
        want_t, want_c = oix.query_features(regions, int(mode), False)
        b.run(mode, False, engine.OUT_FIDS | engine.OUT_OFFSETS | engine.OUT_OFFSETS32)
        b.wait()
        off, off32 = b.offsets(), b.offsets32()
        assert np.array_equal(b.counts(), want_c) and np.array_equal(off[:-1], off32.astype(np.uint64))
        f2 = b.fids()
        got = np.sort(np.concatenate([f2[int(o):int(o) + int(c)] for o, c in zip(off32[:500], want_c[:500])]))
        one_t, _ = oix.query_features(regions[:500], int(mode), False)
        assert np.array_equal(got, np.sort(one_t[:, 0]))
        # three chunks through one batch, the bitmap kept between them
        seen = np.zeros(0, dtype=np.uint32)
        for ci, chunk in enumerate(np.array_split(regions, 3)):
            b.set_regions(chunk)
            b.run(mode, False, engine.OUT_ROOT_BITMAP | (engine.OUT_BITMAP_KEEP if ci else 0))
            b.wait()
            ct, _ = oix.query_features(chunk, int(mode), False)
            seen = np.union1d(seen, ct[:, 0])
            assert np.array_equal(b.unique_roots(), seen)
        assert np.array_equal(seen, np.unique(want_t[:, 0]))
        b.set_regions(regions)
        # the CLI's pass: roots only, counts waived (GFFX_OUT_NO_COUNTS); several passes between two waits, one fold at the wait;
        # the pass's pair total still arrives (per-block sums)
        halves = np.array_split(regions, 2)
        for hi, half in enumerate(halves):
            b.set_regions(half)
            b.run(mode, False, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS | (engine.OUT_BITMAP_KEEP if hi else 0))
        b.wait()
        assert np.array_equal(b.unique_roots(), np.unique(want_t[:, 0]))
        assert b.total_hits == len(oix.query_features(halves[1], int(mode), False)[0])
        assert b.kept_pairs_accumulated == len(want_t)  # ... and the kept pairs of BOTH passes, summed per block on the device
        with pytest.raises(Exception):
            b.counts()  # (not requested)
        b.set_regions(regions)
    b.close()
    ix.close()


def test_knobs_are_read_once_and_set_through_the_api(monkeypatch):
    """GFFX_HIP_* are read when an index / a batch is created, never by a launch; a batch's change through set_option; the values that
    are not defaults are reported (bench.py's config.knobs, the CLI's --stats-json)."""
    roots = synth.gencode_like_roots(3000, seed=5)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    regions = synth.synth_bed(600_000, seed=6)
    monkeypatch.setenv("GFFX_HIP_WIN_FILTER_KB", "8")
    monkeypatch.setenv("GFFX_HIP_WIN_THREADS", "512")
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    b = engine.QueryBatch(ix, len(regions))
    monkeypatch.delenv("GFFX_HIP_WIN_FILTER_KB")
    monkeypatch.setenv("GFFX_HIP_WIN_THREADS", "1024")  # (too late for `b`)
    assert ix.options() == {"GFFX_HIP_WIN_FILTER_KB": 8} and b.options() == {"GFFX_HIP_WIN_THREADS": 512}
    b.set_regions(regions)
    b.run(OverlapMode.Overlap, False, engine.OUT_FIDS | engine.OUT_SEGBASE)
    b.wait()
    assert b.block_threads == 512
    pairs = b.total_hits
    b.set_option("win_threads", 1024)
    b.run(OverlapMode.Overlap, False, engine.OUT_FIDS | engine.OUT_SEGBASE)
    b.wait()
    assert b.block_threads == 1024 and b.total_hits == pairs and b.options() == {"GFFX_HIP_WIN_THREADS": 1024}
    b.set_option("GFFX_HIP_WIN_THREADS", 0)
    b.run(OverlapMode.Overlap, False, engine.OUT_FIDS | engine.OUT_SEGBASE)
    b.wait()
    assert b.block_threads == 1024 and b.options() == {}  # the engine's choice for 600 k regions alone
    for name, value in (("no_such_knob", 1), ("WIN_THREADS", 4096), ("WIN_WIDE", 3)):
        with pytest.raises(Exception):
            b.set_option(name, value)
    b.close()
    ix.close()


@pytest.mark.parametrize("second", [engine.STRATEGY_FUSED, engine.STRATEGY_DIRECT, engine.STRATEGY_SORTED, "empty"])
def test_unwaited_windows_root_pass_does_not_leak_into_the_next_strategys_roots(second):
    """A root pass of the windows strategy leaves its roots in per-block slabs that are folded at the wait.  A NEW set of roots
    (no GFFX_OUT_BITMAP_KEEP) asked from another strategy -- or from an empty batch -- before anybody waited must not get the
    earlier pass's slabs folded into it (round 4's advisor finding: the slab state was only reset inside run_windows)."""
    roots = synth.gencode_like_roots(4000, seed=21)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    regions = synth.synth_bed(20000, seed=22, edge_frac=0.02, roots=roots)
    first, rest = regions[:15000], regions[15000:15400]
    b = engine.QueryBatch(ix, len(regions))
    b.set_regions(first)
    b.run(OverlapMode.Overlap, False, engine.OUT_ROOT_BITMAP, engine.STRATEGY_WINDOWS)  # not waited
    if second == "empty":
        b.set_regions(rest[:0])
        b.run(OverlapMode.Contained, False, engine.OUT_ROOT_BITMAP)
        b.wait()
        assert len(b.unique_roots()) == 0
    else:
        b.set_regions(rest)
        b.run(OverlapMode.Contained, False, engine.OUT_ROOT_BITMAP, second)
        b.wait()
        want_t, _ = oix.query_features(rest, int(OverlapMode.Contained), False)
        assert np.array_equal(b.unique_roots(), np.unique(want_t[:, 0]))
    # ... and the windows strategy still accumulates across passes when asked to
    b.set_regions(first)
    b.run(OverlapMode.Overlap, False, engine.OUT_ROOT_BITMAP, engine.STRATEGY_WINDOWS)
    b.set_regions(rest)
    b.run(OverlapMode.Overlap, False, engine.OUT_ROOT_BITMAP | engine.OUT_BITMAP_KEEP, engine.STRATEGY_WINDOWS)
    b.wait()
    want_t, _ = oix.query_features(np.concatenate([first, rest]), int(OverlapMode.Overlap), False)
    assert np.array_equal(b.unique_roots(), np.unique(want_t[:, 0]))
    b.close()
    ix.close()


@pytest.mark.parametrize("shift", [0, 1, 3])
def test_device_resident_regions_aligned_and_unaligned(shift):
    """The zero-copy path of bench.py: three device-resident u32 columns borrowed by pointer (allocated here with
    hipMalloc through ctypes).  the window kernels read them
    with 16-byte loads when it can; columns that start `shift` elements into an allocation are not 16-byte aligned and
    take the scalar loads."""
    import ctypes

    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    roots = synth.gencode_like_roots(2000, seed=31, chroms=synth.SMALL2)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    regions = synth.synth_bed(10_001, seed=32, chroms=synth.SMALL2, width=(1, 4000), edge_frac=0.05, roots=roots)
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    cols = []
    for c in range(3):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), 4 * (len(regions) + 8)) == 0
        col = np.ascontiguousarray(regions[:, c])
        assert hip.hipMemcpy(p.value + 4 * shift, col.ctypes.data, col.nbytes, 1) == 0  # hipMemcpyHostToDevice
        cols.append(p)
    b = engine.QueryBatch(ix, len(regions))
    b.set_regions_device(*(p.value + 4 * shift for p in cols), len(regions))
    by_chr = np.argsort(regions[:, 0], kind="stable")
    for strategy in STRATEGIES:
        for mode in OverlapMode:
            want_t, want_c = oix.query_features(regions, int(mode), False)
            b.run(mode, False, engine.OUT_FIDS | engine.OUT_OFFSETS, strategy)
            b.wait()
            c, off, fids = b.counts(), b.offsets(), b.fids()
            assert np.array_equal(c, want_c) and b.total_hits == len(want_t)
            wc = want_c.astype(np.int64)
            qid = np.repeat(np.arange(len(regions), dtype=np.int64), wc)
            within = np.arange(len(qid), dtype=np.int64) - np.repeat(np.cumsum(wc) - wc, wc)
            got = np.stack([qid, fids[off[:-1].astype(np.int64)[qid] + within].astype(np.int64)], axis=1)
            want = np.stack([np.repeat(by_chr, wc[by_chr]), want_t[:, 0].astype(np.int64)], axis=1)
            srt = lambda a: a[np.lexsort((a[:, 1], a[:, 0]))]  # noqa: E731
            assert np.array_equal(srt(got), srt(want))
    b.close()
    ix.close()
    for p in cols:
        hip.hipFree(p)


def test_warmup_reserve_hits_and_device_result_pointers():
    """The optional / zero-copy corners of the ABI: gffx_hip_warmup, a pair reservation that makes the capacity replay
    unnecessary, and the device addresses of the results (read back here with hipMemcpy)."""
    import ctypes

    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    engine.warmup(0)
    roots = synth.gencode_like_roots(500, seed=41, chroms=synth.SMALL2)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    regions = synth.synth_bed(3000, seed=42, chroms=synth.SMALL2, width=(1, 200_000))  # ~30 pairs per region
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    want_t, want_c = oix.query_features(regions, 2, False)
    assert len(want_t) > 4 * len(regions)  # more than the default first guess of 2 pairs per region
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    for reserve in (0, len(want_t) + 10):
        b = engine.QueryBatch(ix, len(regions))
        assert b.n_queries == 0
        b.set_regions(regions)
        assert b.n_queries == len(regions)
        if reserve:
            b.reserve_hits(reserve)
        b.run(OverlapMode.Overlap, False, engine.OUT_FIDS | engine.OUT_TRIPLES | engine.OUT_OFFSETS)
        b.wait()
        assert b.total_hits == len(want_t) and np.array_equal(b.counts(), want_c)
        d_counts, d_fids, d_triples = b.device_pointers()
        assert d_counts and d_fids and d_triples
        host = np.empty(len(regions), np.uint32)
        assert hip.hipMemcpy(host.ctypes.data, d_counts, host.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        assert np.array_equal(host, want_c)
        hf = np.empty(len(want_t), np.uint32)
        assert hip.hipMemcpy(hf.ctypes.data, d_fids, hf.nbytes, 2) == 0
        assert np.array_equal(hf, b.fids())
        b.close()
    ix.close()


@pytest.mark.parametrize("n_roots", [300_000, 800_000])
def test_many_roots_bitmap_beyond_the_default_lds(n_roots, monkeypatch):
    """The root bitmap of a root-bitmap pass lives in LDS while it fits: 300 k roots need more than the default 64 KB of
    dynamic LDS (the kernel opts in), 800 k roots do not fit at all (test-before-set global atomics)."""
    rng = np.random.default_rng(n_roots)
    n_chr = 3
    per = n_roots // n_chr
    co = np.arange(n_chr + 1, dtype=np.uint32) * per
    s = np.concatenate([np.sort(rng.integers(0, 400_000_000, per)) for _ in range(n_chr)]).astype(np.uint32)
    e = (s + rng.integers(1, 3000, len(s))).astype(np.uint32)
    f = (rng.permutation(len(s)) * 2 + 1).astype(np.uint32)
    regions = np.stack([rng.integers(0, n_chr, 60_000), rng.integers(0, 400_000_000, 60_000), np.zeros(60_000, np.int64)], axis=1)
    regions[:, 2] = regions[:, 1] + rng.integers(1, 9000, len(regions))
    regions = regions.astype(np.uint32)
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    b = engine.QueryBatch(ix, len(regions))
    b.set_regions(regions)
    for mode in (OverlapMode.Overlap, OverlapMode.Contained):
        want_t, want_c = oix.query_features(regions, int(mode), False)
        b.run(mode, False, engine.OUT_ROOT_BITMAP)
        b.wait()
        assert np.array_equal(b.counts(), want_c) and b.total_hits == len(want_t)
        assert np.array_equal(b.unique_roots(), np.unique(want_t[:, 0]))
        b.run(mode, False, engine.OUT_FIDS | engine.OUT_OFFSETS32 | engine.OUT_ROOT_BITMAP)
        b.wait()
        assert np.array_equal(np.sort(b.fids()), np.sort(want_t[:, 0])) and np.array_equal(b.unique_roots(), np.unique(want_t[:, 0]))
    # the wide form of the root pass (a region's run of roots is a run of bits: in LDS, or straight in the batch's bitmap), after
    # the narrow passes above on the same batch, then accumulated over two halves
    b.set_option("WIN_WIDE", 2)  # (a batch reads its knobs when it is created; later changes go through set_option)
    wide = regions[:20_000].copy()
    wide[:, 2] = wide[:, 1] + rng.integers(1, 150_000, len(wide)).astype(np.uint32)
    want_t, want_c = oix.query_features(wide, int(OverlapMode.Overlap), False)
    b.set_regions(wide)
    b.run(OverlapMode.Overlap, False, engine.OUT_ROOT_BITMAP, engine.STRATEGY_WINDOWS)
    b.wait()
    assert b.wide_form and np.array_equal(b.counts(), want_c) and np.array_equal(b.unique_roots(), np.unique(want_t[:, 0]))
    b.set_regions(wide[:10_000])
    b.run(OverlapMode.Overlap, False, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, engine.STRATEGY_WINDOWS)
    b.set_regions(wide[10_000:])
    b.run(OverlapMode.Overlap, False, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS | engine.OUT_BITMAP_KEEP, engine.STRATEGY_WINDOWS)
    b.wait()
    assert b.wide_form and np.array_equal(b.unique_roots(), np.unique(want_t[:, 0]))
    b.close()
    ix.close()


def test_window_directory_is_coarsened_not_refused(monkeypatch):
    """An index whose window directory would exceed the 32-bit line addressing gets wider windows (longer lists, more of them
    deferred or dense) instead of an error: forced here with a tiny limit."""
    monkeypatch.setenv("GFFX_HIP_WIN_MAX_LINES", "1500")
    roots = synth.gencode_like_roots(6000, seed=31)
    regions = synth.synth_bed(30_000, seed=32, edge_frac=0.02, roots=roots)
    for mode in OverlapMode:
        _check(roots, regions, mode, False, strategy=engine.STRATEGY_WINDOWS)


def test_wide_block_variant(monkeypatch):
    """k_join_pairs<..., T = 1024> (one block per CU, rounds of 4096 regions): forced through GFFX_HIP_WIN_THREADS on the small
    and ragged cases, then the engine's own choice -- a 0.5-2.5 M-region pass that runs ALONE takes it, a pass launched while
    another batch of the index is in flight takes 512-thread blocks -- with full parity of the 1 M-region pass either way."""
    monkeypatch.setenv("GFFX_HIP_WIN_THREADS", "1024")
    roots = synth.gencode_like_roots(300, seed=3, chroms=synth.SMALL2)
    for n in (1, 3, 4095, 4096, 4097, 8200, 20000):
        regions = synth.synth_bed(n, seed=11 + n, chroms=synth.SMALL2, width=(10, 200000), edge_frac=0.2, roots=roots)
        for mode in OverlapMode:
            for inv in (False, True):
                _check(roots, regions, mode, inv, soa=bool(n & 1), strategy=engine.STRATEGY_WINDOWS)
    big = synth.gencode_like_roots(63000, seed=42)
    regions = synth.synth_bed(100_003, seed=1001, edge_frac=0.001, roots=big)
    for mode in OverlapMode:
        _check(big, regions, mode, False, strategy=engine.STRATEGY_WINDOWS)
    monkeypatch.delenv("GFFX_HIP_WIN_THREADS")
    # the engine's choice
    co, s, e, f = big["chr_offsets"], big["start"], big["end"], big["fid"]
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    regions = synth.synth_bed(1_000_000, seed=1001)
    want_t, want_c = oix.query_features(regions, 2, False)
    flags = engine.OUT_FIDS | engine.OUT_OFFSETS32
    a, other = engine.QueryBatch(ix, len(regions)), engine.QueryBatch(ix, len(regions))
    for b in (a, other):
        b.set_regions(regions)

    def equal_to_oracle(b):
        b.wait()
        off, fids = b.offsets32().astype(np.int64), b.fids()
        assert b.total_hits == len(want_t) and np.array_equal(b.counts(), want_c)
        wc = want_c.astype(np.int64)
        qid = np.repeat(np.arange(len(regions), dtype=np.int64), wc)
        within = np.arange(len(qid), dtype=np.int64) - np.repeat(np.cumsum(wc) - wc, wc)
        got = (qid << 32) | fids[off[qid] + within].astype(np.int64)
        by_chr = np.argsort(regions[:, 0], kind="stable")
        want = (np.repeat(by_chr, wc[by_chr]).astype(np.int64) << 32) | want_t[:, 0].astype(np.int64)
        assert np.array_equal(np.sort(got), np.sort(want))

    a.run(OverlapMode.Overlap, False, flags)  # alone
    assert a.block_threads == 1024
    equal_to_oracle(a)
    other.run(OverlapMode.Overlap, False, flags)  # (not synchronised with: in flight as far as the engine knows)
    a.run(OverlapMode.Overlap, False, flags)
    assert a.block_threads == 512
    equal_to_oracle(a)
    equal_to_oracle(other)
    a.run(OverlapMode.Overlap, False, flags)  # alone again
    assert a.block_threads == 1024
    a.wait()
    a.close()
    other.close()


def test_two_cloned_indexes_driven_from_two_threads():
    """Multi-device bookkeeping before the first real 8-GPU run: two copies of the index made with gffx_hip_index_clone (on a
    1-GPU box both land on device 0; on a node with more GPUs the second goes to device 1), each with its own batch, each
    running 1 M-region pair passes with 1024-thread blocks (more than 64 KB of dynamic LDS: the per-(kernel, device) opt-in)
    from its own host thread at the same time -- full parity of both against the oracle."""
    import threading

    big = synth.gencode_like_roots(63000, seed=42)
    co, s, e, f = big["chr_offsets"], big["start"], big["end"], big["fid"]
    ix0 = engine.TreeIndexData.from_roots(co, s, e, f)
    ndev = engine.device_count()
    clones = [ix0.clone(0), ix0.clone(1 % ndev)]
    assert [c.device for c in clones] == [0, 1 % ndev] and all(c.n_roots == ix0.n_roots for c in clones)
    ix0.close()  # (the clones own their device arrays)
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    regions = [synth.synth_bed(1_000_000, seed=1001), synth.synth_bed(1_000_000, seed=77)]
    want = [oix.query_features(r, 2, False) for r in regions]
    flags = engine.OUT_FIDS | engine.OUT_SEGBASE
    errors = []

    def drive(k):
        try:
            b = engine.QueryBatch(clones[k], len(regions[k]))
            b.set_regions(regions[k])
            for _ in range(3):
                b.run(OverlapMode.Overlap, False, flags)
                b.wait()
            want_t, want_c = want[k]
            assert b.total_hits == len(want_t) and np.array_equal(b.counts(), want_c)
            off, fids = b.offsets_from_segbase().astype(np.int64), b.fids()
            wc = want_c.astype(np.int64)
            qid = np.repeat(np.arange(len(wc), dtype=np.int64), wc)
            within = np.arange(len(qid), dtype=np.int64) - np.repeat(np.cumsum(wc) - wc, wc)
            got = (qid << 32) | fids[off[qid] + within].astype(np.int64)
            by_chr = np.argsort(regions[k][:, 0], kind="stable")
            exp = (np.repeat(by_chr, wc[by_chr]).astype(np.int64) << 32) | want_t[:, 0].astype(np.int64)
            assert np.array_equal(np.sort(got), np.sort(exp))
            b.close()
        except Exception as exc:  # noqa: BLE001 (reported by the main thread)
            errors.append((k, repr(exc)))

    monkey = os.environ.get("GFFX_HIP_WIN_THREADS")
    os.environ["GFFX_HIP_WIN_THREADS"] = "1024"
    try:
        threads = [threading.Thread(target=drive, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        if monkey is None:
            os.environ.pop("GFFX_HIP_WIN_THREADS", None)
        else:
            os.environ["GFFX_HIP_WIN_THREADS"] = monkey
    for c in clones:
        c.close()
    assert not errors, errors
