"""Continuation lines (round 6): a list of 5 .. 7 roots keeps its entries 3 .. n - 1 once more in the window line's own
packed format (gffx_device.hpp), and k_join_pairs / k_join_roots test that instead of walking the list's records -- the thread's four
regions in step when some lane of the wave has two such lists or more (ordered input), else the lane's one region picked.  The index
here is built so that such lists are the rule: clusters of 5, 6 and 7 roots with nested or identical coordinates (every window and
sub-window a cluster touches lists all of them), next to clusters of 4 (no tail), 8 and 12 (the walk) and 40 (dense: the sweep).  Every
region's pairs, the segment bases, the triples (the position copy of the lines) and the unique roots equal the oracle's, bit for bit, in
every mode, inverted or not, for sorted and for shuffled batches, both block widths, lone launches and a launch that serves a group.
Reference: utils/tree.rs:98-121, commands/intersect.rs:139-165.
"""
import numpy as np
import pytest

from gffx_amd import engine
from gffx_amd.engine import OverlapMode
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def _clustered_roots(rng, n_chr=3, clusters=1500, span=60_000_000):
    co, start, end = [0], [], []
    for _ in range(n_chr):
        s_all, e_all = [], []
        at = np.sort(rng.integers(0, span, clusters))
        for a in at:
            k = int(rng.choice([1, 2, 4, 5, 5, 6, 6, 7, 7, 8, 12, 40], p=[.2, .1, .1, .12, .12, .1, .1, .05, .05, .03, .02, .01]))
            width = int(rng.integers(300, 30_000))
            if rng.random() < 0.5:  # identical coordinates (ties) ...
                s = np.full(k, a)
                e = np.full(k, a + width)
            else:  # ... or nested ones
                s = a + np.sort(rng.integers(0, max(2, width // 4), k))
                e = a + width - rng.integers(0, max(2, width // 4), k)
            s_all.append(s), e_all.append(np.maximum(e, s + 1))
        s_all, e_all = np.concatenate(s_all), np.concatenate(e_all)
        o = np.argsort(s_all, kind="stable")
        start.append(s_all[o]), end.append(e_all[o])
        co.append(co[-1] + len(o))
    start, end = np.concatenate(start).astype(np.uint32), np.concatenate(end).astype(np.uint32)
    return np.array(co, np.uint32), start, end, (rng.permutation(len(start)).astype(np.uint32) * 5 + 1)


def _regions_on_clusters(rng, co, start, end, n):
    """regions that mostly land on roots (so that the lists matter), a few anywhere; empty and reversed rows among them"""
    chr_of = np.repeat(np.arange(len(co) - 1), np.diff(co))
    pick = rng.integers(0, len(start), n)
    qs = start[pick].astype(np.int64) + rng.integers(-3000, 3000, n)
    qe = qs + rng.integers(-1, 4000, n)
    anywhere = rng.random(n) < 0.1
    qs[anywhere] = rng.integers(0, 60_000_000, int(anywhere.sum()))
    qe[anywhere] = qs[anywhere] + rng.integers(1, 4000, int(anywhere.sum()))
    return np.stack([chr_of[pick], np.maximum(qs, 0), np.maximum(qe, 0)], axis=1).astype(np.uint32)


def _check(oix, ix, regions, mode, invert, threads):
    want_t, want_c = oix.query_features(regions, int(mode), invert)
    b = engine.QueryBatch(ix, len(regions))
    b.set_option("WIN_THREADS", threads)
    b.set_regions(regions)
    wc = want_c.astype(np.int64)
    qid = np.repeat(np.arange(len(regions), dtype=np.int64), wc)
    within = np.arange(len(qid), dtype=np.int64) - np.repeat(np.cumsum(wc) - wc, wc)
    by_chr = np.argsort(regions[:, 0], kind="stable")
    want = np.stack([np.repeat(by_chr, wc[by_chr]), want_t[:, 0].astype(np.int64)], axis=1)
    order = lambda a: a[np.lexsort((a[:, 1], a[:, 0]))]  # noqa: E731
    # pairs with per-region offsets; the pass bench.py times (segment bases); the unique roots alone (what the CLI runs)
    b.run(mode, invert, engine.OUT_FIDS | engine.OUT_OFFSETS, engine.STRATEGY_WINDOWS)
    b.wait()
    c, off, f = b.counts(), b.offsets(), b.fids()
    assert np.array_equal(c, want_c)
    got = np.stack([qid, f[off[:-1].astype(np.int64)[qid] + within].astype(np.int64)], axis=1)
    assert np.array_equal(order(got), order(want))
    b.run(mode, invert, engine.OUT_FIDS | engine.OUT_SEGBASE, engine.STRATEGY_WINDOWS)
    b.wait()
    c3 = b.counts()
    off3 = b.offsets_from_segbase(c3).astype(np.int64)
    got3 = np.stack([qid, b.fids()[off3[qid] + within].astype(np.int64)], axis=1)
    assert np.array_equal(c3, want_c) and np.array_equal(order(got3), order(want))
    b.run(mode, invert, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, engine.STRATEGY_WINDOWS)
    b.wait()
    assert np.array_equal(b.unique_roots(), np.unique(want_t[:, 0])) and b.total_hits == len(want_t)
    # triples: the lines' position copy (and the continuation lines' position record)
    b.run(mode, invert, engine.OUT_FIDS | engine.OUT_TRIPLES | engine.OUT_OFFSETS, engine.STRATEGY_WINDOWS)
    b.wait()
    t = b.triples().astype(np.int64)
    wt = want_t.astype(np.int64)
    key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]  # noqa: E731
    assert t.shape == wt.shape and np.array_equal(key(t), key(wt))
    b.close()


@pytest.mark.parametrize("threads", [512, 1024])
@pytest.mark.parametrize("mode,invert", [(OverlapMode.Overlap, False), (OverlapMode.Contained, False), (OverlapMode.Contained, True),
                                         (OverlapMode.ContainsRegion, False), (OverlapMode.ContainsRegion, True)])
def test_lists_of_five_to_seven_roots(mode, invert, threads):
    rng = np.random.default_rng(6500 + 7 * int(mode) + int(invert))
    co, start, end, fid = _clustered_roots(rng)
    oix = ob.OracleIndex.from_roots(co, start, end, fid)
    ix = engine.TreeIndexData.from_roots(co, start, end, fid)
    regions = _regions_on_clusters(rng, co, start, end, 120_000)
    by_start = np.lexsort((regions[:, 1], regions[:, 0]))
    for regs in (regions, np.ascontiguousarray(regions[by_start]), np.ascontiguousarray(regions[by_start][::-1]), regions[:4099], regions[:1]):
        _check(oix, ix, np.ascontiguousarray(regs), mode, invert, threads)
    ix.close()


def test_lists_of_five_to_seven_roots_in_a_group_launch():
    rng = np.random.default_rng(6599)
    co, start, end, fid = _clustered_roots(rng)
    oix = ob.OracleIndex.from_roots(co, start, end, fid)
    ix = engine.TreeIndexData.from_roots(co, start, end, fid)
    batches, wants, regs = [], [], []
    for i in range(6):
        regions = _regions_on_clusters(rng, co, start, end, 30_000 + 4000 * i)
        if i % 2:
            regions = np.ascontiguousarray(regions[np.lexsort((regions[:, 1], regions[:, 0]))])
        b = engine.QueryBatch(ix, len(regions))
        b.set_regions(regions)
        batches.append(b), wants.append(oix.query_features(regions, int(OverlapMode.Overlap), False)), regs.append(regions)
    assert engine.batches_plan(batches)[1] >= 2  # (one launch serves several of them)
    order = lambda a: a[np.lexsort((a[:, 1], a[:, 0]))]  # noqa: E731
    for _ in range(2):
        engine.run_batches(batches, OverlapMode.Overlap, False, engine.OUT_FIDS | engine.OUT_SEGBASE, engine.STRATEGY_WINDOWS)
        for b, (want_t, want_c), regions in zip(batches, wants, regs):
            b.wait()
            c = b.counts()
            assert np.array_equal(c, want_c) and b.total_hits == len(want_t)
            off = b.offsets_from_segbase(c).astype(np.int64)
            wcn = want_c.astype(np.int64)
            qid = np.repeat(np.arange(len(c), dtype=np.int64), wcn)
            within = np.arange(len(qid), dtype=np.int64) - np.repeat(np.cumsum(wcn) - wcn, wcn)
            got = np.stack([qid, b.fids()[off[qid] + within].astype(np.int64)], axis=1)
            by_chr = np.argsort(regions[:, 0], kind="stable")
            want = np.stack([np.repeat(by_chr, wcn[by_chr]), want_t[:, 0].astype(np.int64)], axis=1)
            assert np.array_equal(order(got), order(want))
    for b in batches:
        b.close()
    ix.close()


@pytest.mark.parametrize("mode,invert", [(OverlapMode.Overlap, False), (OverlapMode.Contained, True), (OverlapMode.ContainsRegion, False)])
def test_windows_without_a_split_level(mode, invert, monkeypatch):
    """GFFX_HIP_WIN_SPLIT=0: the windows' OWN lists continue in win_spill (no sub-lines), so their continuation lines -- built by
    build_window_index_at, not by build_window_splits -- are what the kernels read; also with a small LDS filter"""
    monkeypatch.setenv("GFFX_HIP_WIN_SPLIT", "0")
    monkeypatch.setenv("GFFX_HIP_WIN_FILTER_KB", "8")
    rng = np.random.default_rng(6580 + int(mode))
    co, start, end, fid = _clustered_roots(rng, n_chr=2, clusters=900)
    oix = ob.OracleIndex.from_roots(co, start, end, fid)
    ix = engine.TreeIndexData.from_roots(co, start, end, fid)
    monkeypatch.delenv("GFFX_HIP_WIN_SPLIT")
    monkeypatch.delenv("GFFX_HIP_WIN_FILTER_KB")
    assert ix.options() == {"GFFX_HIP_WIN_SPLIT": 0, "GFFX_HIP_WIN_FILTER_KB": 8}
    regions = _regions_on_clusters(rng, co, start, end, 60_000)
    by_start = np.lexsort((regions[:, 1], regions[:, 0]))
    for regs in (regions, np.ascontiguousarray(regions[by_start])):
        _check(oix, ix, regs, mode, invert, 1024)
    ix.close()
