"""Ordered input (round 6).  A BED file is usually sorted by position: the four consecutive regions of a thread then sit in the same
windows, a gene-dense stretch lands on one wave, rounds overflow their strip, list tails cluster (the continuation lines' in-step path,
DESIGN 4.0f), and a launch's tail is taken by ticket.  Whatever the order of the batch -- sorted by (seqid, start), by end, reversed,
sorted per seqid only, half sorted, clustered on gene-dense stretches, a few strays among sorted rows -- every region's pairs and the
unique roots equal the oracle's, bit for bit, in every mode, for the pair passes and for the root passes.  (The file was written for the
LDS-staged lines of the round's first experiment, which lost and are gone: DESIGN 10.1; the orders it checks stayed.)
Reference: utils/tree.rs:98-121, commands/intersect.rs:139-165.
"""
import numpy as np
import pytest

from gffx_amd import engine, synth
from gffx_amd.engine import OverlapMode
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def _orders(regions, rng):
    n = len(regions)
    by_start = np.lexsort((regions[:, 1], regions[:, 0]))
    by_end = np.lexsort((regions[:, 2], regions[:, 0]))
    out = {"sorted": regions[by_start], "by_end": regions[by_end], "reversed": regions[by_start][::-1]}
    # sorted inside every seqid, the seqids in random order
    chrs = rng.permutation(np.unique(regions[:, 0]))
    srt = regions[by_start]
    out["per_seqid"] = np.concatenate([srt[srt[:, 0] == c] for c in chrs])
    half = regions.copy()
    half[: n // 2] = regions[: n // 2][np.lexsort((regions[: n // 2, 1], regions[: n // 2, 0]))]
    out["half_sorted"] = half
    strays = srt.copy()
    at = rng.choice(n, size=max(1, n // 50), replace=False)
    strays[at] = regions[rng.choice(n, size=len(at))]
    out["sorted_with_strays"] = strays
    return {k: np.ascontiguousarray(v) for k, v in out.items()}


def _check_batch(oix, ix, regions, mode, invert, threads):
    want_t, want_c = oix.query_features(regions, int(mode), invert)
    b = engine.QueryBatch(ix, len(regions))
    b.set_option("WIN_THREADS", threads)
    b.set_regions(regions)
    b.run(mode, invert, engine.OUT_FIDS | engine.OUT_OFFSETS, engine.STRATEGY_WINDOWS)
    b.wait()
    c, off, f = b.counts(), b.offsets(), b.fids()
    assert np.array_equal(c, want_c)
    wc = want_c.astype(np.int64)
    qid = np.repeat(np.arange(len(regions), dtype=np.int64), wc)
    within = np.arange(len(qid), dtype=np.int64) - np.repeat(np.cumsum(wc) - wc, wc)
    got = np.stack([qid, f[off[:-1].astype(np.int64)[qid] + within].astype(np.int64)], axis=1)
    by_chr = np.argsort(regions[:, 0], kind="stable")
    want = np.stack([np.repeat(by_chr, wc[by_chr]), want_t[:, 0].astype(np.int64)], axis=1)
    order = lambda a: a[np.lexsort((a[:, 1], a[:, 0]))]  # noqa: E731
    assert np.array_equal(order(got), order(want))
    # the pass bench.py times (segment bases) and the pass the CLI runs (the unique roots alone)
    b.run(mode, invert, engine.OUT_FIDS | engine.OUT_SEGBASE, engine.STRATEGY_WINDOWS)
    b.wait()
    c3 = b.counts()
    off3 = b.offsets_from_segbase(c3).astype(np.int64)
    got3 = np.stack([qid, b.fids()[off3[qid] + within].astype(np.int64)], axis=1)
    assert np.array_equal(c3, want_c) and np.array_equal(order(got3), order(want))
    b.run(mode, invert, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, engine.STRATEGY_WINDOWS)
    b.wait()
    assert np.array_equal(b.unique_roots(), np.unique(want_t[:, 0]))
    assert b.total_hits == len(want_t)
    b.close()


@pytest.mark.parametrize("threads", [512, 1024])
@pytest.mark.parametrize("mode,invert", [(OverlapMode.Overlap, False), (OverlapMode.Contained, False), (OverlapMode.Contained, True),
                                         (OverlapMode.ContainsRegion, False), (OverlapMode.ContainsRegion, True)])
def test_ordered_batches_on_the_gencode_shaped_index(mode, invert, threads):
    roots = synth.gencode_like_roots(63000, seed=42)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    rng = np.random.default_rng(61)
    # 300 k regions: a sorted wave round spans ~80 lines; 60 k regions on three seqids: ~30 lines (every lane of a wave in a few windows);
    # widths up to 20 k: some rows are wider than the lines answer and take the sweep from inside such a round
    big = synth.synth_bed(300_000, seed=6100, edge_frac=0.01, roots=roots, width=(1, 20_000))
    dense = synth.synth_bed(60_000, seed=6101, edge_frac=0.3, roots=roots, width=(10, 3000))
    dense = dense[dense[:, 0] < 3]
    for name, regs in list(_orders(big, rng).items()) + [("dense_" + k, v) for k, v in _orders(dense, rng).items()]:
        _check_batch(oix, ix, regs, mode, invert, threads)
    ix.close()


@pytest.mark.parametrize("seed", range(4))
def test_ordered_batches_on_small_dense_indexes(seed):
    """tiny coordinate ranges: split windows everywhere (a sorted wave's lines are SUB-lines), list tails of every length (continuation
    lines in step, walked lists, dense windows), empty and reversed rows inside such rounds"""
    rng = np.random.default_rng(700 + seed)
    n_chr = int(rng.integers(1, 4))
    span = int(rng.choice([3000, 200_000, 5_000_000]))
    per = rng.integers(50, 4000, n_chr)
    co = np.concatenate([[0], np.cumsum(per)]).astype(np.uint32)
    n = int(co[-1])
    start = rng.integers(0, span, n).astype(np.uint32)
    end = (start + rng.integers(1, max(2, span // int(rng.choice([3, 50, 1000]))), n)).astype(np.uint32)
    order = np.concatenate([np.argsort(start[co[i]:co[i + 1]], kind="stable") + co[i] for i in range(n_chr)])
    roots = {"chr_offsets": co, "start": start[order], "end": end[order], "fid": rng.permutation(n).astype(np.uint32) * 3}
    oix = ob.OracleIndex.from_roots(co, roots["start"], roots["end"], roots["fid"])
    ix = engine.TreeIndexData.from_roots(co, roots["start"], roots["end"], roots["fid"])
    nq = int(rng.integers(3000, 40_000))
    qs = rng.integers(0, span + 5, nq)
    qe = qs + rng.integers(-2, max(3, span // int(rng.choice([10, 200]))), nq)  # (includes empty and reversed rows)
    regions = np.stack([rng.integers(0, n_chr, nq), qs, np.maximum(qe, 0)], axis=1).astype(np.uint32)
    for name, regs in _orders(regions, rng).items():
        for mode in OverlapMode:
            _check_batch(oix, ix, regs, mode, False, int(rng.choice([512, 1024])))
    ix.close()
