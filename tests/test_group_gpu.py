"""Launches that serve several batches at once (round 6: gffx_hip_batches_run_n hands distinct batches of one index to the windows
strategy together -- every batch a share of one launch's blocks, its rounds taken by ticket): every batch's results equal the
oracle's, bit for bit, exactly as after single `run` calls -- whatever the batches' sizes, whatever ran on a batch's own stream
before or after, for every pass kind, mode and form.  Reference semantics: utils/tree.rs:98-121, commands/intersect.rs:139-165.
"""
import numpy as np
import pytest

from gffx_amd import engine, synth
from gffx_amd.engine import OverlapMode
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def _roots():
    return synth.gencode_like_roots(63000, seed=42)


def _oracle(roots):
    return ob.OracleIndex.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])


def _index(roots):
    return engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])


def _pairs_of(b, regions, counts, offsets=None):
    """(region, root_fid) rows of a waited pass, from per-region offsets or from the segment bases"""
    wc = counts.astype(np.int64)
    off = (b.offsets()[:-1] if offsets is None else offsets).astype(np.int64)
    qid = np.repeat(np.arange(len(regions), dtype=np.int64), wc)
    within = np.arange(len(qid), dtype=np.int64) - np.repeat(np.cumsum(wc) - wc, wc)
    f = b.fids()
    got = np.stack([qid, f[off[qid] + within].astype(np.int64)], axis=1)
    return got[np.lexsort((got[:, 1], got[:, 0]))]


def _want_pairs(oix, regions, mode, invert):
    t, c = oix.query_features(regions, int(mode), invert)
    wc = c.astype(np.int64)
    by_chr = np.argsort(regions[:, 0], kind="stable")  # the oracle walks seqid after seqid, regions in input order
    want = np.stack([np.repeat(by_chr, wc[by_chr]), t[:, 0].astype(np.int64)], axis=1)
    return want[np.lexsort((want[:, 1], want[:, 0]))], c, t


def _make_batches(ix, region_sets):
    out = []
    for r in region_sets:
        b = engine.QueryBatch(ix, max(len(r), 1))
        b.set_regions(r)
        out.append(b)
    return out


SIZES = [[70_000, 70_000], [1, 50_000, 4096], [30_000, 8191, 8193, 100, 60_000], [20_000] * 8, [5000, 300_000, 777, 2048, 12_289, 1, 65_536]]


@pytest.mark.parametrize("sizes", SIZES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("group", [1, 2])
def test_grouped_pair_passes_equal_the_oracle(sizes, group):
    roots = _roots()
    oix, ix = _oracle(roots), _index(roots)
    sets = [synth.synth_bed(n, seed=2000 + 7 * i + n % 13, edge_frac=0.01, roots=roots) for i, n in enumerate(sizes)]
    bs = _make_batches(ix, sets)
    bs[0].set_option("GROUP", group)
    for flags, use_seg in ((engine.OUT_FIDS | engine.OUT_SEGBASE, True), (engine.OUT_FIDS | engine.OUT_OFFSETS, False)):
        # three passes per batch, handed over together (the last pass of every batch is what is checked; the earlier ones ran
        # on the same buffers: cursors, tickets and strips alternate correctly or the last one is wrong)
        engine.run_batches(bs, OverlapMode.Overlap, False, flags, engine.STRATEGY_AUTO, 3 * len(bs))
        for b, r in zip(bs, sets):
            b.wait()
            want, wc, _ = _want_pairs(oix, r, OverlapMode.Overlap, False)
            c = b.counts()
            assert np.array_equal(c, wc)
            assert b.total_hits == len(want)
            got = _pairs_of(b, r, c, b.offsets_from_segbase(c) if use_seg else None)
            assert np.array_equal(got, want)
    for b in bs:
        b.close()
    ix.close()


@pytest.mark.parametrize("mode", list(OverlapMode))
@pytest.mark.parametrize("invert", [False, True])
def test_grouped_modes_triples_and_roots(mode, invert):
    roots = _roots()
    oix, ix = _oracle(roots), _index(roots)
    sets = [synth.synth_bed(n, seed=3100 + i, edge_frac=0.02, roots=roots, width=(10, 30_000 if i % 2 else 8000)) for i, n in enumerate([40_000, 9000, 25_000])]
    bs = _make_batches(ix, sets)
    # triples + per-region offsets + the unique roots (a pair pass and a root pass behind it: two grouped launches)
    flags = engine.OUT_TRIPLES | engine.OUT_FIDS | engine.OUT_OFFSETS | engine.OUT_ROOT_BITMAP
    engine.run_batches(bs, mode, invert, flags, engine.STRATEGY_AUTO, 2 * len(bs))
    for b, r in zip(bs, sets):
        b.wait()
        want, wc, wt = _want_pairs(oix, r, mode, invert)
        assert np.array_equal(b.counts(), wc)
        t = b.triples()
        order = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]  # noqa: E731
        assert np.array_equal(order(t), order(wt))
        assert np.array_equal(_pairs_of(b, r, wc), want)
        assert np.array_equal(b.unique_roots(), np.unique(wt[:, 0]))
    # root passes of their own (what the CLI runs), accumulating over two grouped passes with different regions
    sets2 = [synth.synth_bed(len(r), seed=3200 + i, edge_frac=0.02, roots=roots) for i, r in enumerate(sets)]
    engine.run_batches(bs, mode, invert, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, engine.STRATEGY_AUTO)
    for b, r2 in zip(bs, sets2):
        b.set_regions(r2)  # (the batch's own stream joins the group's launch before the copy overwrites the regions)
    engine.run_batches(bs, mode, invert, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS | engine.OUT_BITMAP_KEEP, engine.STRATEGY_AUTO)
    for b, r, r2 in zip(bs, sets, sets2):
        b.wait()
        t1, _ = oix.query_features(r, int(mode), invert)
        t2, _ = oix.query_features(r2, int(mode), invert)
        assert np.array_equal(b.unique_roots(), np.unique(np.concatenate([t1[:, 0], t2[:, 0]])))
        assert b.kept_pairs_accumulated == len(t1) + len(t2)
    for b in bs:
        b.close()
    ix.close()


def test_grouped_and_single_runs_interleave_on_one_batch():
    """a batch runs alone, then in a group, then alone again with new regions, then in a group on the OTHER group stream: every
    result is the oracle's (the streams are joined wherever the batch changes hands)"""
    roots = _roots()
    oix, ix = _oracle(roots), _index(roots)
    sets = [synth.synth_bed(50_000, seed=4000 + i, edge_frac=0.01, roots=roots) for i in range(5)]
    bs = _make_batches(ix, sets)
    flags = engine.OUT_FIDS | engine.OUT_OFFSETS

    def check(b, r):
        b.wait()
        want, wc, _ = _want_pairs(oix, r, OverlapMode.Overlap, False)
        assert np.array_equal(b.counts(), wc)
        assert np.array_equal(_pairs_of(b, r, wc), want)

    bs[0].run(OverlapMode.Overlap, False, flags)
    engine.run_batches(bs[:3], OverlapMode.Overlap, False, flags)  # group stream 0 behind batch 0's own pass
    bs[1].run(OverlapMode.Overlap, False, flags)                   # own stream behind the group
    engine.run_batches(bs, OverlapMode.Overlap, False, flags)      # five batches: halves of 3 + 2 on the two group streams
    engine.run_batches(bs[1:], OverlapMode.Overlap, False, flags, n_passes=9)  # four batches, nine passes: members change streams
    for b, r in zip(bs, sets):
        check(b, r)
    new = synth.synth_bed(33_333, seed=4100, edge_frac=0.01, roots=roots)
    bs[2].set_regions(new)
    sets[2] = new
    engine.run_batches(bs[1:4], OverlapMode.Overlap, False, flags)
    for b, r in zip(bs, sets):
        check(b, r)
    # an empty batch and an Overlap + invert pass (which launches nothing) in a group
    bs[3].set_regions(np.zeros((0, 3), np.uint32))
    engine.run_batches(bs[2:], OverlapMode.Overlap, False, flags)
    bs[3].wait()
    assert bs[3].total_hits == 0
    check(bs[2], sets[2])
    check(bs[4], sets[4])
    engine.run_batches([bs[0], bs[1]], OverlapMode.Overlap, True, flags)
    for b in bs[:2]:
        b.wait()
        assert b.total_hits == 0 and not b.counts().any()
    for b in bs:
        b.close()
    ix.close()


def test_grouped_mixed_form_and_forced_form():
    """batches with SV-sized rows take the mixed form in a group as well; a batch whose form differs runs on its own"""
    roots = _roots()
    oix, ix = _oracle(roots), _index(roots)
    rng = np.random.default_rng(5)
    sets = []
    for i in range(4):
        r = synth.synth_bed(30_000, seed=5000 + i, edge_frac=0.01, roots=roots)
        if i != 3:  # every tenth row SV-sized; batch 3 stays plain (AUTO: the narrow form -> not groupable with the others)
            w = rng.integers(20_000, 2_000_000, len(r) // 10).astype(np.uint32)
            r[::10, 2][: len(w)] = r[::10, 1][: len(w)] + w
        sets.append(np.ascontiguousarray(r))
    bs = _make_batches(ix, sets)
    for mode in OverlapMode:
        flags = engine.OUT_FIDS | engine.OUT_OFFSETS
        engine.run_batches(bs[:3], mode, False, flags, engine.STRATEGY_AUTO, 6)  # one launch of the mixed form for the three
        engine.run_batches(bs, mode, False, flags, engine.STRATEGY_AUTO, 2 * len(bs))  # (pass by pass: the forms differ)
        for i, (b, r) in enumerate(zip(bs, sets)):
            b.wait()
            assert b.wide_form == (i != 3)
            want, wc, _ = _want_pairs(oix, r, mode, False)
            assert np.array_equal(b.counts(), wc)
            assert np.array_equal(_pairs_of(b, r, wc), want)
    for b in bs:
        b.close()
    ix.close()


def test_grouped_capacity_replay_and_timed_runs():
    """a batch whose pair buffer is too small replays at its wait, also after a grouped launch; the timed group runs leave every
    batch's results intact"""
    roots = _roots()
    oix, ix = _oracle(roots), _index(roots)
    sets = [synth.synth_bed(60_000, seed=6000 + i, edge_frac=0.3, roots=roots, width=(100, 15_000)) for i in range(3)]
    bs = _make_batches(ix, sets)
    for b in bs:
        b.reserve_hits(1024)  # far too few
    flags = engine.OUT_FIDS | engine.OUT_SEGBASE
    engine.run_batches(bs, OverlapMode.Overlap, False, flags)
    for b, r in zip(bs, sets):
        b.wait()
        want, wc, _ = _want_pairs(oix, r, OverlapMode.Overlap, False)
        c = b.counts()
        assert np.array_equal(c, wc)
        assert np.array_equal(_pairs_of(b, r, c, b.offsets_from_segbase(c)), want)
    us, grouped = engine.timed_group_runs(bs, OverlapMode.Overlap, False, flags, engine.STRATEGY_AUTO, 5)
    assert grouped and us > 0
    for b, r in zip(bs, sets):
        b.wait()
        want, wc, _ = _want_pairs(oix, r, OverlapMode.Overlap, False)
        c = b.counts()
        assert np.array_equal(c, wc)
        assert np.array_equal(_pairs_of(b, r, c, b.offsets_from_segbase(c)), want)
    for b in bs:
        b.close()
    ix.close()


@pytest.mark.parametrize("tickets", [1, 2, 3, 4])
@pytest.mark.parametrize("threads,blocks", [(512, 3), (1024, 5), (512, 16)])
def test_rounds_by_ticket_on_small_grids(tickets, threads, blocks):
    """the launch's tail (or every round) handed out by ticket, forced on grids of a few blocks so that every block takes many rounds
    by ticket; ticket-kind launches alternate with plain ones (a root pass takes no tickets) on ONE batch: the ticket words must be
    zero whenever a launch of the ticket kind starts (round 6: they alternate per KIND of launch, not per pass)"""
    roots = _roots()
    oix, ix = _oracle(roots), _index(roots)
    regions = synth.synth_bed(150_000, seed=7000 + tickets, edge_frac=0.02, roots=roots)
    srt = np.ascontiguousarray(regions[np.lexsort((regions[:, 1], regions[:, 0]))])
    for regs in (regions, srt):
        b = engine.QueryBatch(ix, len(regs))
        b.set_option("TICKETS", tickets)
        b.set_option("WIN_THREADS", threads)
        b.set_option("FUSED_BLOCKS", blocks)
        b.set_regions(regs)
        want, wc, wt = _want_pairs(oix, regs, OverlapMode.Overlap, False)
        for rep in range(3):
            b.run(OverlapMode.Overlap, False, engine.OUT_FIDS | engine.OUT_SEGBASE, engine.STRATEGY_WINDOWS)
            b.wait()
            c = b.counts()
            assert np.array_equal(c, wc) and b.total_hits == len(want), (rep, b.total_hits, len(want))
            assert np.array_equal(_pairs_of(b, regs, c, b.offsets_from_segbase(c)), want)
            b.run(OverlapMode.Overlap, False, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, engine.STRATEGY_WINDOWS)  # (a plain launch in between)
            b.wait()
            assert np.array_equal(b.unique_roots(), np.unique(wt[:, 0]))
            b.run(OverlapMode.Overlap, False, engine.OUT_TRIPLES | engine.OUT_OFFSETS | engine.OUT_ROOT_BITMAP, engine.STRATEGY_WINDOWS)
            b.wait()
            assert np.array_equal(b.counts(), wc) and b.total_hits == len(want)
        b.close()
    ix.close()
