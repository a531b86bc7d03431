"""The wide / mixed form of k_join_pairs / k_join_roots (regions of any width, every mode, inverted or not) == oracle, bit for bit.

A region [qs, qe) overlaps the roots over its first base and the roots that start inside it (gffx_device.hpp, "ranks"):
the kernel reads the line of qs, the line of qe - 1 and a rank word for each.  Forced on every eligible pass of the windows
strategy with GFFX_HIP_WIN_WIDE=2; AUTO's own choice (after a pass that sent most regions to the sweep) is covered at the end.
Reference semantics: utils/tree.rs:110 (query_interval), commands/intersect.rs:145-161.
"""
import numpy as np
import pytest

from gffx_amd import engine, synth
from gffx_amd.engine import OverlapMode
from oracle import binding as ob

pytestmark = pytest.mark.gpu

OV = int(OverlapMode.Overlap)


def _rows(t):
    t = np.asarray(t, dtype=np.uint32).reshape(-1, 3)
    return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]


def _pairs_of(regions, counts, offsets, words):
    wc = counts.astype(np.int64)
    qid = np.repeat(np.arange(len(regions), dtype=np.int64), wc)
    within = np.arange(len(qid), dtype=np.int64) - np.repeat(np.cumsum(wc) - wc, wc)
    p = np.stack([qid, words[offsets.astype(np.int64)[qid] + within].astype(np.int64)], axis=1)
    return p[np.lexsort((p[:, 1], p[:, 0]))]


def _want_pairs(regions, want_t, want_c):
    wc = want_c.astype(np.int64)
    by_chr = np.argsort(regions[:, 0], kind="stable")  # the oracle walks seqid after seqid, regions in input order
    p = np.stack([np.repeat(by_chr, wc[by_chr]), want_t[:, 0].astype(np.int64)], axis=1)
    return p[np.lexsort((p[:, 1], p[:, 0]))]


def _check_wide(roots, regions, strategy=engine.STRATEGY_WINDOWS, soa=False, OV=OV, inv=False):
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    want_t, want_c = oix.query_features(regions, int(OV), inv)
    want_p = _want_pairs(regions, want_t, want_c)
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    b = engine.QueryBatch(ix, max(len(regions), 1))
    if soa:
        b.set_regions_soa(regions[:, 0], regions[:, 1], regions[:, 2])
    else:
        b.set_regions(regions)
    # root_fids + triples + u64 offsets (the position pass and k_expand_pairs)
    b.run(OV, inv, engine.OUT_FIDS | engine.OUT_TRIPLES | engine.OUT_OFFSETS, strategy)
    b.wait()
    assert b.total_hits == len(want_t)
    assert np.array_equal(b.counts(), want_c)
    got_t = b.triples()
    assert np.array_equal(_rows(got_t), _rows(want_t))
    assert np.array_equal(b.fids(), got_t[:, 0])
    assert np.array_equal(_pairs_of(regions, want_c, b.offsets()[:-1], b.fids()), want_p)
    # root_fids + one base per group of 256 regions (the pass bench.py times)
    b.run(OV, inv, engine.OUT_FIDS | engine.OUT_SEGBASE, strategy)
    b.wait()
    c3, f3 = b.counts(), b.fids()
    assert np.array_equal(c3, want_c) and len(f3) == len(want_t)
    assert np.array_equal(_pairs_of(regions, want_c, b.offsets_from_segbase(c3), f3), want_p)
    # u32 offsets
    b.run(OV, inv, engine.OUT_FIDS | engine.OUT_OFFSETS32, strategy)
    b.wait()
    assert np.array_equal(b.counts(), want_c)
    assert np.array_equal(_pairs_of(regions, want_c, b.offsets32(), b.fids()), want_p)
    # counts alone
    b.run(OV, inv, 0, strategy)
    b.wait()
    assert np.array_equal(b.counts(), want_c) and b.total_hits == len(want_t)
    assert b.wide_form
    # the unique roots: a root pass of its own (with and without counts: what the CLI asks for), and behind a pair pass
    want_u = np.unique(want_t[:, 0])
    b.run(OV, inv, engine.OUT_ROOT_BITMAP, strategy)
    b.wait()
    assert b.wide_form and np.array_equal(b.unique_roots(), want_u) and np.array_equal(b.counts(), want_c) and b.total_hits == len(want_t)
    b.run(OV, inv, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, strategy)
    b.wait()
    assert b.wide_form and np.array_equal(b.unique_roots(), want_u) and b.total_hits == len(want_t)
    b.run(OV, inv, engine.OUT_FIDS | engine.OUT_OFFSETS | engine.OUT_ROOT_BITMAP, strategy)
    b.wait()
    assert b.wide_form and np.array_equal(b.unique_roots(), want_u) and np.array_equal(b.counts(), want_c)
    assert np.array_equal(_pairs_of(regions, want_c, b.offsets()[:-1], b.fids()), want_p)
    b.close()
    ix.close()
    return len(want_t)


def _mixed_widths(rng, n, chroms, n_chr_extra=0):
    """regions of every width class: a base, narrower than a line answers, a few windows, megabases, whole seqids, and the edge
    rows the reference keeps as they are (empty, reversed, start 0, beyond the seqid's last root, unknown seqids)"""
    clen = np.array([l for _, l in chroms], dtype=np.int64)
    c = rng.integers(0, len(chroms) + n_chr_extra, n)
    L = clen[np.minimum(c, len(chroms) - 1)]
    cls = rng.integers(0, 8, n)
    w = np.select([cls == 0, cls == 1, cls == 2, cls == 3, cls == 4, cls == 5],
                  [np.ones(n, np.int64), rng.integers(1, 2000, n), rng.integers(2000, 40000, n), rng.integers(40000, 400000, n),
                   rng.integers(400000, 5_000_000, n), L], rng.integers(1, 100000, n))
    s = (rng.random(n) * L).astype(np.int64)
    e = s + w
    k = cls == 6  # empty / reversed
    e[k] = s[k] - rng.integers(0, 3, int(k.sum()))
    k = cls == 7  # from base 0 / far beyond the end of the seqid
    half = rng.random(n) < 0.5
    s[k & half] = 0
    s[k & ~half] = L[k & ~half] + rng.integers(0, 1 << 20, int((k & ~half).sum()))
    e[k & ~half] = s[k & ~half] + w[k & ~half]
    e = np.clip(e, 0, 0xFFFFFFFF)
    s = np.clip(s, 0, 0xFFFFFFFF)
    out = np.empty((n, 3), dtype=np.uint32)
    out[:, 0], out[:, 1], out[:, 2] = c, s, e
    return out


@pytest.fixture
def wide_forced(monkeypatch):
    monkeypatch.setenv("GFFX_HIP_WIN_WIDE", "2")


def test_wide_form_gencode_like(wide_forced):
    roots = synth.gencode_like_roots(63000, seed=42)
    rng = np.random.default_rng(5)
    regions = _mixed_widths(rng, 40_000, synth.GRCH38)
    pairs = _check_wide(roots, regions)
    assert pairs > 1_000_000
    assert _check_wide(roots, regions, OV=OverlapMode.Contained) > 100_000  # (round 5: a wide lane's run filtered by the roots' ends)
    # (round 5, second half: the roots over qs that reach the region's end -- by their true ends where the line's 16 bits do not tell --,
    #  and the inverted passes)
    assert _check_wide(roots, regions, OV=OverlapMode.ContainsRegion) > 1000
    assert _check_wide(roots, regions, OV=OverlapMode.ContainsRegion, inv=True, soa=True) > 1_000_000
    assert _check_wide(roots, regions, OV=OverlapMode.Contained, inv=True) > 10_000
    regions = synth.synth_bed(30_011, seed=77, width=(100, 200000), edge_frac=0.05, roots=roots)  # bench.py's wide_regions shape
    _check_wide(roots, regions, soa=True)
    _check_wide(roots, regions, OV=OverlapMode.Contained)
    _check_wide(roots, regions, OV=OverlapMode.ContainsRegion)
    _check_wide(roots, regions, OV=OverlapMode.ContainsRegion, inv=True)
    _check_wide(roots, regions, OV=OverlapMode.Contained, inv=True, soa=True)


@pytest.mark.parametrize("seed", range(6))
def test_wide_form_small_indexes(wide_forced, seed):
    """few roots per seqid, seqids without roots, dense stacks (lines whose lists continue elsewhere -> the sweep), split windows"""
    rng = np.random.default_rng(100 + seed)
    chroms = [("a", 3_000_000), ("b", 50_000), ("c", 800_000), ("d", 1000), ("e", 2_000_000)]
    starts, ends, offs = [], [], [0]
    for ci, (_, clen) in enumerate(chroms):
        k = [int(rng.integers(200, 3000)), 0, int(rng.integers(1, 40)), 1, int(rng.integers(500, 1500))][ci] if seed % 2 == 0 else \
            int(rng.integers(0, 1500))
        st = np.sort(rng.integers(0, max(1, clen - 10), k))
        ln = np.exp(rng.normal(np.log(3000.0), 1.8, size=k)).astype(np.int64) + 1
        if ci == 4 and k:  # a stack of near-identical roots: long lists
            st[: k // 3] = st[k // 3] + rng.integers(0, 50, k // 3)
            st = np.sort(st)
        if seed == 3 and k:
            ln[rng.random(k) < 0.05] = 0  # empty intervals (end == start): kept when qs < start < qe
        starts.append(st)
        ends.append(np.minimum(st + ln, 0xFFFFFFF0))
        offs.append(offs[-1] + k)
    start = np.concatenate(starts).astype(np.uint32)
    end = np.concatenate(ends).astype(np.uint32)
    roots = {"chr_offsets": np.array(offs, dtype=np.uint32), "start": start, "end": end,
             "fid": rng.permutation(len(start)).astype(np.uint32) * 3 + 1}
    regions = _mixed_widths(rng, 6000 + 257 * seed, chroms, n_chr_extra=0)
    _check_wide(roots, regions, soa=bool(seed & 1))
    _check_wide(roots, regions, soa=not (seed & 1), OV=OverlapMode.Contained)
    _check_wide(roots, regions, soa=bool(seed & 2), OV=OverlapMode.ContainsRegion, inv=bool(seed & 1))
    _check_wide(roots, regions, soa=not (seed & 2), OV=OverlapMode.Contained if seed & 1 else OverlapMode.ContainsRegion, inv=True)


def test_wide_form_is_autos_choice_for_wide_batches(monkeypatch):
    """AUTO without the prior from the host's rows (as for regions that are already on the device): the first pass over wide
    regions runs the narrow form (most regions take the sweep), the batch's later overlap-mode passes the wide form -- and a pass of
    another mode leaves it again.  With the prior the first pass is wide already."""
    monkeypatch.setenv("GFFX_HIP_WIDTH_SAMPLE", "0")
    roots = synth.gencode_like_roots(20000, seed=3)
    regions = synth.synth_bed(50_000, seed=9, width=(30000, 300000))
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    want_t, want_c = oix.query_features(regions, OV, False)
    want_p = _want_pairs(regions, want_t, want_c)
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    b = engine.QueryBatch(ix, len(regions))
    b.set_regions(regions)
    for i in range(3):
        b.run(OV, False, engine.OUT_FIDS | engine.OUT_OFFSETS, engine.STRATEGY_AUTO)
        b.wait()
        assert np.array_equal(b.counts(), want_c)
        assert np.array_equal(_pairs_of(regions, want_c, b.offsets()[:-1], b.fids()), want_p)
        assert b.wide_form == (i > 0)
    for mode, inv in ((OverlapMode.Contained, False), (OverlapMode.ContainsRegion, False), (OverlapMode.Contained, True)):
        wt, wc = oix.query_features(regions, int(mode), inv)
        b.run(mode, inv, engine.OUT_FIDS | engine.OUT_OFFSETS, engine.STRATEGY_AUTO)
        b.wait()
        # (every mode takes the mixed form since round 5, inverted or not)
        assert b.wide_form and np.array_equal(b.counts(), wc)
        assert np.array_equal(_pairs_of(regions, wc, b.offsets()[:-1], b.fids()), _want_pairs(regions, wt, wc))
    b.run(OV, False, engine.OUT_FIDS | engine.OUT_ROOT_BITMAP, engine.STRATEGY_AUTO)
    b.wait()
    assert b.wide_form and np.array_equal(b.unique_roots(), np.unique(want_t[:, 0]))
    b.run(OV, False, engine.OUT_FIDS | engine.OUT_SEGBASE, engine.STRATEGY_AUTO)
    b.wait()
    assert b.wide_form and np.array_equal(b.counts(), want_c)
    b.set_option("WIDTH_SAMPLE", 1)
    b.set_regions(regions)  # the host's rows again, sampled this time
    b.run(OV, False, engine.OUT_FIDS | engine.OUT_OFFSETS, engine.STRATEGY_AUTO)
    b.wait()
    assert b.wide_form and np.array_equal(b.counts(), want_c)
    assert np.array_equal(_pairs_of(regions, want_c, b.offsets()[:-1], b.fids()), want_p)
    b.close()
    ix.close()


@pytest.mark.parametrize("frac,mixed", [(0.10, True), (0.02, True), (0.002, False), (0.0, False)])
def test_mixed_batches_take_the_mixed_form(frac, mixed, monkeypatch):
    """A BED file with SOME SV-sized rows (round 5): more than one row in 128 wider than their seqid's lines answer and
    AUTO's overlap-mode passes -- pair passes, triples, the CLI's root pass -- run the MIXED form of the window kernels, in which every
    lane serves its region the narrow way (one line) or the wide way (two lines, two ranks); fewer, and the narrow form keeps the batch
    (its few wide rows sweep).  Either way every region's pairs equal the oracle's.  Host rows are judged from a sample, device-resident
    ones from the first waited pass; the other modes are not touched by any of this."""
    roots = synth.gencode_like_roots(30000, seed=4)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    regions = synth.synth_bed(300_000, seed=12, edge_frac=0.01, roots=roots)
    rng = np.random.default_rng(13)
    pick = rng.choice(len(regions), int(frac * len(regions)), replace=False)
    regions[pick, 2] = np.minimum(regions[pick, 1].astype(np.int64) + rng.integers(20_000, 2_000_000, len(pick)), 0xFFFFFFF0).astype(np.uint32)
    want_t, want_c = oix.query_features(regions, OV, False)
    want_p = _want_pairs(regions, want_t, want_c)
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    b = engine.QueryBatch(ix, len(regions))
    b.set_regions(regions)  # (host rows: sampled)
    b.run(OV, False, engine.OUT_FIDS | engine.OUT_OFFSETS)
    b.wait()
    assert b.wide_form == mixed and np.array_equal(b.counts(), want_c)
    assert np.array_equal(_pairs_of(regions, want_c, b.offsets()[:-1], b.fids()), want_p)
    b.run(OV, False, engine.OUT_TRIPLES | engine.OUT_OFFSETS)
    b.wait()
    assert b.wide_form == mixed and np.array_equal(_rows(b.triples()), _rows(want_t))
    b.run(OV, False, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS)
    b.wait()
    assert b.wide_form == mixed and np.array_equal(b.unique_roots(), np.unique(want_t[:, 0])) and b.total_hits == len(want_t)
    wt, wc = oix.query_features(regions, int(OverlapMode.Contained), False)
    for flags in (engine.OUT_FIDS | engine.OUT_OFFSETS, engine.OUT_TRIPLES | engine.OUT_OFFSETS32):
        b.run(OverlapMode.Contained, False, flags)
        b.wait()
        assert b.wide_form == mixed and np.array_equal(b.counts(), wc) and b.total_hits == len(wt)
    assert np.array_equal(_rows(b.triples()), _rows(wt))
    b.run(OverlapMode.Contained, False, engine.OUT_ROOT_BITMAP)
    b.wait()
    assert b.wide_form == mixed and np.array_equal(b.unique_roots(), np.unique(wt[:, 0])) and np.array_equal(b.counts(), wc)
    for mode, inv in ((OverlapMode.ContainsRegion, False), (OverlapMode.Contained, True), (OverlapMode.ContainsRegion, True)):
        wt, wc = oix.query_features(regions, int(mode), inv)
        b.run(mode, inv, engine.OUT_FIDS | engine.OUT_OFFSETS)
        b.wait()
        assert b.wide_form == mixed and np.array_equal(b.counts(), wc)
        assert np.array_equal(_pairs_of(regions, wc, b.offsets()[:-1], b.fids()), _want_pairs(regions, wt, wc))
        b.run(mode, inv, engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS)
        b.wait()
        assert b.wide_form == mixed and np.array_equal(b.unique_roots(), np.unique(wt[:, 0])) and b.total_hits == len(wt)
    # the same regions resident on the device: no sample, the first waited pass (narrow form) counts the rows its lines did not answer
    b.set_option("WIDTH_SAMPLE", 0)
    b.set_regions(regions)
    for i in range(2):
        b.run(OV, False, engine.OUT_FIDS | engine.OUT_SEGBASE)
        b.wait()
        assert b.wide_form == (mixed and i > 0) and np.array_equal(b.counts(), want_c)
        assert np.array_equal(_pairs_of(regions, want_c, b.offsets_from_segbase(b.counts()), b.fids()), want_p)
    b.close()
    ix.close()


def test_wide_form_full_size_properties(wide_forced, monkeypatch):
    """1 M regions of bench.py's wide shape.  EVERY count against first principles: a root with end >= start that ends at or
    before qs also starts before qe, so the roots a region [qs, qe) overlaps are #{start < qe} - #{end <= qs} -- two binary
    searches per region over the seqid's sorted starts and sorted ends.  The counts add up to the pairs; a 20 k-region sample
    agrees with the oracle pair by pair; the sweep kernel's pass over the same batch keeps the same multiset of root_fids."""
    roots = synth.gencode_like_roots(63000, seed=42)
    co, rs, re = roots["chr_offsets"], roots["start"].astype(np.int64), roots["end"].astype(np.int64)
    regions = synth.synth_bed(1_000_000, seed=1004, width=(100, 200000))
    ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
    b = engine.QueryBatch(ix, len(regions))
    b.set_regions(regions)
    b.run(OV, False, engine.OUT_FIDS | engine.OUT_OFFSETS, engine.STRATEGY_WINDOWS)
    b.wait()
    assert b.wide_form
    c, off, f = b.counts(), b.offsets(), b.fids()
    assert int(c.sum()) == b.total_hits == len(f)
    want = np.zeros(len(regions), dtype=np.int64)
    for k in range(len(co) - 1):
        sel = np.nonzero(regions[:, 0] == k)[0]
        st, en = rs[co[k]:co[k + 1]], np.sort(re[co[k]:co[k + 1]])  # (starts are sorted per seqid)
        want[sel] = np.searchsorted(st, regions[sel, 2].astype(np.int64), "left") - np.searchsorted(en, regions[sel, 1].astype(np.int64), "right")
    assert np.array_equal(c.astype(np.int64), want)
    oix = ob.OracleIndex.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
    pick = np.random.default_rng(1).choice(len(regions), 20000, replace=False)
    wt, wc = oix.query_features(regions[pick], OV, False)
    assert np.array_equal(c[pick], wc)
    assert np.array_equal(_pairs_of(regions[pick], wc, off[:-1][pick], f), _want_pairs(regions[pick], wt, wc))
    # the other modes at full size, through properties that need no oracle (intersect.rs:145-161): a mode's pass and its inverted
    # pass split the overlap pass's pairs between them, region by region -- counts, and the sorted root_fids of the sample's regions;
    # the sample itself against the oracle
    fs = np.sort(f)
    for mode in (OverlapMode.Contained, OverlapMode.ContainsRegion):
        got = []
        for inv in (False, True):
            b.run(mode, inv, engine.OUT_FIDS | engine.OUT_OFFSETS, engine.STRATEGY_WINDOWS)
            b.wait()
            assert b.wide_form
            cm, om, fm = b.counts(), b.offsets(), b.fids()
            assert int(cm.sum()) == b.total_hits == len(fm)
            wt, wc = oix.query_features(regions[pick], int(mode), inv)
            assert np.array_equal(cm[pick], wc)
            assert np.array_equal(_pairs_of(regions[pick], wc, om[:-1][pick], fm), _want_pairs(regions[pick], wt, wc))
            got.append((cm, fm))
        assert np.array_equal(got[0][0] + got[1][0], c)
        assert np.array_equal(np.sort(np.concatenate([got[0][1], got[1][1]])), fs)
    b.set_option("WIN_WIDE", 0)  # (a batch reads its knobs from the environment once, when it is created; later: set_option)
    b.run(OV, False, engine.OUT_FIDS | engine.OUT_OFFSETS, engine.STRATEGY_FUSED)
    b.wait()
    assert not b.wide_form and np.array_equal(b.counts(), c) and np.array_equal(fs, np.sort(b.fids()))
    b.close()
    ix.close()


def test_wide_form_on_a_cloned_index(wide_forced):
    """gffx_hip_index_clone copies the wide form's line table and the root_fids by position with everything else."""
    roots = synth.gencode_like_roots(20000, seed=11)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    regions = _mixed_widths(np.random.default_rng(12), 20_000, synth.GRCH38)
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    want_t, want_c = oix.query_features(regions, OV, False)
    ix0 = engine.TreeIndexData.from_roots(co, s, e, f)
    ix = ix0.clone(0)
    ix0.close()  # (the clone owns its device arrays)
    b = engine.QueryBatch(ix, len(regions))
    b.set_regions(regions)
    b.run(OV, False, engine.OUT_FIDS | engine.OUT_OFFSETS | engine.OUT_ROOT_BITMAP, engine.STRATEGY_WINDOWS)
    b.wait()
    assert b.wide_form and np.array_equal(b.counts(), want_c)
    assert np.array_equal(_pairs_of(regions, want_c, b.offsets()[:-1], b.fids()), _want_pairs(regions, want_t, want_c))
    assert np.array_equal(b.unique_roots(), np.unique(want_t[:, 0]))
    b.close()
    ix.close()


def test_root_passes_of_both_forms_in_turn_on_an_index_with_many_windows(monkeypatch):
    """Found by tools/fuzz_parity.py: 300 seqids of 250 Mbp with a few dozen roots each have 600 k windows -- a 76 KB split
    bitmap.  The narrow root pass sheds it (the coverage filter is there too), the wide one kept it and had no room left for the
    block's root bitmap: its bits went straight to the batch's bitmap, and the fold of the slabs of the EARLIER narrow pass then
    wiped them.  The root bitmap now comes first in a block's LDS, and the fold ORs."""
    rng = np.random.default_rng(1552)
    n_chr, span = 300, 250_000_000
    per = np.minimum(rng.choice([0, 1, 2, 10, 200, 3000], size=n_chr, p=[.1, .1, .1, .3, .3, .1]), 50)
    co = np.concatenate([[0], np.cumsum(per)]).astype(np.uint32)
    R = int(co[-1])
    s = np.concatenate([np.sort(rng.integers(0, span, k)) for k in per]).astype(np.int64)
    ln = np.where(rng.random(R) < 0.1, rng.integers(1, span // 2, R), rng.integers(1, span // 500, R))
    e = np.minimum(s + ln, 0xFFFFFFFF)
    roots = {"chr_offsets": co, "start": s.astype(np.uint32), "end": e.astype(np.uint32), "fid": (np.arange(R, dtype=np.uint32) * 3 + 1)}
    nq = 63
    qc = rng.integers(0, n_chr, nq)
    qs = rng.integers(0, span, nq)
    qe = np.minimum(qs + np.where(rng.random(nq) < 0.5, rng.integers(1, 80_000, nq), rng.integers(1, span, nq)), 0xFFFFFFFF)
    regions = np.stack([qc, qs, qe], axis=1).astype(np.uint32)
    oix = ob.OracleIndex.from_roots(co, roots["start"], roots["end"], roots["fid"])
    want_t, want_c = oix.query_features(regions, OV, False)
    want_u = np.unique(want_t[:, 0])
    assert len(want_u) > 20
    ix = engine.TreeIndexData.from_roots(co, roots["start"], roots["end"], roots["fid"])
    b = engine.QueryBatch(ix, nq)
    b.set_regions(regions)
    for wide in ("1", "2", "1", "2", "2"):
        b.set_option("GFFX_HIP_WIN_WIDE", int(wide))
        b.run(OV, False, engine.OUT_ROOT_BITMAP, engine.STRATEGY_WINDOWS)
        b.wait()
        assert b.wide_form == (wide == "2")
        assert np.array_equal(b.unique_roots(), want_u) and np.array_equal(b.counts(), want_c), wide
    # ... and accumulated over two halves of the regions (GFFX_OUT_BITMAP_KEEP), one half per form
    half = nq // 2
    b.set_regions(regions[:half])
    b.set_option("win_wide", 1)
    b.run(OV, False, engine.OUT_ROOT_BITMAP, engine.STRATEGY_WINDOWS)
    b.set_regions(regions[half:])
    b.set_option("win_wide", 2)
    b.run(OV, False, engine.OUT_ROOT_BITMAP | engine.OUT_BITMAP_KEEP, engine.STRATEGY_WINDOWS)
    b.wait()
    assert b.wide_form and np.array_equal(b.unique_roots(), want_u)
    b.close()
    ix.close()
