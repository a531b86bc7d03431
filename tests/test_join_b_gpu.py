"""Join B parity on the GPU: the sorted/prefix-max rewrite in k_lines_exists == the oracle's
literal linear scan of commands/intersect.rs:500-521, for every mode, incl. degenerate rows."""
import numpy as np
import pytest

from gffx_amd import engine, synth
from gffx_amd.engine import OverlapMode
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def _oracle_keep(seq, s, e, regions, n_seq, mode):
    order = np.argsort(regions[:, 0], kind="stable")
    r = regions[order]
    off = np.concatenate([[0], np.cumsum(np.bincount(r[:, 0], minlength=n_seq))])
    out = np.zeros(len(seq), dtype=bool)
    for i in range(len(seq)):
        c = int(seq[i])
        if c >= n_seq or off[c + 1] == off[c]:
            continue
        out[i] = ob.line_predicate(int(s[i]), int(e[i]), r[off[c]:off[c + 1], 1], r[off[c]:off[c + 1], 2], int(mode))
    return out


@pytest.mark.parametrize("seed", range(4))
def test_random_small_coordinates_all_modes(seed):
    rng = np.random.default_rng(seed)
    n_seq, n_lines, nq = 4, 3000, int(rng.integers(1, 60))
    span = int(rng.choice([12, 60, 1000]))
    seq = rng.integers(0, n_seq + 1, n_lines).astype(np.uint32)  # n_seq == "unknown seqid" rows
    seq[rng.random(n_lines) < 0.05] = engine.LineTable.NO_SEQ
    s = rng.integers(0, span, n_lines).astype(np.uint32)
    e = rng.integers(0, span, n_lines).astype(np.uint32)  # includes s > e lines
    regions = np.stack([rng.integers(0, n_seq - 1, nq), rng.integers(0, span, nq), rng.integers(0, span, nq)],
                       axis=1).astype(np.uint32)  # seqid n_seq-1 never has a region
    lt = engine.LineTable(seq, s, e)
    for mode in OverlapMode:
        got = lt.test(regions, n_seq, mode)
        assert np.array_equal(got, _oracle_keep(seq, s, e, regions, n_seq, mode)), mode
    # no regions at all -> nothing kept
    assert not lt.test(np.zeros((0, 3), np.uint32), n_seq, OverlapMode.Overlap).any()


def test_u32_extremes_and_empty_table():
    seq = np.array([0, 0, 0, 0], np.uint32)
    s = np.array([0, 0xFFFFFFFF, 5, 0xFFFFFFFE], np.uint32)
    e = np.array([0xFFFFFFFF, 0, 5, 0xFFFFFFFF], np.uint32)
    regions = np.array([[0, 0xFFFFFFFF, 0xFFFFFFFF], [0, 0, 0], [0, 5, 4]], np.uint32)
    lt = engine.LineTable(seq, s, e)
    for mode in OverlapMode:
        assert np.array_equal(lt.test(regions, 1, mode), _oracle_keep(seq, s, e, regions, 1, mode))
    empty = engine.LineTable([], [], [])
    assert empty.test(regions, 1).shape == (0,)
    with pytest.raises(engine._ffi.GffxHipError):
        lt.test(np.array([[3, 1, 2]], np.uint32), 1)


def test_gencode_like_lines_against_sampled_oracle():
    """~200 k lines (genes + children) x 100 k regions: sampled literal-scan parity + the
    property that a line kept in contained mode is kept in overlap mode."""
    roots = synth.gencode_like_roots(20000, seed=42)
    rng = np.random.default_rng(1)
    co = roots["chr_offsets"]
    chr_of = np.repeat(np.arange(len(co) - 1), np.diff(co)).astype(np.uint32)
    rep = 10
    seq = np.repeat(chr_of, rep)
    gs = np.repeat(roots["start"].astype(np.int64) + 1, rep)
    ge = np.repeat(roots["end"].astype(np.int64), rep)
    a = gs + (rng.random(len(gs)) * (ge - gs + 1)).astype(np.int64)
    b = gs + (rng.random(len(gs)) * (ge - gs + 1)).astype(np.int64)
    s, e = np.minimum(a, b).astype(np.uint32), np.maximum(a, b).astype(np.uint32)
    regions = synth.synth_bed(100_000, seed=1001, edge_frac=0.01, roots=roots)
    lt = engine.LineTable(seq, s, e)
    keep = {m: lt.test(regions, 25, m) for m in OverlapMode}
    assert keep[OverlapMode.Overlap].any()
    assert not (keep[OverlapMode.Contained] & ~keep[OverlapMode.Overlap]).any()
    sel = rng.choice(len(seq), 400, replace=False)
    for m in OverlapMode:
        want = _oracle_keep(seq[sel], s[sel], e[sel], regions, 25, m)
        assert np.array_equal(keep[m][sel], want), m
