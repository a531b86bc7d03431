"""Development tool (GPU box): randomized differential tests of the other device entry points against their
definitions -- Join B (k_lines_exists vs the oracle's literal scan), covered bases (k_segments_covered vs a per-base
numpy evaluation) and depth (k_depth_regions vs the numpy definition of tests/test_depth_gpu.py).
python tools/fuzz_lines.py [iterations] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gffx_amd import engine
from tests.test_join_b_gpu import _oracle_keep
from tests.test_coverage_gpu import _numpy_covered
from tests.test_depth_gpu import _numpy_depth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
for it in range(iters):
    # ---- Join B
    n_seq = int(rng.choice([1, 2, 5, 30]))
    n_lines = int(rng.choice([1, 63, 64, 65, 1000, 2500]))
    nq = int(rng.choice([0, 1, 2, 17, 300, 5000]))
    span = int(rng.choice([12, 300, 100_000, 0xFFFFFFF0]))
    seq = rng.integers(0, n_seq + 1, n_lines).astype(np.uint32)
    seq[rng.random(n_lines) < 0.03] = engine.LineTable.NO_SEQ
    s = rng.integers(0, span, n_lines, dtype=np.int64)
    e = np.where(rng.random(n_lines) < 0.9, s + rng.integers(0, max(2, span // 50), n_lines), rng.integers(0, span, n_lines))
    e = np.clip(e, 0, 0xFFFFFFFF)
    qs = rng.integers(0, span, nq, dtype=np.int64)
    qe = np.where(rng.random(nq) < 0.85, qs + rng.integers(0, max(2, span // 20), nq), rng.integers(0, span, nq))
    regions = np.stack([rng.integers(0, max(1, n_seq - (n_seq > 1)), nq), qs, np.clip(qe, 0, 0xFFFFFFFF)], axis=1).astype(np.uint32)
    lt = engine.LineTable(seq, s.astype(np.uint32), e.astype(np.uint32))
    for mode in (0, 1, 2):
        got = lt.test(regions, n_seq, mode)
        want = _oracle_keep(seq, s.astype(np.uint32), e.astype(np.uint32), regions, n_seq, mode) if nq else np.zeros(n_lines, bool)
        if not np.array_equal(got, want):
            print("JOIN B MISMATCH iteration", it, "mode", mode, n_seq, n_lines, nq, span)
            sys.exit(1)
    # ---- covered bases
    n_seq, span = int(rng.choice([1, 3, 8])), int(rng.choice([50, 5000, 300_000]))
    nq, nseg = int(rng.choice([0, 1, 40, 3000])), int(rng.choice([1, 100, 4000]))
    r = np.empty((nq, 3), np.uint32)
    r[:, 0] = rng.integers(0, n_seq, nq)
    r[:, 1] = rng.integers(0, span, nq)
    r[:, 2] = r[:, 1] + rng.integers(1, max(2, span // rng.choice([2, 50, 1000])), nq)
    sq = rng.integers(0, n_seq, nseg).astype(np.uint32)
    ss = rng.integers(0, span + span // 4 + 1, nseg).astype(np.uint32)
    se = (ss + rng.integers(0, max(2, span // 3), nseg)).astype(np.uint32)
    got = engine.segments_covered(sq, ss, se, r, n_seq)
    if not np.array_equal(got, _numpy_covered(sq, ss, se, r, n_seq)):
        print("COVERED MISMATCH iteration", it, n_seq, span, nq, nseg)
        sys.exit(1)
    # ---- depth
    n_roots = int(rng.choice([1, 5, 60, 400]))
    span = int(rng.choice([2000, 200_000]))
    co = np.array([0, n_roots // 2, n_roots], np.uint32)
    S = np.concatenate([np.sort(rng.integers(0, span, int(co[1]))), np.sort(rng.integers(0, span, n_roots - int(co[1])))]).astype(np.uint32)
    E = (S + rng.integers(1, max(2, span // 10), n_roots)).astype(np.uint32)
    F = (rng.permutation(n_roots) * 2).astype(np.uint32)
    if n_roots > 3:
        F[1] = F[0]  # two tree intervals with one fid
    roots = {"chr_offsets": co, "start": S, "end": E, "fid": F}
    n_fid = int(F.max()) + 2
    block_of_fid = np.full(n_fid, 0xFFFFFFFF, np.uint32)
    owners = {int(F[i]): i for i in range(n_roots)}
    block_off, ls, le, lg, g = [0], [], [], [], 0
    for b, f in enumerate(sorted(owners)):
        if rng.random() < 0.05:
            continue
        block_of_fid[f] = len(block_off) - 1
        i = owners[f]
        for _ in range(int(rng.choice([0, 1, 3, 70]))):
            for _ in range(int(rng.choice([1, 1, 2, 6]))):
                a = int(rng.integers(max(0, int(S[i]) - 50), int(E[i]) + 50))
                ls.append(a)
                le.append(a + int(rng.integers(1, max(2, (int(E[i]) - int(S[i])) // 2 + 2))))
                lg.append(g)
            g += 1
        block_off.append(len(ls))
    ls, le, lg = np.array(ls, np.uint32), np.array(le, np.uint32), np.array(lg, np.uint32)
    block_off = np.array(block_off, np.uint64)
    nq = int(rng.choice([1, 64, 65, 900]))
    reg = np.stack([rng.integers(0, 2, nq), rng.integers(0, span, nq), np.zeros(nq, np.int64)], axis=1)
    reg[:, 2] = reg[:, 1] + rng.integers(1, max(2, span // rng.choice([3, 100])), nq)
    reg = reg.astype(np.uint32)
    ix = engine.TreeIndexData.from_roots(co, S, E, F)
    table = engine.DepthTable(g, block_off, ls, le, lg, block_of_fid)
    bt = engine.QueryBatch(ix, nq)
    bt.set_regions(reg)
    bt.run(2, False, engine.OUT_FIDS | engine.OUT_OFFSETS, int(rng.choice([0, 1, 2, 3, 5])))
    bt.wait()
    table.accumulate(bt)
    want = _numpy_depth(roots, block_of_fid, block_off, ls, le, lg, g, reg)
    for a, w in zip(table.results(), want):
        if not np.array_equal(a, w):
            print("DEPTH MISMATCH iteration", it, n_roots, span, nq, g)
            sys.exit(1)
    bt.close()
    ix.close()
print("fuzz ok: %d iterations (Join B x 3 modes, covered bases, depth), %.0f s" % (iters, time.time() - t0))
