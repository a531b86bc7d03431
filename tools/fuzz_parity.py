"""Development tool (GPU box): randomized differential test of Join A -- every strategy x mode x invert against the
oracle on random index shapes (sparse / dense / nested / duplicate intervals, tiny and u32-wide coordinates, empty
seqids, many seqids) and random region mixes (narrow, wide, empty-width, reversed, out-of-range positions).
python tools/fuzz_parity.py [iterations] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gffx_amd import engine
from oracle import binding as ob

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
STRATS = [engine.STRATEGY_DIRECT, engine.STRATEGY_SORTED, engine.STRATEGY_FUSED, engine.STRATEGY_WINDOWS]
t0 = time.time()
n_checks = n_wide = n_group = 0
for it in range(iters):
    n_chr = int(rng.choice([1, 2, 3, 7, 40, 300]))
    span = int(rng.choice([200, 5_000, 1_000_000, 250_000_000, 0xFFFFFF00]))
    per = rng.choice([0, 1, 2, 10, 200, 3000], size=n_chr, p=[.1, .1, .1, .3, .3, .1])
    if n_chr > 50:
        per = np.minimum(per, 50)
    co = np.concatenate([[0], np.cumsum(per)]).astype(np.uint32)
    R = int(co[-1])
    shape = rng.choice(["short", "mixed", "nested", "dupes"])
    s = rng.integers(0, span, R, dtype=np.int64)
    if shape == "short":
        ln = rng.integers(1, max(2, span // 2000 + 2), R)
    elif shape == "mixed":
        ln = np.where(rng.random(R) < 0.1, rng.integers(1, span // 2 + 2, R), rng.integers(1, max(2, span // 500 + 2), R))
    elif shape == "nested":
        s = np.sort(s)
        ln = rng.integers(1, span, R) // (1 + np.arange(R) % 7)
        ln = np.maximum(ln, 1)
    else:
        s = rng.choice(s[: max(1, R // 5)] if R else s, R) if R else s
        ln = rng.choice([1, 10, 1000], R)
    e = np.minimum(s + ln, 0xFFFFFFFF)
    s = np.minimum(s, e - 1).clip(0)
    f = rng.permutation(R).astype(np.uint32) * 3 + 1
    if R and rng.random() < 0.3:
        f[rng.integers(0, R, max(1, R // 10))] = f[0]  # duplicate fids
    if R and rng.random() < 0.3:  # empty intervals (end == start): the reference keeps them when qs < start < qe
        e = np.where(rng.random(R) < 0.05, s, e)  # (end < start is outside the domain: IntervalTree::build recurses forever, tree.rs:48-50)
    s, e = s.astype(np.uint32), e.astype(np.uint32)
    nq = int(rng.choice([1, 5, 63, 64, 65, 2047, 2048, 2049, 6000]))
    qc = rng.integers(0, n_chr, nq).astype(np.uint32)
    qs = rng.integers(0, min(span + span // 10 + 2, 0xFFFFFFFF), nq, dtype=np.int64)
    kind = rng.random(nq)
    w = np.where(kind < 0.6, rng.integers(1, max(2, span // 3000 + 2), nq),
                 np.where(kind < 0.8, rng.integers(1, span + 2, nq), np.where(kind < 0.9, 0, -rng.integers(1, 50, nq))))
    qe = np.clip(qs + w, 0, 0xFFFFFFFF)
    if R and rng.random() < 0.5:  # regions touching interval boundaries exactly
        pick = rng.integers(0, R, nq // 4 + 1)
        qs[: len(pick)] = s[pick][: nq]
        qe[: len(pick)] = e[pick][: nq]
        c_of = np.searchsorted(co, pick, side="right") - 1
        qc[: len(pick)] = c_of[: nq]
    regions = np.stack([qc, qs.astype(np.uint32), qe.astype(np.uint32)], axis=1)
    oix = ob.OracleIndex.from_roots(co, s, e, f)
    ix = engine.TreeIndexData.from_roots(co, s, e, f)
    b = engine.QueryBatch(ix, nq)
    b.set_regions(regions) if it & 1 else b.set_regions_soa(regions[:, 0], regions[:, 1], regions[:, 2])
    by_chr = np.argsort(regions[:, 0], kind="stable")
    for mode in (0, 1, 2):
        for inv in (False, True):
            want_t, want_c = oix.query_features(regions, mode, inv)
            wc = want_c.astype(np.int64)
            want_pairs = np.stack([np.repeat(by_chr, wc[by_chr]), want_t[:, 0].astype(np.int64)], axis=1)
            want_pairs = want_pairs[np.lexsort((want_pairs[:, 1], want_pairs[:, 0]))]
            qid = np.repeat(np.arange(nq, dtype=np.int64), wc)
            within = np.arange(len(qid), dtype=np.int64) - np.repeat(np.cumsum(wc) - wc, wc)
            for strat in STRATS:
                if strat == engine.STRATEGY_SORTED and n_chr > 4000:
                    continue
                flag_sets = [engine.OUT_FIDS | engine.OUT_OFFSETS, engine.OUT_TRIPLES | engine.OUT_ROOT_BITMAP | engine.OUT_OFFSETS]
                if strat == engine.STRATEGY_WINDOWS:  # u32 offsets next to the u64 ones; the CLI's bitmap-only pass
                    flag_sets += [engine.OUT_FIDS | engine.OUT_OFFSETS | engine.OUT_OFFSETS32, engine.OUT_ROOT_BITMAP,
                                  engine.OUT_FIDS | engine.OUT_SEGBASE]  # (the pass bench.py times: one base per 256 regions)
                    if it % 3 == 0:  # the 1024-thread variant of the pair pass (the engine takes it for batches of 500 000 regions or more that run alone)
                        b.set_option("WIN_THREADS", 1024)
                    else:
                        b.set_option("WIN_THREADS", 0)
                runs = [(fl, False) for fl in flag_sets]
                if strat == engine.STRATEGY_WINDOWS and not (mode == 2 and inv):  # (every mode, inverted or not: round 5)
                    # the wide form of the pair and root passes (regions of any width from two lines and two ranks; AUTO's choice for wide
                    # batches), forced on these regions whatever their widths
                    runs += [(fl, True) for fl in (engine.OUT_FIDS | engine.OUT_OFFSETS, engine.OUT_TRIPLES | engine.OUT_OFFSETS, 0,
                                                   engine.OUT_FIDS | engine.OUT_OFFSETS | engine.OUT_OFFSETS32, engine.OUT_FIDS | engine.OUT_SEGBASE,
                                                   engine.OUT_ROOT_BITMAP, engine.OUT_TRIPLES | engine.OUT_ROOT_BITMAP | engine.OUT_OFFSETS)]
                for flags, wide in runs:
                    b.set_option("WIN_WIDE", 2 if wide else 1)
                    b.run(mode, inv, flags, strat)
                    b.wait()
                    if b.wide_form and not wide:
                        print("wide form taken unasked, iteration", it, "flags", flags, flush=True)
                        sys.exit(1)
                    n_wide += b.wide_form  # (an index without windows, or with an interval that ends before it starts, has no wide form)
                    c = b.counts()
                    ok = np.array_equal(c, want_c) and b.total_hits == len(want_t)
                    if flags == engine.OUT_ROOT_BITMAP:
                        ok = ok and np.array_equal(b.unique_roots(), np.unique(want_t[:, 0]))
                        n_checks += 1
                        if ok:
                            continue
                    if flags == 0:  # counts alone
                        n_checks += 1
                        if ok:
                            continue
                    if flags & engine.OUT_SEGBASE:
                        off = np.concatenate([b.offsets_from_segbase(c), [np.uint64(b.total_hits)]]).astype(np.uint64) if ok else None
                    else:
                        off = b.offsets() if ok or flags not in (0, engine.OUT_ROOT_BITMAP) else None
                    if ok and flags & engine.OUT_OFFSETS32:
                        ok = np.array_equal(off[:-1], b.offsets32().astype(np.uint64))
                    if ok and flags & engine.OUT_FIDS:
                        got = np.stack([qid, b.fids()[off[:-1].astype(np.int64)[qid] + within].astype(np.int64)], axis=1)
                        ok = np.array_equal(got[np.lexsort((got[:, 1], got[:, 0]))], want_pairs)
                    if ok and flags & engine.OUT_TRIPLES:
                        t = b.triples()
                        srt = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]  # noqa: E731
                        ok = np.array_equal(srt(t), srt(want_t)) and (not (flags & engine.OUT_ROOT_BITMAP) or
                                                                       np.array_equal(b.unique_roots(), np.unique(want_t[:, 0])))
                    n_checks += 1
                    if not ok:
                        os.makedirs("gpurun_out", exist_ok=True)
                        np.savez("gpurun_out/fuzz_fail.npz", co=co, s=s, e=e, f=f, regions=regions)
                        print("MISMATCH iteration", it, "strategy", strat, "mode", mode, "invert", inv, "flags", flags, "wide form", wide,
                              "n_chr", n_chr, "R", R, "nq", nq, "shape", shape, "span", span, flush=True)
                        sys.exit(1)
    # round 6: the same index, 2-6 batches of slices / shuffles / sorted copies of the regions handed over TOGETHER (one launch per
    # group: engine.run_batches), pair passes, triples + roots and root passes, the narrow and the forced mixed form -- every batch
    # against the oracle on ITS regions
    ng = int(rng.choice([2, 3, 4, 6]))
    gsets = []
    for g in range(ng):
        kind_g = g % 3
        sub = regions[rng.permutation(nq)[: max(1, int(rng.integers(1, nq + 1)))]]
        if kind_g == 1:
            sub = sub[np.lexsort((sub[:, 1], sub[:, 0]))]  # sorted by (seqid, start)
        gsets.append(np.ascontiguousarray(sub))
    gb = []
    for r_ in gsets:
        bb = engine.QueryBatch(ix, len(r_))
        bb.set_regions(r_)
        gb.append(bb)
    gb[0].set_option("GROUP", int(rng.choice([1, 2, 3])))
    for mode in (0, 1, 2):
        inv = bool(rng.integers(0, 2)) and mode != 2
        for flags, wide in ((engine.OUT_FIDS | engine.OUT_OFFSETS, False), (engine.OUT_FIDS | engine.OUT_SEGBASE, True),
                            (engine.OUT_TRIPLES | engine.OUT_ROOT_BITMAP | engine.OUT_OFFSETS, False), (engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, bool(it & 1))):
            for bb in gb:
                bb.set_option("WIN_WIDE", 2 if wide else 1)
            engine.run_batches(gb, mode, inv, flags, engine.STRATEGY_WINDOWS, n_passes=2 * ng + 1)
            for bb, r_ in zip(gb, gsets):
                bb.wait()
                wt, wcn = oix.query_features(r_, mode, inv)
                ok = bb.total_hits == len(wt)
                if not (flags & engine.OUT_NO_COUNTS):
                    ok = ok and np.array_equal(bb.counts(), wcn)
                if ok and (flags & engine.OUT_ROOT_BITMAP):
                    ok = np.array_equal(bb.unique_roots(), np.unique(wt[:, 0]))
                if ok and (flags & engine.OUT_FIDS):
                    c_ = wcn.astype(np.int64)
                    off = (bb.offsets_from_segbase(bb.counts()) if flags & engine.OUT_SEGBASE else bb.offsets()[:-1]).astype(np.int64)
                    qid_ = np.repeat(np.arange(len(r_), dtype=np.int64), c_)
                    within_ = np.arange(len(qid_), dtype=np.int64) - np.repeat(np.cumsum(c_) - c_, c_)
                    got = np.stack([qid_, bb.fids()[off[qid_] + within_].astype(np.int64)], axis=1)
                    bc = np.argsort(r_[:, 0], kind="stable")
                    want = np.stack([np.repeat(bc, c_[bc]), wt[:, 0].astype(np.int64)], axis=1)
                    ok = np.array_equal(got[np.lexsort((got[:, 1], got[:, 0]))], want[np.lexsort((want[:, 1], want[:, 0]))])
                if ok and (flags & engine.OUT_TRIPLES):
                    srt = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]  # noqa: E731
                    ok = np.array_equal(srt(bb.triples()), srt(wt))
                n_checks += 1
                n_group += 1
                if not ok:
                    os.makedirs("gpurun_out", exist_ok=True)
                    np.savez("gpurun_out/fuzz_fail.npz", co=co, s=s, e=e, f=f, regions=r_)
                    print("MISMATCH (batches in one launch) iteration", it, "mode", mode, "invert", inv, "flags", flags, "wide form", wide, "batches", ng,
                          "n_chr", n_chr, "R", R, "nq", len(r_), flush=True)
                    sys.exit(1)
    for bb in gb:
        bb.close()
    b.close()
    ix.close()
print("fuzz ok: %d iterations, %d passes checked (%d of them in the wide form, %d of batches served in groups), %.0f s" % (iters, n_checks, n_wide, n_group, time.time() - t0))
