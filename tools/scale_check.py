"""Scale check on the GPU box (development tool): a batch far larger than the bench's (default 50 M regions) through the
slots, the fused and the partitioned strategy (the latter in several sub-batches: its tile regions exceed the workspace budget),
with size-independent properties and sampled oracle parity.  python tools/scale_check.py [n_regions]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gffx_amd import engine, synth
from oracle import binding as ob

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
roots = synth.gencode_like_roots(63000, seed=42)
t0 = time.perf_counter()
regions = np.concatenate([synth.synth_bed(min(10_000_000, n - a), seed=2000 + a) for a in range(0, n, 10_000_000)])
print("generated %d regions in %.1f s" % (len(regions), time.perf_counter() - t0), flush=True)
co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
ix = engine.TreeIndexData.from_roots(co, s, e, f)
oix = ob.OracleIndex.from_roots(co, s, e, f)
sel = np.random.default_rng(1).choice(n, size=20000, replace=False)
want_t, want_c = oix.query_features(regions[sel], 2, False)
b = engine.QueryBatch(ix, n)
b.set_regions(regions)
for name, strat, flags in (("windows", engine.STRATEGY_WINDOWS, engine.OUT_FIDS | engine.OUT_OFFSETS),
                           ("fused", engine.STRATEGY_FUSED, engine.OUT_FIDS | engine.OUT_OFFSETS),
                           ("partitioned", engine.STRATEGY_SORTED, engine.OUT_FIDS | engine.OUT_OFFSETS)):
    t0 = time.perf_counter(); b.run(2, False, flags, strat); b.wait(); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); b.run(2, False, flags, strat); b.wait(); dt2 = time.perf_counter() - t1
    c = b.counts(); off = b.offsets(); fids = b.fids()
    assert int(c.sum(dtype=np.uint64)) == b.total_hits == int(off[-1]) == len(fids)
    assert np.array_equal(c[sel], want_c)
    got = np.concatenate([fids[int(off[i]):int(off[i]) + int(c[i])] for i in sel[:2000]])
    one_t, _ = oix.query_features(regions[sel[:2000]], 2, False)
    assert np.array_equal(np.sort(got), np.sort(one_t[:, 0]))
    print("%s: %d regions, %d pairs, first pass %.1f ms, second %.1f ms (%.1f G regions/s incl. host sync)"
          % (name, n, b.total_hits, 1e3 * dt, 1e3 * dt2, n / dt2 / 1e9), flush=True)
print("scale check ok")
