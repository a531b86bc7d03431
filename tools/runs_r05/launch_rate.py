import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from gffx_amd import engine, synth
import bench
roots = synth.gencode_like_roots(63000, seed=42)
ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
dev = torch.device("cuda", 0)
flags = engine.OUT_FIDS | engine.OUT_SEGBASE
for nq in (1_000_000, 250_000, 4096):
    regs = synth.synth_bed(nq, seed=1001)
    cols = bench.to_dev(torch, regs, dev)
    for infl in (1, 2, 3, 4):
        p = bench.Pass(engine, ix, cols, nq, infl, 2, flags, 0)
        p.size_and_warm(2)
        best = None
        for rep in range(3):
            p.run_n(2 * infl); p.sync(); torch.cuda.synchronize()
            n = 600
            t0 = time.perf_counter(); p.run_n(n); t1 = time.perf_counter(); p.sync(); torch.cuda.synchronize(); t2 = time.perf_counter()
            r = ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6)
            best = r if best is None or r[1] < best[1] else best
        print("nq %8d inflight %d threads %4d: host enqueue %6.2f us per pass, total %6.2f us per pass" % (nq, infl, p.batches[0].block_threads, best[0], best[1]), flush=True)
        for bb in p.batches: bb.close()
