"""root passes (GFFX_OUT_ROOT_BITMAP | NO_COUNTS | BITMAP_KEEP: what a streaming caller runs) with 1 - 4 batches in flight, the grid forced"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from gffx_amd import engine, synth
import bench
roots = synth.gencode_like_roots(63000, seed=42)
ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
dev = torch.device("cuda", 0)
flags = engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS | engine.OUT_BITMAP_KEEP
regs = synth.synth_bed(1_000_000, seed=1001)
cols = bench.to_dev(torch, regs, dev)
for blocks in (0, 256):
    for infl in (1, 2, 3, 4):
        p = bench.Pass(engine, ix, cols, len(regs), infl, 2, flags, 0)
        for bb in p.batches:
            bb.set_option("BITMAP_BLOCKS", blocks)
            bb.set_option("WIN_THREADS", 512 if infl > 1 else 0)
        p.size_and_warm(2)
        best = None
        for rep in range(3):
            p.run_n(2 * infl); p.sync(); torch.cuda.synchronize()
            n = 600
            t0 = time.perf_counter(); p.run_n(n); p.sync(); torch.cuda.synchronize(); t2 = time.perf_counter()
            us = (t2 - t0) / n * 1e6
            best = us if best is None or us < best else best
        print("root pass, BITMAP_BLOCKS %3d, %d in flight: %6.2f us per pass" % (blocks, infl, best), flush=True)
        for bb in p.batches: bb.close()
