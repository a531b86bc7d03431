import sys, os
sys.path.insert(0, os.getcwd())
import torch
from gffx_amd import engine, synth
import bench
print("package", engine.__file__, flush=True)
roots = synth.gencode_like_roots(63000, seed=42)
ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
dev = torch.device("cuda", 0)
regs = bench.widen_every(synth.synth_bed(1_000_000, seed=1001), 10)
for res in ("host", "device"):
    b = engine.QueryBatch(ix, len(regs))
    if res == "host":
        b.set_regions(regs)
    else:
        cols = bench.to_dev(torch, regs, dev)
        b.set_regions_device(cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), len(regs), keep=cols)
    for mode in (2, 0):
        for flags, fname in ((engine.OUT_FIDS | engine.OUT_SEGBASE, "fids+segbase"), (engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, "roots"), (engine.OUT_ROOT_BITMAP, "roots+counts")):
            for rep in range(2):
                b.run(mode, False, flags); b.wait()
                us = b.timed_runs(mode, False, flags, 0, 30)
                print("%-6s mode %d %-12s form %d threads %d: %8.2f us" % (res, mode, fname, b.wide_form, b.block_threads, us), flush=True)
    b.close()
