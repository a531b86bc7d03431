"""Development tool (GPU box): what the first calls of a fresh process cost through the C-ABI (HIP runtime +
context creation, code-object load, buffer allocation) -- the fixed part of every `gffx` CLI run.
python tools/first_call_costs.py [--no-warmup]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
t00 = time.perf_counter()
from gffx_amd import engine, synth
from gffx_amd._ffi import lib
L = lib()
roots = synth.gencode_like_roots(63000, seed=42)
regions = synth.synth_bed(1_000_000, seed=1001)
def lap(what, t0):
    print("%-46s %8.2f ms" % (what, 1e3 * (time.perf_counter() - t0)), flush=True)
if "--no-warmup" not in sys.argv:
    t0 = time.perf_counter(); rc = L.gffx_hip_warmup(0); lap("gffx_hip_warmup (rc %d)" % rc, t0)
    t0 = time.perf_counter(); rc = L.gffx_hip_warmup(0); lap("gffx_hip_warmup again", t0)
t0 = time.perf_counter()
ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
lap("index_create (63 k roots)", t0)
t0 = time.perf_counter(); b = engine.QueryBatch(ix, len(regions)); lap("batch_create (1 M regions)", t0)
t0 = time.perf_counter(); b.set_regions(regions); lap("set_regions_host", t0)
for i in range(3):
    t0 = time.perf_counter(); b.run(2, False, engine.OUT_ROOT_BITMAP); lap("run (root bitmap) #%d" % i, t0)
    t0 = time.perf_counter(); b.wait(); lap("wait #%d" % i, t0)
t0 = time.perf_counter(); bm = b.root_bitmap(); lap("copy_root_bitmap", t0)
t0 = time.perf_counter(); b.run(2, False, engine.OUT_FIDS | engine.OUT_OFFSETS); b.wait(); lap("run+wait (fids, offsets) first", t0)
t0 = time.perf_counter(); b.run(2, False, engine.OUT_FIDS | engine.OUT_OFFSETS); b.wait(); lap("run+wait (fids, offsets) again", t0)
t0 = time.perf_counter()
ix2 = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
lap("index_create again (warm process)", t0)
