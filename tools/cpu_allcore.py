#!/usr/bin/env python3
"""All-core CPU number for Join A (NOT the reference's behaviour: commands/intersect.rs:124-166 is serial): the oracle's
tree walk over the bench batch, the regions split evenly over one forked worker per core.  Run as its own process by
bench.py --cpu-allcore so that nothing forks from a process that has initialised the GPU.  Prints one JSON line."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gffx_amd import synth  # noqa: E402
from oracle import binding as ob  # noqa: E402

_OIX = None
_REG = None
_MODE = 2


_INNER = 20  # passes over the worker's slice per dispatch (set so that a dispatch is ~0.25 s of work: the pool's
             # round trip over a few hundred workers is ~10 ms)


def _work(sl):
    n = 0
    for _ in range(_INNER):
        t, _ = _OIX.query_features(_REG[sl[0]:sl[1]], _MODE, False)
        n = len(t)
    return n


def main():
    global _OIX, _REG, _MODE
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    _MODE = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
    cores = os.cpu_count() or 1
    roots = synth.gencode_like_roots(63000, seed=42)
    _REG = synth.synth_bed(nq, seed=1001)
    _OIX = ob.OracleIndex.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
    global _INNER
    step = -(-nq // cores)
    slices = [(a, min(nq, a + step)) for a in range(0, nq, step)]
    t0 = time.perf_counter()
    _OIX.query_features(_REG[slices[0][0]:slices[0][1]], _MODE, False)
    _OIX.query_features(_REG[slices[0][0]:slices[0][1]], _MODE, False)
    t_slice = max((time.perf_counter() - t0) / 2, 1e-6)
    _INNER = max(1, min(5000, int(0.25 / t_slice)))
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_work, slices)  # warm
        done, used, reps, pairs = 0, 0.0, 0, 0
        while used < budget and reps < 1000000:
            t0 = time.perf_counter()
            pairs = sum(pool.map(_work, slices))
            used += time.perf_counter() - t0
            done += nq * _INNER
            reps += _INNER
    print(json.dumps({"value": done / used, "unit": "queries/s", "cores": cores, "kind": "port",
                      "sample": "%d x the full %d-region batch split over %d forked workers (oracle tree walk per worker); "
                                "NOT the reference's behaviour, which runs this loop on one thread" % (reps, nq, cores),
                      "pairs_per_batch": pairs}))


if __name__ == "__main__":
    main()
