#!/usr/bin/env python3
"""bench.py's C2 data for tools/kbench.hip (KB_DATA=...) and tools/list_stats.hip: the GENCODE-shaped roots (seed 42) and the headline's
1 M regions (seed 1001) as one binary file {n_chr, n_roots, n_regions}, chr_offsets, starts, ends, region triples -- all u32.
    python tools/export_bench_data.py tools/_kb/data/bench_c2.bin
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gffx_amd import synth  # noqa: E402


def main(path, n_regions=1_000_000):
    roots = synth.gencode_like_roots(63000, seed=42)
    co, s, e = (roots[k].astype(np.uint32) for k in ("chr_offsets", "start", "end"))
    regions = synth.synth_bed(n_regions, seed=1001).astype(np.uint32)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        np.array([len(co) - 1, len(s), len(regions)], np.uint32).tofile(f)
        co.tofile(f), s.tofile(f), e.tofile(f), regions.tofile(f)
    print("%s: %d seqids, %d roots, %d regions" % (path, len(co) - 1, len(s), len(regions)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tools/_kb/data/bench_c2.bin")
