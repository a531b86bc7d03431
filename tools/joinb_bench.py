#!/usr/bin/env python3
"""Join B on the GPU box: HIP-event times of the device preparation of the region tables and of k_lines_exists, on
bench.py's Join B workload (the 3.4 M lines of the synthetic GENCODE-like annotation x N regions of synth_bed).
    python tools/joinb_bench.py [--quick] [n_regions ...]     (default: 1000000)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gffx_amd import engine, synth  # noqa: E402
from gffx_amd.engine import OverlapMode  # noqa: E402


def main():
    quick = "--quick" in sys.argv  # Overlap mode on the plain BED only (for rocprofv3 runs)
    sizes = [int(x) for x in sys.argv[1:] if not x.startswith("--")] or [1_000_000]
    roots = synth.gencode_like_roots(63000, seed=42)
    tab = synth.gencode_like_block_table(roots)
    per_block = np.diff(tab["block_line_off"]).astype(np.int64)
    chr_of_root = np.repeat(np.arange(len(roots["chr_offsets"]) - 1), np.diff(roots["chr_offsets"]))
    seq = np.repeat(chr_of_root, per_block).astype(np.uint32)
    lt = engine.LineTable(seq, tab["line_start"] + 1, tab["line_end"])
    n_seq = len(roots["chr_offsets"]) - 1
    for nq in sizes:
        regions = synth.synth_bed(nq, seed=1001, roots=roots)
        zl = regions.copy()  # every 10th row zero-length: start + 1 > end
        zl[::10, 2] = zl[::10, 1] - 1
        for name, regs in (("bed", regions), ("bed_10pct_zero_length", zl))[: 1 if quick else 2]:
            for mode in ([OverlapMode.Overlap] if quick else OverlapMode):
                kept = lt.test(regs, n_seq, mode)
                us, prep = [], []
                for _ in range(5):
                    lt.test(regs, n_seq, mode)
                    us.append(1e3 * lt.last_kernel_ms)
                    prep.append(1e3 * lt.last_prep_ms)
                k = float(np.median(us))
                print(json.dumps({"regions": nq, "input": name, "mode": mode.name, "lines": int(lt.n), "kernel_us": round(k, 2),
                                  "prep_us": round(float(np.median(prep)), 1), "sort_passes": lt.last_sort_passes, "prep_us_per_1m": round(float(np.median(prep)) * 1e6 / nq, 1),
                                  "GBps": round(13.0 * lt.n / (k * 1e-6) / 1e9, 1), "frac_of_8TBps": round(13.0 * lt.n / (k * 1e-6) / 8e12, 3),
                                  "kept": int(kept.sum())}), flush=True)
    lt.close()


if __name__ == "__main__":
    main()
