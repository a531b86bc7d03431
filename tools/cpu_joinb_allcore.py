#!/usr/bin/env python3
"""Reference-style CPU number for Join B on all cores: the reference runs gff_line_overlaps_queries on rayon's global pool
(all cores: CommonArgs::init_rayon is never called) and scans ALL regions of the line's seqid per line
(commands/intersect.rs:266-329, :500-521).  Here: the oracle's literal scan (oracle_line_predicate) over a bounded random
sample of the lines of a GENCODE-shaped 3.4 M-line table, one forked worker per core.  Run as its own process by bench.py
(nothing forks from a process that initialised the GPU).  Prints one JSON line.
    python tools/cpu_joinb_allcore.py [n_regions] [mode] [budget_seconds]"""
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

_G = {}


def _work(lines):
    seq, s, e, off, qs, qe, mode = _G["seq"], _G["s"], _G["e"], _G["off"], _G["qs"], _G["qe"], _G["mode"]
    kept = 0
    for i in lines:
        c = int(seq[i])
        kept += bool(ob.line_predicate(int(s[i]), int(e[i]), qs[off[c]:off[c + 1]], qe[off[c]:off[c + 1]], mode))
    return kept


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
    cores = os.cpu_count() or 1
    roots = synth.gencode_like_roots(63000, seed=42)
    regions = synth.synth_bed(nq, seed=1001)
    n_seq = len(roots["chr_offsets"]) - 1
    tab = synth.gencode_like_block_table(roots)
    per_block = np.diff(tab["block_line_off"]).astype(np.int64)
    chr_of_root = np.repeat(np.arange(n_seq), np.diff(roots["chr_offsets"]))
    order = np.argsort(regions[:, 0], kind="stable")
    r = regions[order]
    _G.update(seq=np.repeat(chr_of_root, per_block).astype(np.uint32), s=tab["line_start"] + 1, e=tab["line_end"],
              off=np.concatenate([[0], np.cumsum(np.bincount(r[:, 0], minlength=n_seq))]),
              qs=np.ascontiguousarray(r[:, 1]), qe=np.ascontiguousarray(r[:, 2]), mode=mode)
    n_lines = len(_G["seq"])
    rng = np.random.default_rng(5)
    ob.line_predicate(1, 2, _G["qs"][:10], _G["qe"][:10], mode)  # load the library before forking
    with mp.get_context("fork").Pool(cores) as pool:
        per = 40  # lines per worker and dispatch: ~40 x 40 k regions x ~1 ns
        pool.map(_work, [rng.choice(n_lines, size=4, replace=False).tolist() for _ in range(cores)])  # warm
        done, used, kept = 0, 0.0, 0
        while used < budget:
            jobs = [rng.choice(n_lines, size=per, replace=False).tolist() for _ in range(cores)]
            t0 = time.perf_counter()
            kept += sum(pool.map(_work, jobs))
            used += time.perf_counter() - t0
            done += per * cores
            if used < 0.5 * budget / 4:
                per *= 2
    print(json.dumps({"value": done / used, "unit": "lines/s", "cores": cores, "kind": "port",
                      "sample": "%d random lines of a %d-line table x all %d regions of the line's seqid (literal scan, "
                                "commands/intersect.rs:500-521), %d forked workers; whole-table time at this rate: %.1f s"
                                % (done, n_lines, nq, cores, n_lines / (done / used)),
                      "kept_fraction": kept / max(done, 1)}))


if __name__ == "__main__":
    main()
