#!/usr/bin/env python3
"""profiles/traffic_latest.json from a tools/profile_pmc.sh counter summary: per-kernel FETCH_SIZE / WRITE_SIZE
(KiB per dispatch) -> HBM bytes per launch and per pass, corrected as MI355X_MICROARCH.md prescribes
(separate --pmc passes; FETCH_SIZE doubled: gfx950 reports half the bytes of a wide read).

    python tools/traffic_from_pmc.py gpurun_out/<tag>_pmc.txt k_join_pairs [k_other ...] > profiles/traffic_latest.json

The file is stamped with the commit it was measured at (GFFX_COMMIT, else `git rev-parse`): bench.py prints that stamp next
to the number, so a stale figure is visible as such.
"""
import json
import os
import re
import subprocess
import sys


def main(path, kernels):
    cur, vals = None, {}
    for ln in open(path):
        ln = ln.rstrip("\n")
        if ln and not ln.startswith(" ") and not ln.startswith("#"):
            cur = ln
            continue
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches=(\d+)\s+avg=([\d.]+)", ln)
        if m and cur:
            for k in kernels:
                if ("gffx::" + k + "<") in cur or ("gffx::" + k + "(") in cur:
                    vals.setdefault(k, {})[m.group(1) + "_KiB"] = float(m.group(3))
    out, total = {}, 0.0
    for k, v in vals.items():
        b = 2.0 * v.get("FETCH_SIZE_KiB", 0.0) * 1024 + v.get("WRITE_SIZE_KiB", 0.0) * 1024
        v["hbm_bytes_corrected"] = b
        out[k] = v
        total += b
    out["hbm_bytes_per_pass"] = total
    out["kernel"] = "+".join(sorted(vals))
    commit = os.environ.get("GFFX_COMMIT")
    if not commit:
        try:
            commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
        except Exception:
            commit = None
    out["commit"] = commit
    out["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_pmc.sh); per-dispatch "
                   "averages; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of a wide read); "
                   "the batch is re-read every pass, so Infinity-Cache hits are included (the guide: they are counted)")
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
