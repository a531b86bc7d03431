#!/bin/bash
# round 6, call 13: the whole GPU suite on the tree as it is, the fuzz campaigns with fresh seeds, and the rocpd schema (for a per-grid PMC summary)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_13.txt
: > $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 >> $O
timeout 1500 python tools/fuzz_parity.py 1200 6601 2>&1 | tail -2 >> $O
timeout 600 python tools/fuzz_lines.py 600 6602 2>&1 | tail -2 >> $O
timeout 900 python tools/fuzz_cli.py 60 6603 2>&1 | tail -2 >> $O
(cd /tmp && export TMPDIR=/tmp && cd $R && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES -d gpurun_out/r6_13_db -o run -- python3 bench.py --traffic-child --inflight 8 > /dev/null 2>&1; python3 - >> $O <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/r6_13_db/**/*_results.db", recursive=True)
if db:
    c = sqlite3.connect(db[0]).cursor()
    for t in ("counters_collection", "kernels"):
        try:
            print(t, [r[1] for r in c.execute("PRAGMA table_info(%s)" % t).fetchall()])
        except Exception as e:
            print(t, "error", e)
    try:
        print(c.execute("select * from counters_collection limit 1").fetchall())
    except Exception as e:
        print("error", e)
PY
rm -rf gpurun_out/r6_13_db)
cat $O | cut -c1-600
