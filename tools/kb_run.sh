#!/bin/bash
# run every kbench variant in tools/_kb (on the GPU box): tools/kb_run.sh "<blocks list>" [kbench args...]
cd "$(dirname "$0")/_kb"
BL="$1"; shift
for f in *; do for g in $BL; do echo "== $f JOIN_BLOCKS=$g $*"; GFFX_HIP_JOIN_BLOCKS=$g timeout 120 ./$f "$@" 2>&1 | grep -v "^nq="; done; done
