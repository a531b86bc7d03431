#!/bin/bash
# Build kbench variants (compile-time knobs) into tools/_kb/ ; usage: tools/kb_build.sh name "-D..." [name "-D..."]...
cd "$(dirname "$0")/.." && mkdir -p tools/_kb
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -mllvm -amdgpu-atomic-optimizer-strategy=None $2 tools/kbench.hip -o tools/_kb/$1 &
  shift 2
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done
done
wait
ls tools/_kb
