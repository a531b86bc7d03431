#!/bin/bash
# usage: gpuretry.sh <timeout> <script>   -- retries while the pool is busy (exit 3)
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "bash $2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
