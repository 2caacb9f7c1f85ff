#!/bin/bash
# gpurun with retries while no GPU slot is free (exit 3 / "transient": nothing is charged).  usage: tools/gpurun_retry.sh <timeout> '<command>'
for i in $(seq 1 40); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"
  exit 0
done
echo "$out"
exit 3
