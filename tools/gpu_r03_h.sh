#!/bin/bash
# conv k32 ablations (timing only): what bounds the kernel?
for f in 0 1 2 4 6 8 14; do
  echo "== VD_BX3_K32_FLAGS=$f"; VD_BX3_K32_FLAGS=$f timeout 300 python tools/shape_probe.py conv3 2>&1 | grep -E "(384->  128|384-> 128|512-> 256|512->  256) @(32|16)\^2"
done
