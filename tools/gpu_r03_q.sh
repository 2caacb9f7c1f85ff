#!/bin/bash
for rep in 1 2 3; do
  VD_BX3_K32_UP32=0 python tools/up_probe.py 2>&1 | grep up-conv | sed 's/^/old  /'
  VD_BX3_K32_UP32=1 python tools/up_probe.py 2>&1 | grep up-conv | sed 's/^/k32  /'
done
