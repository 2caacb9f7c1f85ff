#!/bin/bash
# one gpurun call, several pytest invocations, each with its own limit and its own streamed log (a killed call keeps what finished)
O=gpurun_out/r02
mkdir -p $O
run() { name=$1; lim=$2; shift 2; timeout $lim python -m pytest "$@" -q -s -m gpu -p no:cacheprovider > $O/$name.log 2>&1; echo "$name rc=$?" >> $O/summary.log; grep -E "\[parity\]|passed|failed|Error|error" $O/$name.log | tail -25; }
: > $O/summary.log
"$@"
cat $O/summary.log
