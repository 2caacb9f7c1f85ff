#!/bin/bash
# round 6: (1) DVFS check of the persistent convolution (random / zero activations / all zero), (2) kernel trace of the default training step, per queue
O=gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
VD_K32P_LAG=0 python tools/k32p_probe.py > $O/k32p_dvfs_random.txt 2>&1
VD_K32P_LAG=0 K32P_ZERO=1 python tools/k32p_probe.py > $O/k32p_dvfs_zero_x.txt 2>&1
VD_K32P_LAG=0 K32P_ZERO=2 python tools/k32p_probe.py > $O/k32p_dvfs_zero_all.txt 2>&1
VD_K32P_LAG=0 python tools/k32p_probe.py > $O/k32p_dvfs_random2.txt 2>&1
paste <(cut -c1-52 $O/k32p_dvfs_random.txt) <(cut -c22-52 $O/k32p_dvfs_zero_x.txt) <(cut -c22-52 $O/k32p_dvfs_zero_all.txt) <(cut -c22-52 $O/k32p_dvfs_random2.txt)
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_train -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 4 --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path > $GRAFT_REPO_ROOT/$O/trace_train.json 2> $GRAFT_REPO_ROOT/$O/trace_train.err
cd $GRAFT_REPO_ROOT
python3 tools/trace_queues.py $O/trace_train 4 > $O/trace_queues.txt 2>&1
python3 tools/trace_gaps.py $O/trace_train 4 > $O/trace_gaps.txt 2>&1
rm -rf $O/trace_train
cat $O/trace_queues.txt; head -40 $O/trace_gaps.txt; tail -3 $O/trace_train.json | cut -c1-600
