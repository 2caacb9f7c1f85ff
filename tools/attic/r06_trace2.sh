#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_train -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 4 --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path > $GRAFT_REPO_ROOT/$O/trace_train.json 2> $GRAFT_REPO_ROOT/$O/trace_train.err
cd $GRAFT_REPO_ROOT
python3 tools/trace_queues.py $O/trace_train 4 200 > $O/trace_queues_full.txt 2>&1
rm -rf $O/trace_train
for r in 1 2; do
python bench.py --mode train --no-cpu --no-exact --no-f16 --no-roofline --no-ddp-path --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['value'], d['ms_per_step'], d.get('host_submit_ms_per_step'))" >> $O/train_now.txt
done
cat $O/train_now.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
