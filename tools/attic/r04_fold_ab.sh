#!/bin/bash
# round 4: training forward without normalise passes (act_out side output) -- tests + same-box A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_unet_gpu.py tests/test_headline_parity_gpu.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/fold_tests.txt
python -m pytest tests/test_hip_kernels.py tests/test_train_sample_gpu.py -x -q -m gpu 2>&1 | tail -5 >> gpurun_out/fold_tests.txt
for r in 1 2; do for f in 0 1; do
  echo "fold=$f" >> gpurun_out/fold_ab.txt
  VILLAN_FOLD_GN_TRAIN=$f python bench.py --mode train --no-exact --no-cpu --no-f16 --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/fold_ab.txt
done; done
cat gpurun_out/fold_tests.txt gpurun_out/fold_ab.txt
