#!/bin/bash
# round 5, step j: the split + SH-tail layout on the wave-staged k_preprocess paths (C5 iteration: 5M + 8 boxes)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_instances.py tests/test_gpu_training.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_dist_train.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2 3; do
  for val in 0 1; do
    VEGS_PRE_HALF=$val PYTHONPATH=. timeout 300 python profiles/tools/iteration_bench.py --gaussians 5000000 --boxes 8 --iters 16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C5 VEGS_PRE_HALF=$val', {k: d[k] for k in d if k.endswith('_ms')})"
  done
done
