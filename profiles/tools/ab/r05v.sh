#!/bin/bash
# round 5, step v: the depth sort queued before the host's round trip (speculative plan), VEGS_SPEC_SORT=0/1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_stress.py tests/test_gpu_training.py tests/test_gpu_views.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2 3; do
  for val in 0 1; do
    VEGS_SPEC_SORT=$val timeout 300 python bench.py --stages --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d['roofline']['stage_ms']
print('VEGS_SPEC_SORT=$val', d['ms_per_step'], {k: st.get(k) for k in ('compact','depth_sort','emit','tile_sort')})"
  done
done
