#!/bin/bash
# round 5, step m: chain mode of k_seg_alpha (walkers inside the launch) against the classic order, VEGS_SEG_CHAIN=0/1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2 3; do
  for val in 0 1; do
    VEGS_SEG_CHAIN=$val timeout 300 python bench.py --stages --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d['roofline']['stage_ms']
print('VEGS_SEG_CHAIN=$val', d['ms_per_step'], {k: st[k] for k in ('render_fwd','render_bwd','k_seg_bwd')})"
  done
done
bash profiles/tools/ab/lib_ab.sh 2 default k16 k48
