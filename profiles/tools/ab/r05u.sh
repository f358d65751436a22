#!/bin/bash
# round 5, step u: light tiles (2 ... 8 needed segments) finished by k_seg_blend, VEGS_SEG_LIGHT=0/1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2 3; do
  for val in 0 1; do
    VEGS_SEG_LIGHT=$val timeout 300 python bench.py --stages --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d['roofline']['stage_ms']
print('VEGS_SEG_LIGHT=$val', d['ms_per_step'], {k: st.get(k) for k in ('render_fwd','render_bwd','k_seg_bwd')})"
  done
done
VEGS_SEG_LIGHT=1 bash profiles/tools/kstats.sh light1 2>&1 | grep -E "k_seg_blend|k_seg_combine|ms_per_step" | cut -c1-120
