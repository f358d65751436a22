#!/bin/bash
# round 5, step s: the tile sort as one scatter (k_split_*) against the two onesweep passes, VEGS_TILE_SPLIT=0/1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2 3; do
  for val in 0 1; do
    VEGS_TILE_SPLIT=$val timeout 300 python bench.py --stages --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d['roofline']['stage_ms']
print('VEGS_TILE_SPLIT=$val', d['ms_per_step'], {k: st.get(k) for k in ('emit','tile_sort','ranges','render_fwd')})"
  done
done
if [ -f profiles/tools/ab/libvegsrast_coal.so ]; then
  cp vegs_amd/_lib/libvegsrast.so /tmp/lib_default.so; cp profiles/tools/ab/libvegsrast_coal.so vegs_amd/_lib/libvegsrast.so
  bash profiles/tools/kstats.sh coal 2>&1 | grep -E "k_split|ms_per_step" | cut -c1-100
  cp /tmp/lib_default.so vegs_amd/_lib/libvegsrast.so
fi
