#!/bin/bash
# k_seg_bwd with parts compiled out at run time (VEGS_BWD_ABLATE: 1 no atomics, 2 no pixel loop, 3 prologue + quadrant
# lists only, 4 loads + compaction only): where does the time go?   gpurun -- 'bash profiles/tools/bwd_ablate.sh'
for a in 0 1 2 3 4; do
  VEGS_BWD_ABLATE=$a python bench.py --no-cpu-baseline --no-variants --repeats 1 --steps 16 --stages 2>&1 >/dev/null | grep "stage breakdown" | sed "s/.*'render_bwd'/ablate $a: render_bwd/"
done
