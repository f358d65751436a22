#!/bin/bash
# needs the EXPERIMENT build of profiles/experiments/r04_sweep_env_overrides.patch.txt (VEGS_EXP_* are not read by the tree's library)
# emission of the rectangles of more than 64 tiles: inline (VEGS_EXP_INLINE=100000) or listed for k_emit_big (=0), per disc scale
for sc in 1 1.5 2 3; do
for thr in 100000 0; do
  r=$(VEGS_EXP_INLINE=$thr python bench.py --stages --no-variants --no-cpu-baseline --disc-scale $sc --repeats 3 2>&1 | grep "stage breakdown\|R_lists" | sed "s/.*'emit': \([0-9.]*\).*/emit \1/; s/.*\"V\": \([0-9.]*\), \"R\": [0-9.]*, \"R_lists\": \([0-9.]*\).*/V \1 R_lists \2/" | tr '\n' ' ')
  echo "scale $sc inline_thr $thr: $r"
done
done
