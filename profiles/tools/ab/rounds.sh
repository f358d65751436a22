#!/bin/bash
# needs the EXPERIMENT build of profiles/experiments/r04_sweep_env_overrides.patch.txt (VEGS_EXP_* are not read by the tree's library)
# sweep of the segment-round constants (experiment build with VEGS_EXP_* overrides): render_fwd per disc scale
run() { sc=$1; cfg=$2
  if [ "$cfg" = "off" ]; then f=1024; a=6; b=0; else f=2048; set -- $cfg; a=$1; b=$2; fi
  r=$(VEGS_RAST_FLAGS=$f VEGS_EXP_FIRST=$a VEGS_EXP_SECOND=$b python bench.py --stages --no-variants --no-cpu-baseline --disc-scale $sc --repeats 3 2>&1 | grep "stage breakdown\|R_lists" | sed "s/.*'render_fwd': \([0-9.]*\).*'render_bwd': \([0-9.]*\).*/fwd \1 bwd \2/; s/.*\"R_lists\": \([0-9.]*\).*/R_lists \1/" | tr '\n' ' ')
  echo "scale $sc  first/second $cfg: $r"; }
for cfg in "off" "6 48" "4 48" "8 48" "6 64" "4 24"; do run 1.5 "$cfg"; done
for cfg in "6 64" "5 48" "8 48" "4 48" "6 32"; do run 2 "$cfg"; done
for cfg in "3 6" "2 8" "3 12" "2 4" "2 16"; do run 3 "$cfg"; done
for cfg in "off" "6 24" "3 8" "2 8" "3 16" "2 4"; do run 5 "$cfg"; done
