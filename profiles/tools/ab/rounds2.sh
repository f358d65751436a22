#!/bin/bash
# needs the EXPERIMENT build of profiles/experiments/r04_sweep_env_overrides.patch.txt (VEGS_EXP_* are not read by the tree's library)
run() { sc=$1; cfg=$2
  if [ "$cfg" = "off" ]; then f=1024; a=6; b=0; else f=2048; set -- $cfg; a=$1; b=$2; fi
  r=$(VEGS_RAST_FLAGS=$f VEGS_EXP_FIRST=$a VEGS_EXP_SECOND=$b python bench.py --stages --no-variants --no-cpu-baseline --disc-scale $sc --repeats 3 2>&1 | grep "stage breakdown" | sed "s/.*'render_fwd': \([0-9.]*\).*'render_bwd': \([0-9.]*\).*/fwd \1 bwd \2/" | tr '\n' ' ')
  echo "scale $sc  first/second $cfg: $r"; }
for cfg in "off" "6 64" "6 96" "8 64" "6 128" "10 64" "12 96" "off"; do run 1 "$cfg"; done
for cfg in "6 64" "6 96" "8 64" "6 128"; do run 1.5 "$cfg"; done
for cfg in "6 64" "4 64" "4 32"; do run 2 "$cfg"; done
