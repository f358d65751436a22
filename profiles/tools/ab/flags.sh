#!/bin/bash
# stage times under VrFlags given as arguments (0 = default):  flags.sh "<bench args>" 0 2048 ...
args=$1; shift
for rep in 1 2; do
for f in "$@"; do
  echo "== flags $f"; VEGS_RAST_FLAGS=$f python bench.py --stages --no-variants --no-cpu-baseline $args 2>&1 | grep "stage breakdown" | sed 's/.*breakdown//' | cut -c1-330
done
done
