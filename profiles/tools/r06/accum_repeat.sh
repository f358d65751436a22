#!/bin/bash
# (round 6) tests/test_gpu_accumulate.py's two-streams case repeated with several library builds:
#   bash profiles/tools/r06/accum_repeat.sh <tries> <name> ...        name = default | a vegs_amd/_lib/libvegsrast_<name>.so
tries=$1; shift
for lib in "$@"; do
  if [ $lib = default ]; then unset VEGS_LIB; else export VEGS_LIB=$PWD/vegs_amd/_lib/libvegsrast_$lib.so; fi
  for t in $(seq 1 $tries); do
    r=$(python -m pytest tests/test_gpu_accumulate.py -q -x -k "bit_for_bit" 2>&1 | grep -E "passed|failed|AssertionError: \(" | tr '\n' ' ' | cut -c1-200)
    echo "$lib try $t: $r"
  done
done
