#!/bin/bash
# A/B of whole library builds on one box, interleaved (run ON the GPU box):
#   bash profiles/tools/ab/libs_ab.sh <tries> <name> <name> ...     name = default | a vegs_amd/_lib/libvegsrast_<name>.so
# (built beforehand: python -m vegs_amd.build --variant <name>, or profiles/tools/ab/build_at.sh <commit> <name>)
tries=$1; shift
for t in $(seq 1 $tries); do
  for lib in "$@"; do
    if [ $lib = default ]; then unset VEGS_LIB; else export VEGS_LIB=$PWD/vegs_amd/_lib/libvegsrast_$lib.so; fi
    python bench.py --stages --no-variants --no-cpu-baseline --steps 20 --warmup 4 ${BENCH_ARGS} > /tmp/o.json 2> /tmp/e.log; rc=$?
    echo "$lib try $t rc=$rc $(python -c "import json;d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1]);s=d['roofline']['stage_ms'];print(d['ms_per_step'], {k: round(v, 4) for k, v in s.items()})" 2>/dev/null)"
  done
done
