#!/bin/bash
# quick correctness gate for an experimental library: profiles/tools/ab/check.sh <name>
cp profiles/tools/ab/lib_$1.so vegs_amd/_lib/libvegsrast.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -2
VEGS_FUZZ_SEEDS=0:120 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -1
