#!/bin/bash
cp vegs_amd/_lib/libvegsrast.so /tmp/lib_keep.so
cp profiles/tools/ab/libvegsrast_w32fb.so vegs_amd/_lib/libvegsrast.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
cp /tmp/lib_keep.so vegs_amd/_lib/libvegsrast.so
bash profiles/tools/ab/lib_ab.sh 3 default w32 fb w32fb
