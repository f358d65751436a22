#!/bin/bash
# round 5, step s: the tile sort as one scatter (k_split_*) against the two onesweep passes, VEGS_TILE_SPLIT=0/1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -5
bash profiles/tools/ab/lib_ab.sh 3 default split1
bash profiles/tools/kstats.sh split2 2>&1 | grep -E "k_split|ms_per_step" | cut -c1-120
