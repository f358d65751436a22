#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -3
bash profiles/tools/ab/lib_ab.sh 2 default prev h16 h32 h64
bash profiles/tools/kstats.sh qdef 2>&1 | grep -E "k_seg_combine|k_seg_suffix|k_seg_scan|k_seg_alpha|ms_per_step" | cut -c1-120
