#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -3
bash profiles/tools/ab/lib_ab.sh 3 default chain1
