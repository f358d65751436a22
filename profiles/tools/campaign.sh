#!/bin/bash
# The widened random sweeps of DESIGN.md section 6 in one call (GPU box; ~10 min):  bash profiles/tools/campaign.sh
mkdir -p gpurun_out
{
echo "== full GPU suite"; python -m pytest tests -q -m gpu -x 2>&1 | tail -3
echo "== fuzz 0:1200"; VEGS_FUZZ_SEEDS=0:1200 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -3
echo "== fuzz 0:400, full tile lists on both sides";  VEGS_FUZZ_SEEDS=0:400 VEGS_FUZZ_FLAGS=32768 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -3
echo "== fuzz 0:400, rounds on + scan binning + deterministic"; VEGS_FUZZ_SEEDS=0:400 VEGS_FUZZ_HIP_FLAGS=2816 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -3
echo "== densify 0:600"; VEGS_FUZZ_SEEDS=0:600 python -m pytest tests/test_gpu_densify.py -q -x -k random_settings 2>&1 | tail -3
echo "== street sweep"; PYTHONPATH=.:tests timeout 1500 python profiles/tools/sweep_street.py 2>&1 | tail -8
echo "== neighbours"; PYTHONPATH=. timeout 600 python profiles/tools/sweep_neighbours.py 500 2>&1 | tail -5
} > gpurun_out/campaign.txt 2>&1
tail -40 gpurun_out/campaign.txt
