#!/bin/bash
# Fresh fuzz seeds + soak on the round's FINAL build (GPU box, ~10 min).
mkdir -p gpurun_out
{
echo "== fuzz 23400:25800 (atomic backward)"; VEGS_FUZZ_SEEDS=23400:25800 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -2
echo "== fuzz 25800:27000, deterministic backward"; VEGS_FUZZ_SEEDS=25800:27000 VEGS_FUZZ_HIP_FLAGS=256 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -2
echo "== fuzz 27000:27800, full tile lists + rounds on + scan binning + deterministic"; VEGS_FUZZ_SEEDS=27000:27800 VEGS_FUZZ_FLAGS=32768 VEGS_FUZZ_HIP_FLAGS=2816 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -2
echo "== soak: VEGS_STRESS_ROUNDS=400"; VEGS_STRESS_ROUNDS=400 python -m pytest tests/test_gpu_stress.py -q 2>&1 | tail -2
echo "== accumulate tests x 10"; for i in 1 2 3 4 5 6 7 8 9 10; do python -m pytest tests/test_gpu_accumulate.py -q 2>&1 | tail -1; done
} > gpurun_out/r06_fuzz_final.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r06_fuzz_final.txt | grep -v "^\.\.*" | tail -30
