#!/bin/bash
# Round 6 campaign on the final build (GPU box, ~15 min): the full suite, FRESH fuzz seeds (5200 ...), the sweeps, and the soak --
# incl. 4,000 headline forwards + backwards under the depth sort's post-mortem (VEGS_STRESS_ROUNDS=1000).
mkdir -p gpurun_out
{
echo "== full GPU suite"; python -m pytest tests -q -m gpu 2>&1 | tail -3
echo "== fuzz 5200:6400"; VEGS_FUZZ_SEEDS=5200:6400 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -3
echo "== fuzz 6400:6800, full tile lists on both sides";  VEGS_FUZZ_SEEDS=6400:6800 VEGS_FUZZ_FLAGS=32768 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -3
echo "== fuzz 6800:7200, rounds on + scan binning + deterministic"; VEGS_FUZZ_SEEDS=6800:7200 VEGS_FUZZ_HIP_FLAGS=2816 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -3
echo "== fuzz 7200:7600, VERIFY_BINNING on the HIP side"; VEGS_FUZZ_SEEDS=7200:7600 VEGS_FUZZ_HIP_FLAGS=16384 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -3
echo "== densify 600:1200"; VEGS_FUZZ_SEEDS=600:1200 python -m pytest tests/test_gpu_densify.py -q -k random_settings 2>&1 | tail -3
echo "== street sweep"; PYTHONPATH=.:tests timeout 1500 python profiles/tools/sweep_street.py 2>&1 | tail -4
echo "== neighbours"; PYTHONPATH=. timeout 600 python profiles/tools/sweep_neighbours.py 500 2>&1 | tail -5
echo "== soak: VEGS_STRESS_ROUNDS=1000 (1000 forwards under a competing stream, 333 rounds of two views in flight, 100 chain-mode sweep cases, 4000 forwards + backwards under the post-mortem)"
VEGS_STRESS_ROUNDS=1000 python -m pytest tests/test_gpu_stress.py -q 2>&1 | tail -3
} > gpurun_out/r06_campaign.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r06_campaign.txt | tail -40
