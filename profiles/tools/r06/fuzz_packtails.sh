#!/bin/bash
# Fresh fuzz seeds on the build with k_seg_bwd's row-packed tail chunks (GPU box, ~10 min): small random scenes are where the
# packed paths are the COMMON case (most regions hold fewer than 32 relevant entries).
mkdir -p gpurun_out
{
echo "== fuzz 18600:21000 (default: atomic backward)"; VEGS_FUZZ_SEEDS=18600:21000 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -3
echo "== fuzz 21000:22200, deterministic backward"; VEGS_FUZZ_SEEDS=21000:22200 VEGS_FUZZ_HIP_FLAGS=256 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -3
echo "== fuzz 22200:23400, full tile lists on both sides"; VEGS_FUZZ_SEEDS=22200:23400 VEGS_FUZZ_FLAGS=32768 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -3
echo "== fuzz 8000:9200 again (the seeds of the round's earlier findings: 9345 is in 9200:10400 below)"; VEGS_FUZZ_SEEDS=8000:9200 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -3
echo "== fuzz 9200:10400 + 11000:11200 + 17900:18500 again"; for r in 9200:10400 11000:11200 17900:18500; do VEGS_FUZZ_SEEDS=$r python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -2; done
} > gpurun_out/r06_fuzz_packtails.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r06_fuzz_packtails.txt | tail -30
