#!/bin/bash
# after a change of the tile lists: parity + fuzz + full size + street sweep + the bench with its variants
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_c_harness.py tests/test_gpu_render_all.py -q -x 2>&1 | tail -3
VEGS_FUZZ_SEEDS=0:400 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -3
PYTHONPATH=.:tests timeout 1500 python profiles/tools/sweep_street.py 2>&1 | tail -8
python bench.py --stages > gpurun_out/cells_bench.json 2> gpurun_out/cells_bench.err; tail -c 600 gpurun_out/cells_bench.err
} > gpurun_out/quick_cells.txt 2>&1
tail -30 gpurun_out/quick_cells.txt
