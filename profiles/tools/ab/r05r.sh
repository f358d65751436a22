#!/bin/bash
# round 5, end: 8 ranks sharing the one GPU through bench.py --gpus 8 exactly as the driver launches it (--exchange auto:
# factored first, the direct exchange in sacrificial children), and the C5 step on 1 and 2 ranks -- with the final build
mkdir -p gpurun_out
export VEGS_DIST_BACKEND=gloo
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29721 bench.py --gpus 8 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-variants ) 2>gpurun_out/dry8_auto_err.log | tee gpurun_out/r05_dryrun_8ranks_auto.json | cut -c1-600
grep -v "Gloo\|amdgpu.ids\|socket.cpp" gpurun_out/dry8_auto_err.log | tail -4
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29723 bench.py --gpus 2 --workload c5 --steps 4 --warmup 2 --repeats 1 --exchange direct ) 2>gpurun_out/c5_2_err.log | tee gpurun_out/r05_bench_c5_2ranks_direct.json | cut -c1-400
grep -v "Gloo\|amdgpu.ids\|socket.cpp" gpurun_out/c5_2_err.log | tail -3
unset VEGS_DIST_BACKEND
timeout 600 python bench.py --workload c5 --steps 16 --warmup 4 2>gpurun_out/c5_err.log | tee gpurun_out/r05_bench_c5_n1.json | cut -c1-400
