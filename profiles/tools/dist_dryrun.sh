# Multi-rank dry runs on ONE GPU (profiles/r04_dryrun_*.json): 8 processes at the full 2 M-Gaussian headline size through
# bench.py --gpus 8 as the driver launches it, with the gloo transport (RCCL refuses several ranks per device) and with the
# direct hipIpc exchange; and BASELINE C5's full training step on 1 rank and on 2 ranks.
mkdir -p gpurun_out
export VEGS_DIST_BACKEND=gloo
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 8 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-variants ) 2>gpurun_out/dry8_gloo_err.log | tee gpurun_out/r04_dryrun_8ranks_gloo.json | cut -c1-700
grep -v "Gloo\|amdgpu.ids\|socket.cpp" gpurun_out/dry8_gloo_err.log | tail -5
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29712 bench.py --gpus 8 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-variants --exchange direct ) 2>gpurun_out/dry8_direct_err.log | tee gpurun_out/r04_dryrun_8ranks_direct.json | cut -c1-700
grep -v "Gloo\|amdgpu.ids\|socket.cpp" gpurun_out/dry8_direct_err.log | tail -5
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 2 --workload c5 --steps 4 --warmup 2 --repeats 1 --exchange direct ) 2>gpurun_out/c5_2_err.log | tee gpurun_out/r04_bench_c5_2ranks_direct.json | cut -c1-400
grep -v "Gloo\|amdgpu.ids\|socket.cpp" gpurun_out/c5_2_err.log | tail -5
unset VEGS_DIST_BACKEND
python bench.py --workload c5 --steps 16 --warmup 4 2>gpurun_out/c5_err.log | tee gpurun_out/r04_bench_c5_n1.json | cut -c1-400
