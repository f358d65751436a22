# Round 6: multi-rank dry runs on ONE GPU (gloo transport: RCCL refuses several ranks per device).
#   8 ranks x 1 view at the full 2 M headline size (the driver's launch line), auto exchange;
#   4 ranks x 2 views per step: the factored + in-place multi-view exchange added to bench.py this round;
#   BASELINE C5's full step on 1 rank and on 2 ranks (direct exchange).
mkdir -p gpurun_out
export VEGS_DIST_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 8 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-variants 2>gpurun_out/dry8_err.log | tee gpurun_out/r06_dryrun_8ranks_auto.json | cut -c1-300
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29714 bench.py --gpus 4 --views-per-step 2 --exchange factored --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-variants 2>gpurun_out/dry4x2_err.log | tee gpurun_out/r06_dryrun_4ranks_2views_factored.json | cut -c1-300
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 2 --workload c5 --steps 4 --warmup 2 --repeats 1 --exchange direct 2>gpurun_out/c5_2_err.log | tee gpurun_out/r06_bench_c5_2ranks_direct.json | cut -c1-300
unset VEGS_DIST_BACKEND
python bench.py --workload c5 --steps 16 --warmup 4 2>gpurun_out/c5_err.log | tee gpurun_out/r06_bench_c5_n1.json | cut -c1-300
for f in dry8 dry4x2 c5_2 c5; do grep -v "Gloo\|amdgpu.ids\|socket.cpp\|pmc_traffic" gpurun_out/${f}_err.log | tail -3; done
