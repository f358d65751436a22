#!/bin/bash
# Everything profiles/<tag>_* is made of, in one gpurun call (run ON the GPU box):
#   gpurun --timeout 2400 -- "VEGS_COMMIT=$(git rev-parse --short HEAD) bash profiles/tools/collect_round.sh r03"
# kernel statistics, HBM traffic counters (FETCH_SIZE and WRITE_SIZE in separate passes), two SQ passes, the full bench
# line (all variants + CPU baselines) and the whole-iteration numbers.  Outputs land in gpurun_out/; copy what is to be
# judged into profiles/ (profiles/README.md).
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
bash profiles/tools/kstats.sh $tag
bash profiles/tools/pmc.sh $tag "FETCH_SIZE" "WRITE_SIZE" > /dev/null
bash profiles/tools/pmc.sh ${tag}_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" > /dev/null
# the traffic file bench.py reads, from the counters just collected (same sources: its hash guard passes); the copy in
# gpurun_out/ travels back (VEGS_COMMIT: the box has no .git)
python profiles/make_traffic.py gpurun_out/pmc_${tag}.json gpurun_out/pmc_${tag}_sq.json > gpurun_out/${tag}_pmc_traffic.json \
  && cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic.json
python bench.py --stages > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_err.log
PYTHONPATH=$root python profiles/tools/iteration_bench.py > gpurun_out/${tag}_iteration_c3.json 2>/dev/null
PYTHONPATH=$root python profiles/tools/iteration_bench.py --gaussians 5000000 --boxes 8 --iters 16 > gpurun_out/${tag}_iteration_c5.json 2>/dev/null
python profiles/tools/streams_probe.py > gpurun_out/${tag}_streams.txt 2>/dev/null
tail -2 gpurun_out/${tag}_bench_err.log
cat gpurun_out/${tag}_iteration_c3.json gpurun_out/${tag}_iteration_c5.json gpurun_out/${tag}_streams.txt
