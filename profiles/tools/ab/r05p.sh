#!/bin/bash
# round 5, last campaign: everything at once
mkdir -p gpurun_out
{
echo "== stress 1200 (chain sweep 120 cases)"; VEGS_STRESS_ROUNDS=1200 timeout 1500 python -m pytest tests/test_gpu_stress.py -q -x 2>&1 | tail -3
echo "== dist / xgmi"; timeout 900 python -m pytest tests/test_gpu_xgmi.py tests/test_gpu_dist.py tests/test_gpu_dist_train.py -q -x -m gpu 2>&1 | tail -3
} > gpurun_out/r05_campaign2.txt 2>&1
bash profiles/tools/campaign.sh > /dev/null 2>&1
cat gpurun_out/r05_campaign2.txt; grep -E "^==|passed|failed|ok$" gpurun_out/campaign.txt
