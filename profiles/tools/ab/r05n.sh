#!/bin/bash
mkdir -p gpurun_out
for val in 1; do
  VEGS_SEG_CHAIN=$val bash profiles/tools/kstats.sh chain$val 2>&1 | grep -E "k_seg|ms_per_step"
done
