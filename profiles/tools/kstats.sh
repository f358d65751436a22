#!/bin/bash
# Per-kernel time table of bench.py under rocprofv3 (run ON the GPU box through gpurun):
#   gpurun -- 'bash profiles/tools/kstats.sh <tag> [bench args]'
# writes gpurun_out/<tag>_kernel_stats.csv, <tag>_bench.json and prints the vr:: kernels.
tag=${1:-k}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $root/bench.py --no-cpu-baseline --no-variants --steps 32 --warmup 8 "$@" > $root/gpurun_out/${tag}_bench.json 2> $out/err.log
cp $out/*/*kernel_stats.csv $root/gpurun_out/${tag}_kernel_stats.csv
python - "$root/gpurun_out/${tag}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    n = r['Name']
    if 'vr::' in n and 'count_' not in n and 'mark_visible' not in n:
        name = n.split('(')[0].replace('void ', '')
        per_view = float(r['TotalDurationNs']) / 1e3 / 64.0      # 64 forwards in the run (16 setup + 48 timed)
        print(f"{name:58s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
python -c "
import json,sys
d=json.loads(open('$root/gpurun_out/${tag}_bench.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'views/s', d['value'], 'stage_ms', d['roofline']['stage_ms'])"
