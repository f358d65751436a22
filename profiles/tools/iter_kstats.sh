#!/bin/bash
# Per-kernel time table of one whole training iteration (profiles/tools/iteration_bench.py, all three modes) under
# rocprofv3 (run ON the GPU box through gpurun):  gpurun -- 'bash profiles/tools/iter_kstats.sh <tag>'
tag=${1:-it}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$root timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $root/profiles/tools/iteration_bench.py --iters 16 --only ${2:-C} ${3:-} ${4:-} ${5:-} ${6:-} > $root/gpurun_out/${tag}_iter.json 2> $out/err.log
cp $out/*/*kernel_stats.csv $root/gpurun_out/${tag}_kernel_stats.csv
python - "$root/gpurun_out/${tag}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:45]:
    name = r['Name'].split('(')[0].replace('void ', '')[:86]
    print(f"{name:86s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  per-iter {float(r['TotalDurationNs'])/1e3/22:8.1f}")
PY
cat $root/gpurun_out/${tag}_iter.json
