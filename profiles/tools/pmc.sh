#!/bin/bash
# PMC counters of bench.py's kernels (run ON the GPU box through gpurun; one rocprofv3 pass per counter set, no tracing
# in the same run):   gpurun -- 'bash profiles/tools/pmc.sh <tag> "<counters>" ["<counters>" ...]'
# writes gpurun_out/pmc_<tag>.json = per-kernel mean counter values per dispatch (profiles/summarize_pmc.py).
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
dirs=""
i=0
for set in "$@"; do
  out=$root/gpurun_out/pmc_${tag}_$i
  mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --pmc $set --output-format csv -d $out -- python $root/bench.py --no-cpu-baseline --no-variants --steps 6 --warmup 2 > $out/bench.json 2> $out/err.log)
  dirs="$dirs $out"
  i=$((i+1))
done
python $root/profiles/summarize_pmc.py $dirs > $root/gpurun_out/pmc_${tag}.json
python - "$root/gpurun_out/pmc_${tag}.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in sorted(d):
    if 'k_seg' in k or 'preprocess' in k or 'radix' in k:
        print(k[:40], {c: round(v) for c, v in d[k].items()})
PY
