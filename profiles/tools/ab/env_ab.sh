#!/bin/bash
# A/B of an environment switch on the headline workload's stage times (run ON the GPU box):
#   bash profiles/tools/ab/env_ab.sh VEGS_PRE_HALF 0 1
# prints, per setting, ms per view and the per-stage milliseconds of bench.py --stages (two runs each, interleaved)
var=$1; shift
for rep in 1 2; do
  for val in "$@"; do
    env $var=$val python bench.py --stages --no-variants --no-cpu-baseline --steps 20 --warmup 8 2> /tmp/ab_err.log > /tmp/ab_out.json
    python - "$var=$val" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/ab_out.json") if l.startswith("{")][-1])
st = d["roofline"]["stage_ms"]
print(sys.argv[1], "ms/view", d["ms_per_step"], "regions", d["ms_per_step_regions"], {k: round(v, 4) for k, v in st.items()})
PY
  done
done
