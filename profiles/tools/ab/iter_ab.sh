#!/bin/bash
# A/B of an environment switch on the fused C3 training iteration (raw parameters + split SH storage):
#   bash profiles/tools/ab/iter_ab.sh VEGS_PRE_HALF 0 1
var=$1; shift
for rep in 1 2 3; do
  for val in "$@"; do
    env $var=$val PYTHONPATH=. python profiles/tools/iteration_bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$val', {k: d[k] for k in d if k.endswith('_ms')})"
  done
done
