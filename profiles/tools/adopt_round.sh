#!/bin/bash
# copies what collect_round.sh left in gpurun_out/ to profiles/ under the round's names:  bash profiles/tools/adopt_round.sh r04
tag=${1:?tag}
cp gpurun_out/${tag}_kernel_stats.csv profiles/${tag}_kernel_stats.csv
cp gpurun_out/pmc_${tag}.json profiles/${tag}_pmc_summary.json
cp gpurun_out/pmc_${tag}_sq.json profiles/${tag}_sq_counters.json
cp gpurun_out/${tag}_bench_line.json profiles/${tag}_bench_line.json
cp gpurun_out/${tag}_streams.txt profiles/${tag}_streams.txt
cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic.json
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
a = json.loads(open(f"gpurun_out/{tag}_iteration_c3.json").read().strip().splitlines()[-1])
b = json.loads(open(f"gpurun_out/{tag}_iteration_c5.json").read().strip().splitlines()[-1])
json.dump({"c3": a, "c5_5M_8_boxes": b, "note": "profiles/tools/iteration_bench.py (single process; instance tensors as plain leaves).  The C5 step WITH optimised instance models and BoxModels: bench.py --workload c5."},
          open(f"profiles/{tag}_iteration.json", "w"), indent=1)
d = json.loads(open(f"profiles/{tag}_bench_line.json").read().strip().splitlines()[-1])
rf = d["roofline"]
print(d["value"], d["ms_per_step"], d["ms_per_step_regions"], "frac", rf["frac"], "ref", rf["frac_on_reference_lists"], "traffic", rf["traffic"], rf["traffic_collected"], "valu", rf["secondary"]["valu"])
for v in d["variants"]:
    print(" ", v["ms_per_view"], v["workload"][:90])
PY
