cd profiles/tools/ubench && ./lds_pixrec 2>&1 | tee ../../../gpurun_out/r05e_ubench.log; cd ../../..
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace -d /tmp/ub_pmc -o ub -- $GRAFT_REPO_ROOT/profiles/tools/ubench/lds_pixrec > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r05e_ubench_pmc.log
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/ub_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k, {c: int(x) for c, x in v.items()}, "conflict/active = %.3f" % (v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_ACTIVE_INST_LDS", 1), 1)))
PY
bash profiles/tools/ab/env_ab.sh VEGS_PRE_HALF 0 1 5 2>&1 | tee gpurun_out/r05e_ab.log
