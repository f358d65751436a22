"""Aggregate a rocprofv3 --pmc counter_collection CSV into per-kernel averages (json on stdout).

usage: python profiles/summarize_pmc.py <dir-with-*_counter_collection.csv> [...]
Each dispatch row carries Kernel_Name, Counter_Name, Counter_Value; the output maps kernel name ->
{counter: mean value per dispatch, "dispatches": n}.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in sys.argv[1:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    k = row.get("Kernel_Name", "?").split("(")[0]
                    a = acc[k][row.get("Counter_Name", "?")]
                    a[0] += float(row.get("Counter_Value", 0) or 0)
                    a[1] += 1
    out = {k: {c: v[0] / max(v[1], 1) for c, v in cs.items()} | {"dispatches": max(v[1] for v in cs.values())}
           for k, cs in acc.items()}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
