"""profiles/pmc_traffic.json from summarize_pmc.py summaries: HBM bytes per launch of every vr:: kernel, and (second
argument, optional) its VALU wave-instructions per launch from the SQ pass.  Records WHAT it was collected on: the date, the
commit, and the sha256 of every kernel source -- bench.py reports `roofline.traffic` / `roofline.secondary.valu` only while
the source of the roofline kernel still has the recorded hash (a stale figure would otherwise be copied silently).
usage: python profiles/make_traffic.py profiles/r04_pmc_summary.json [profiles/r04_sq_counters.json] > profiles/pmc_traffic.json"""
import datetime
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vegs_amd", "csrc")


def source_hashes():
    out = {}
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            out[f] = hashlib.sha256(open(os.path.join(CSRC, f), "rb").read()).hexdigest()[:16]
    return out


def per_kernel(d, counter, scale=1.0, extra=None):
    acc = {}
    for k, v in d.items():
        if "vr::" not in k or counter not in v:
            continue
        name = k.split("vr::")[1].split("<")[0].split("(")[0]
        b = v[counter] * scale + (extra(v) if extra else 0.0)
        a = acc.setdefault(name, [0.0, 0])
        a[0] += b * v["dispatches"]
        a[1] += v["dispatches"]
    return {name: int(tot / max(n, 1)) for name, (tot, n) in sorted(acc.items())}


if __name__ == "__main__":
    d = json.load(open(sys.argv[1]))
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        commit = ""
    out = {"_note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                    "separate passes (" + sys.argv[1] + "); FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM (gfx950 counts "
                    "wide coalesced reads at half; WRITE_SIZE uncalibrated); kernels with several template instances are summed "
                    "weighted by dispatches",
           "_collected": {"date": datetime.date.today().isoformat(), "commit": commit or os.environ.get("VEGS_COMMIT", "")},
           "_sources": source_hashes()}
    out.update(per_kernel(d, "FETCH_SIZE", 2048.0, lambda v: v.get("WRITE_SIZE", 0.0) * 1024))
    if len(sys.argv) > 2:
        sq = json.load(open(sys.argv[2]))
        out["_valu_insts"] = per_kernel(sq, "SQ_INSTS_VALU")
        out["_valu_note"] = "SQ_INSTS_VALU per launch (wave-level VALU instructions; " + sys.argv[2] + ")"
    json.dump(out, sys.stdout, indent=1)
