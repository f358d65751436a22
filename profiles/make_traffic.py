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


# kernels of one view's forward / backward (everything vr_forward / vr_backward launches on the headline path; the counting and
# export kernels of bench.py's bookkeeping are not part of a view)
FWD = ["k_preprocess", "k_scan_reduce", "k_scan_totals", "k_compact_apply", "k_digit_sum", "k_onesweep", "k_emit_scan", "k_emit_big",
       "k_split_count", "k_split_scan", "k_split_base", "k_split_scatter", "k_perm_check", "k_digit_hist", "k_tile_ranges",
       "k_seg_table", "k_seg_alpha", "k_seg_scan", "k_seg_merge", "k_seg_blend", "k_seg_combine"]
BWD = ["k_seg_u", "k_seg_suffix", "k_seg_bwd", "k_preprocess_bwd", "k_sh_factor"]


def whole_view(d, per_launch):
    """HBM bytes of ONE view: sum over its kernels of bytes per launch x launches per view (a kernel's dispatches over the
    dispatches of its half's anchor kernel: k_preprocess runs once per forward, k_preprocess_bwd once per backward)."""
    disp = {}
    for k, v in d.items():
        if "vr::" in k:
            name = k.split("vr::")[1].split("<")[0].split("(")[0]
            disp[name] = disp.get(name, 0) + v["dispatches"]
    rows, total = {}, 0.0
    for names, anchor in ((FWD, "k_preprocess"), (BWD, "k_preprocess_bwd")):
        for n in names:
            if n in per_launch and disp.get(anchor):
                per_view = disp[n] / disp[anchor]
                rows[n] = {"launches_per_view": round(per_view, 3), "bytes_per_view": int(per_launch[n] * per_view)}
                total += per_launch[n] * per_view
    return {"bytes": int(total), "kernels": rows}


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
    per_launch = per_kernel(d, "FETCH_SIZE", 2048.0, lambda v: v.get("WRITE_SIZE", 0.0) * 1024)
    out.update(per_launch)
    out["_whole_view"] = whole_view(d, per_launch)
    if len(sys.argv) > 2:
        sq = json.load(open(sys.argv[2]))
        out["_valu_insts"] = per_kernel(sq, "SQ_INSTS_VALU")
        out["_valu_note"] = "SQ_INSTS_VALU per launch (wave-level VALU instructions; " + sys.argv[2] + ")"
    json.dump(out, sys.stdout, indent=1)
