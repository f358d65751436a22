"""Offline statistics of the render backward's (entry, pixel) contribution matrix on the headline scene (CPU oracle).
usage: python profiles/tools/pairstats.py [P]"""
import ctypes as C, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle as orc
from vegs_amd import scenes
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/libpairstats.so"
subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", so, os.path.join(here, "pairstats.c"), "-lm"])
L = C.CDLL(so)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
sc, deg = scenes.scene_street(P=P, length=250.0, sh_degree=3, seed=2)
cam = scenes.kitti_camera(20.0, -0.3, 1376, 376)
oc = orc.make_cam(376, 1376, cam.tanfovx, cam.tanfovy, [0, 0, 0], 1.0, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, deg, 16)
out, st = orc.forward(oc, sc["means3D"], sc["shs"], None, sc["opacities"], sc["scales"], sc["rotations"], None)
o = np.zeros(8); hm = np.zeros(65); hn = np.zeros(257)
p = lambda a: a.ctypes.data_as(C.c_void_p)
L.pair_stats(376, 1376, p(st["ranges"]), p(st["point_list"]), p(st["xy"]), p(st["conic_op"]), p(st["n_contrib"]), p(o), p(hm), p(hn))
print("R", st["R"], "units(nonempty)", o[0], "units(needed)", o[5], "pairs", o[1], "sum max_c", o[2], "dense batches", o[3], "sum nrel_exact", o[4], "half-wave trips", o[7])
print("avg pairs/unit", o[1] / o[0], "avg max_c", o[2] / o[0], "avg nrel", o[4] / o[0])
cum = np.cumsum(hm) / hm.sum()
print("max_c percentiles:", {q: int(np.searchsorted(cum, q)) for q in (0.25, 0.5, 0.75, 0.9, 0.99)})
cumn = np.cumsum(hn) / hn.sum()
print("nrel percentiles:", {q: int(np.searchsorted(cumn, q)) for q in (0.25, 0.5, 0.75, 0.9, 0.99)})
print("trips weighted hist (max_c bucket: share of sum max_c):", {k: round(float((hm[a:b] * np.arange(a, b)).sum() / o[2]), 3) for k, (a, b) in {"1-4": (1, 5), "5-8": (5, 9), "9-16": (9, 17), "17-32": (17, 33), "33-64": (33, 65)}.items()})

import sys as _s
if len(_s.argv) > 2 and _s.argv[2] == "chunks":
    q = None
q = np.zeros(12)
L.quad_stats(376, 1376, p(st["ranges"]), p(st["point_list"]), p(st["xy"]), p(st["conic_op"]), p(st["n_contrib"]), p(q))
print("bwd pixel-pair trips now", q[0], "rows-of-16-per-quadrant trips", q[1], "ratio", q[0] / q[1], "| sum n", q[2], "sum_q n_q", q[3], "quadrants per entry", q[3] / q[2])
print("fwd needed segs: sum n", q[4], "sum max_q n_q", q[5], "ratio", q[4] / q[5])
print("fwd all segs: sum n", q[6], "sum max_q n_q", q[7], "ratio", q[6] / q[7], "quadrants per entry", q[8] / q[6], "units", q[9])
print("fwd FIRST segs: sum n", q[10], "sum max_q n_q", q[11], "ratio", q[10] / max(q[11], 1))

ch = np.zeros(13)
L.chunk_stats(376, 1376, p(st["ranges"]), p(st["point_list"]), p(st["xy"]), p(st["conic_op"]), p(st["n_contrib"]), p(ch))
for sname, k in (("64 entries x 2 pixels (now)", 0), ("32 entries x 4 pixels (pairs pp, pp+16)", 1), ("32 entries x 4 pixels (4 x 1 runs)", 2)):
    cheap, rej, acc = ch[3 * k:3 * k + 3]
    print(f"k_seg_bwd {sname}: chunks {ch[9 + k]:.0f} trips cheap {cheap:.0f} reject {rej:.0f} accepted {acc:.0f}  cost(4/25/116) {(4 * cheap + 25 * rej + 116 * acc) / 1e6:.1f} M")
print("units", ch[12])

ts = np.zeros(27)
L.tail_stats(376, 1376, p(st["ranges"]), p(st["point_list"]), p(st["xy"]), p(st["conic_op"]), p(st["n_contrib"]), p(ts))
print(f"tails: units {ts[14]:.0f} chunks {ts[9]:.0f}; last chunk <=16: {ts[10]:.0f}, 17..32: {ts[11]:.0f}, 33..48: {ts[12]:.0f}, 49..64: {ts[13]:.0f}")
for name, k in (("now", 0), ("tails <= 16 row-packed (8 trips x 4 pairs)", 1), ("... and tails <= 32 half-packed (16 trips x 2 pairs)", 2)):
    cheap, rej, acc = ts[3 * k:3 * k + 3]
    print(f"k_seg_bwd {name}: trips cheap {cheap:.0f} reject {rej:.0f} accepted {acc:.0f}  cost(4/25/116) {(4 * cheap + 25 * rej + 116 * acc) / 1e6:.1f} M")
print(f"accepted trips now inside tails <= 16: {ts[15]:.0f}, inside tails 17..32: {ts[16]:.0f}")
cheap, rej, acc = ts[17:20]
print(f"k_seg_bwd ... and entries behind every pixel's last contributor dropped before chunking: held entries {ts[20]:.0f} -> {ts[21]:.0f}, chunks {ts[9]:.0f} -> {ts[22]:.0f}, trips cheap {cheap:.0f} reject {rej:.0f} accepted {acc:.0f}  cost(4/25/116) {(4 * cheap + 25 * rej + 116 * acc) / 1e6:.1f} M")
cheap, rej, acc = ts[23:26]
print(f"k_seg_bwd ... and tails of 33 .. 48 entries as 32 half-packed + a row-packed rest: chunks {ts[26]:.0f}, trips cheap {cheap:.0f} reject {rej:.0f} accepted {acc:.0f}  cost(4/25/116) {(4 * cheap + 25 * rej + 116 * acc) / 1e6:.1f} M")
