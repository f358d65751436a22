"""One seed of tests/test_gpu_fuzz.py with diagnostics: which gradient rows leave the per-row allowance in the atomic and in the
deterministic backward (twice each), their conic conditioning, and the measured summation sensitivity of every tensor.
    PYTHONPATH=.:tests python profiles/tools/fuzz_case.py <seed>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_gpu_parity as tp
import test_gpu_fuzz as tf
from helpers import grad_mismatch, ill_conditioned, conic_conditioning, oracle_cam, summation_sensitivity
from oracle import oracle as orc
from vegs_amd import scenes, rasterizer
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 72
dev = torch.device('cuda', 0)
# replicate test_random_configuration's setup
rng = np.random.default_rng(9000 + seed)
P = int(rng.choice([1, 2, 63, 64, 65, 300, 1500, 4000]))
W, H = int(rng.integers(17, 300)), int(rng.integers(9, 200))
deg = int(rng.integers(0, 4))
M = int(rng.choice([m for m in (1, 4, 9, 16) if m >= (deg + 1) ** 2]))
extent = float(rng.choice([0.05, 0.5, 3.0])); scale = float(rng.choice([1e-4, 0.01, 0.05, 0.5]))
sc, _ = scenes.scene_random(P=P, sh_degree=3, seed=seed, extent=extent, scale=scale)
sc["shs"] = np.ascontiguousarray(sc["shs"][:, :M])
if seed % 3 == 0:
    sc["opacities"] = rng.choice([0.0, 1.0, 0.3, 0.9999], size=(P, 1)).astype(np.float32)
    sc["rotations"] = (sc["rotations"] * rng.uniform(0.2, 3.0, (P, 1))).astype(np.float32)
eye = rng.normal(size=3); eye = eye / np.linalg.norm(eye) * float(rng.choice([0.2, 1.0, 2.5]))
cam = scenes.lookat_camera(eye, rng.normal(size=3) * 0.1, [0, 0, 1.0], W, H, float(rng.uniform(30, 110)))
mod = float(rng.choice([1.0, 0.5, 1.7])); bg = rng.uniform(0, 1, 3).astype(np.float32)
pre_col, pre_cov = bool(rng.integers(0, 2)) and seed % 2 == 1, bool(rng.integers(0, 2)) and seed % 4 == 3
inputs = dict(means3D=sc["means3D"], shs=None if pre_col else sc["shs"],
              colors_precomp=rng.uniform(0, 1, (P, 3)).astype(np.float32) if pre_col else None, opacities=sc["opacities"],
              scales=None if pre_cov else sc["scales"], rotations=None if pre_cov else sc["rotations"],
              cov3D_precomp=tf._cov6(sc["scales"], sc["rotations"], mod) if pre_cov else None)
gmask = tuple(int(v) for v in rng.integers(0, 2, 5))
if not any(gmask): gmask = (1, 0, 0, 0, 0)
print("P", P, "WxH", W, H, "deg", deg, "M", M, "extent", extent, "scale", scale, "mod", mod, "gmask", gmask, "eye", np.linalg.norm(eye))
oc = oracle_cam(cam, bg, deg, mod, M)
o_out, st = orc.forward(oc, **inputs)
g = np.random.default_rng(seed)
shapes = [(3, H, W), (1, H, W), (4, H, W), (3, H, W), (1, H, W)]
gouts = [g.normal(size=s).astype(np.float32) if m else None for s, m in zip(shapes, gmask)]
og = orc.backward(oc, st, *gouts, abs_sums=True)
ill, explain = ill_conditioned(st)
for flags, name in ((0, "atomic"), (rasterizer.FLAG_DETERMINISTIC, "deterministic")):
    for rep in range(2):
        _, hg, _ = tp._run_hip(tp._settings(cam, bg, deg, mod, dev), inputs, dev, gouts, flags=flags)
        for k in ("scales", "rotations", "means3D", "opacities"):
            if hg.get(k) is None or og.get(k) is None: continue
            bad, ratio = grad_mismatch(hg[k], og[k], 1e-3, 1e-6)
            if len(bad):
                print(name, rep, k, [(int(i), round(float(ratio[i]), 2), explain(int(i)), "radius", int(st["radii"][i])) for i in bad[:5]])
mv = summation_sensitivity(oc, st, og, rtol=1e-3, floor=1e-6)
for k, v in mv.items():
    top = np.argsort(-v)[:3]
    print("sensitivity", k, [(int(i), round(float(v[i]), 2)) for i in top])
