"""One fuzz seed, one gradient row, three witnesses: the HIP kernels (atomic and deterministic), the C oracle (its double sums)
and the float64 autograd restatement (oracle/torch_ref.py).  Who is off when HIP and the C oracle disagree on a row?
    PYTHONPATH=.:tests python profiles/tools/r06/fuzz_row.py <seed> <tensor> <row> [hip flags]"""
import sys
import numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_gpu_parity as tp
import test_gpu_fuzz as tf
from helpers import oracle_cam
from oracle import oracle as orc, torch_ref
from vegs_amd import scenes, rasterizer
seed, name, row = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
hip_flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device('cuda', 0)
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
print("P", P, "WxH", W, H, "deg", deg, "M", M, "extent", extent, "scale", scale, "mod", mod, "gmask", gmask, "pre_col", pre_col, "pre_cov", pre_cov)
oc = oracle_cam(cam, bg, deg, mod, M)
o_out, st = orc.forward(oc, **inputs)
g = np.random.default_rng(seed)
shapes = [(3, H, W), (1, H, W), (4, H, W), (3, H, W), (1, H, W)]
gouts = [g.normal(size=s).astype(np.float32) if m else None for s, m in zip(shapes, gmask)]
og = orc.backward(oc, st, *gouts, abs_sums=True)
print("C oracle      ", name, row, og[name][row], "radius", st["radii"][row], "opacity", inputs["opacities"][row], "abs-sum of the per-fragment terms (opacity)", og["_per_gaussian"]["abs"][row, 5])
for flags, tag in ((hip_flags, "hip"), (hip_flags | rasterizer.FLAG_DETERMINISTIC, "hip deterministic")):
    _, hg, _ = tp._run_hip(tp._settings(cam, bg, deg, mod, dev), inputs, dev, gouts, flags=flags)
    print(f"{tag:14s}", name, row, hg[name][row])
# float64 autograd
T = {k: (None if v is None else torch.tensor(np.asarray(v, np.float64), requires_grad=True)) for k, v in inputs.items()}
m2d = torch.zeros(P, 3, dtype=torch.float64, requires_grad=True)
res = torch_ref.rasterize(T["means3D"], T["shs"], T["colors_precomp"], T["opacities"], T["scales"], T["rotations"], T["cov3D_precomp"],
                          H=H, W=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.tensor(bg, dtype=torch.float64), scale_modifier=mod,
                          viewmatrix=torch.tensor(cam.world_view_transform), projmatrix=torch.tensor(cam.full_proj_transform),
                          campos=torch.tensor(cam.camera_center), sh_degree=deg, means2D=m2d)
loss = sum((r * torch.tensor(gg, dtype=torch.float64)).sum() for r, gg in zip(res[:5], gouts) if gg is not None)
loss.backward()
ref = m2d.grad if name == "means2D" else T[name].grad
print("float64 ref   ", name, row, ref[row].numpy())
for n, a, b in zip(("color", "depth", "cov_quat", "cov_scale", "alpha"), res[:5], (o_out[k] for k in ("color", "depth", "cov_quat", "cov_scale", "alpha"))):
    print("   forward", n, "C oracle vs float64 max |diff|", float(np.abs(a.detach().numpy() - b).max()))
