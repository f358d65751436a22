"""Street scenes at realistic list densities from random viewpoints (along / across / above the corridor, cameras inside
the geometry), discs at 1x and 3x size, against the oracle: radii, lists, ranges and images bit-exact, gradients per row
(tests/test_gpu_parity.py: _check_against_oracle), with the segment rounds left to the library and forced on.
    PYTHONPATH=.:tests python profiles/tools/sweep_street.py [n_cases] [P]"""
import sys

import numpy as np
import torch

import test_gpu_parity as tp
from vegs_amd import rasterizer, scenes

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
P = int(sys.argv[2]) if len(sys.argv) > 2 else 150_000
for case in range(N):
    rng = np.random.default_rng(500 + case)
    sc, deg = scenes.scene_street(P=P, length=float(rng.choice([40.0, 120.0])), sh_degree=int(rng.integers(0, 4)), seed=case)
    k = float(rng.choice([1.0, 3.0, 6.0]))
    sc["scales"] = (sc["scales"] * k).astype(np.float32)
    W, H = int(rng.choice([344, 688, 1376])), int(rng.choice([94, 188, 376]))
    eye = np.array([rng.uniform(0, 30), rng.uniform(-4, 4), rng.uniform(-1, 3)])
    target = eye + np.array([rng.normal(), rng.normal(), rng.normal() * 0.3])
    cam = scenes.lookat_camera(eye, target, [0, 0, 1.0], W, H, float(rng.uniform(40, 100)))
    inputs = dict(means3D=sc["means3D"], shs=sc["shs"], colors_precomp=None, opacities=sc["opacities"], scales=sc["scales"],
                  rotations=sc["rotations"], cov3D_precomp=None)
    hip = int(rng.choice([0, rasterizer.FLAG_ROUNDS_ON, rasterizer.FLAG_ROUNDS_OFF, rasterizer.FLAG_DETERMINISTIC]))
    _, _, _, st = tp._check_against_oracle(inputs, cam, rng.uniform(0, 1, 3).astype(np.float32), deg, 1.0, dev, seed=case,
                                           M=sc["shs"].shape[1], hip_flags=hip)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    print(f"case {case}: {W}x{H} deg {deg} discs x{k:g} flags {hip}: V={int((st['radii'] > 0).sum())} R={st['R']} "
          f"({st['R'] / 256 / T:.1f} segments per tile) ok", flush=True)
print("street sweep ok")
