"""(round 6 debugging aid) DETERMINISTIC-mode gradients of bench camera 5 on the headline scene -> gpurun_out/detgrads_<tag>.npz"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from vegs_amd import _capi, rasterizer, scenes
from test_gpu_parity import _run_hip, _settings
_capi.load()
dev = torch.device("cuda:0")
sc, deg = scenes.scene_street(P=2_000_000, length=250.0, sh_degree=3, seed=2)
inputs = dict(means3D=sc["means3D"], shs=sc["shs"], colors_precomp=None, opacities=sc["opacities"], scales=sc["scales"],
              rotations=sc["rotations"], cov3D_precomp=None)
cam = scenes.kitti_camera(20.0, -0.3, 1376, 376)
H, W = 376, 1376
rng = np.random.default_rng(31)
gouts = [rng.normal(size=sh).astype(np.float32) * 1e-3 if m else None
         for sh, m in zip([(3, H, W), (1, H, W), (4, H, W), (3, H, W), (1, H, W)], (1, 0, 1, 1, 0))]
_, g, _ = _run_hip(_settings(cam, [0, 0, 0], deg, 1.0, dev), inputs, dev, gouts, flags=rasterizer.FLAG_DETERMINISTIC)
out = {}
for k in ("scales", "rotations", "means3D"):          # (sparse: rows with a non-zero gradient)
    a = g[k].reshape(g[k].shape[0], -1)
    nz = np.flatnonzero(np.abs(a).max(axis=1) > 0)
    out[k + "_rows"] = nz.astype(np.int32)
    out[k + "_vals"] = a[nz]
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"detgrads_{sys.argv[1]}.npz"), **out)
