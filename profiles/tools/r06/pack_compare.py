"""Row-packed tail chunks of k_seg_bwd (round 6): the DETERMINISTIC-mode gradients of two library builds on the same views.
  python profiles/tools/r06/pack_compare.py dump <tag>      (with VEGS_LIB naming the build; writes gpurun_out/packcmp_<tag>.npz)
  python profiles/tools/r06/pack_compare.py cmp <tagA> <tagB>
The two builds add the same per-fragment terms of an (entry, region) slot in a different order (one chain of 32 pixel pairs
against 4 x 8 or 2 x 16): per Gaussian row the difference must stay at fp32 summation level of the row's largest component."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = ("means2D", "opacities", "shs", "means3D", "scales", "rotations")
if sys.argv[1] == "dump":
    import torch
    from vegs_amd import _capi, rasterizer, scenes
    from test_gpu_parity import _run_hip, _settings
    _capi.load()
    dev = torch.device("cuda:0")
    sc, deg = scenes.scene_street(P=2_000_000, length=250.0, sh_degree=3, seed=2)
    inputs = dict(means3D=sc["means3D"], shs=sc["shs"], colors_precomp=None, opacities=sc["opacities"], scales=sc["scales"],
                  rotations=sc["rotations"], cov3D_precomp=None)
    out = {}
    for ci, (s, y) in enumerate([(5 // 2, 0.3 if 5 % 2 == 0 else -0.3), (10 // 2, 0.3)]):
        cam = scenes.kitti_camera(10.0 * s, y, 1376, 376)
        H, W = 376, 1376
        rng = np.random.default_rng(31)
        gouts = [rng.normal(size=sh).astype(np.float32) * 1e-3 if m else None
                 for sh, m in zip([(3, H, W), (1, H, W), (4, H, W), (3, H, W), (1, H, W)], (1, 0, 1, 1, 0))]
        for mode, fl in (("det", rasterizer.FLAG_DETERMINISTIC), ("atomic", 0)):
            _, g, _ = _run_hip(_settings(cam, [0, 0, 0], deg, 1.0, dev), inputs, dev, gouts, flags=fl)
            for k in NAMES:
                out[f"v{ci}_{mode}_{k}"] = g[k].reshape(g[k].shape[0], -1)
    np.savez(os.path.join(ROOT, "gpurun_out", f"packcmp_{sys.argv[2]}.npz"), **out)
    print("dumped", sys.argv[2], os.environ.get("VEGS_LIB", "default"))
else:
    a = np.load(os.path.join(ROOT, "gpurun_out", f"packcmp_{sys.argv[2]}.npz"))
    b = np.load(os.path.join(ROOT, "gpurun_out", f"packcmp_{sys.argv[3]}.npz"))
    for key in a.files:
        x, y = a[key].astype(np.float64), b[key].astype(np.float64)
        scale = np.maximum(np.abs(y).max(axis=1), 1e-30)
        rel = np.abs(x - y).max(axis=1) / scale
        live = np.abs(y).max(axis=1) > 1e-7 * np.abs(y).max()
        r = rel[live]
        print(f"{key:24s} rows {live.sum():8d}  identical {np.mean(r == 0):.3f}  median {np.median(r):.2e}  p99 {np.quantile(r, 0.99):.2e}  "
              f"p99.99 {np.quantile(r, 0.9999):.2e}  max {r.max():.2e}  rows > 1e-3: {(r > 1e-3).sum()}  > 1e-2: {(r > 1e-2).sum()}")
