#!/usr/bin/env python
"""dump_fork_goldens.py -- produce REFERENCE outputs of the real CUDA rasterizer for this repo's committed input cases.

Why: VEGS pins `emjay73/diff_gaussian_rasterization_with_depth` (reference .gitmodules:7-9) as an un-vendored
submodule, and it is CUDA-only: it can be neither read nor run where this repo is built, so the oracle's rasterizer core
is pinned only to a restatement of the published algorithm ("parity unpinned").  This script is the missing half: run it
ONCE on any NVIDIA box where the reference environment is installed and copy its output back.

It is standalone on purpose -- numpy + torch + the real `diff_gaussian_rasterization` package, nothing from this
repository is imported.  For every `raster_case_*.npz` in --cases (the files under tests/golden/ of this repo: inputs
and camera of a small scene) it
  1. builds GaussianRasterizationSettings exactly as reference gaussian_renderer/__init__.py:38-51 does,
  2. calls the rasterizer exactly as reference gaussian_renderer/__init__.py:86-94 does
     (-> color, depth, cov_quat, cov_scale, alpha, radii),
  3. back-propagates the stored upstream gradients: once all five `gout_*` arrays (what a generic caller could feed),
     once only the three outputs VEGS' losses reach (colour, cov_quat, cov_scale: train.py:152-168),
and writes  <out>/fork_<case>.npz  with  out_*, radii, grad_* (five upstream gradients) and grad3_* (three).

    python dump_fork_goldens.py --cases /path/to/repo/tests/golden --out /path/to/repo/tests/golden/fork

Then, in this repo:  python -m pytest tests/test_fork_goldens.py -q        (CPU: the oracle against the fork)
                     python -m pytest tests/test_fork_goldens.py -q -m gpu (MI355X: the HIP kernels against the fork)
The test reports which combination of the fork switches (include/vegs_rast.h VrFlags bits 0-3) reproduces the fork within
1e-4 abs on the forward images -- that combination is the VEGS_RAST_FLAGS value to train with.
"""
import argparse
import glob
import os

import numpy as np
import torch

OUTS = ["color", "depth", "cov_quat", "cov_scale", "alpha"]
INS = ["means3D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp"]


def run_case(path, device):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    z = np.load(path)
    P, W, H, deg = (int(v) for v in z["meta"])
    dev = torch.device(device)

    def t(name, grad=False):
        key = "in_" + name
        if key not in z.files:
            return None
        return torch.tensor(z[key], dtype=torch.float32, device=dev, requires_grad=grad)

    result = {}
    for tag, which in (("grad", OUTS), ("grad3", ["color", "cov_quat", "cov_scale"])):
        ins = {n: t(n, grad=True) for n in INS}
        means2D = torch.zeros((P, 3), dtype=torch.float32, device=dev, requires_grad=True) + 0
        means2D.retain_grad()
        settings = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=float(z["tanfov"][0]), tanfovy=float(z["tanfov"][1]),
            bg=torch.tensor(z["bg"], dtype=torch.float32, device=dev), scale_modifier=float(z["scale_modifier"]),
            viewmatrix=torch.tensor(z["viewmatrix"], dtype=torch.float32, device=dev),
            projmatrix=torch.tensor(z["projmatrix"], dtype=torch.float32, device=dev), sh_degree=deg,
            campos=torch.tensor(z["campos"], dtype=torch.float32, device=dev), prefiltered=False, debug=False)
        res = GaussianRasterizer(raster_settings=settings)(
            means3D=ins["means3D"], means2D=means2D, shs=ins["shs"], colors_precomp=ins["colors_precomp"],
            opacities=ins["opacities"], scales=ins["scales"], rotations=ins["rotations"],
            cov3D_precomp=ins["cov3D_precomp"])
        color, depth, cov_quat, cov_scale, alpha, radii = res
        outs = dict(color=color, depth=depth, cov_quat=cov_quat, cov_scale=cov_scale, alpha=alpha)
        if tag == "grad":
            for n in OUTS:
                result["out_" + n] = outs[n].detach().cpu().numpy()
            result["radii"] = radii.detach().cpu().numpy().astype(np.int32)
        tensors, grads = [], []
        for n in which:
            g_np = z["gout_" + n]
            if "crop" in z.files:      # a case that stores a CROP of its images: the upstream gradients are zero outside it
                y0, y1, x0, x1 = (int(v) for v in z["crop"])
                full = np.zeros((g_np.shape[0], H, W), np.float32)
                full[:, y0:y1, x0:x1] = g_np
                g_np = full
            g = torch.tensor(g_np, dtype=torch.float32, device=dev).reshape(outs[n].shape)
            if outs[n].requires_grad:
                tensors.append(outs[n])
                grads.append(g)
        torch.autograd.backward(tensors, grads)
        for n in INS:
            if ins[n] is not None and ins[n].grad is not None:
                result[f"{tag}_{n}"] = ins[n].grad.detach().cpu().numpy()
        if means2D.grad is not None:
            result[f"{tag}_means2D"] = means2D.grad.detach().cpu().numpy()
    return result


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--cases", required=True, help="directory holding raster_case_*.npz (tests/golden of the repo)")
    ap.add_argument("--out", required=True, help="output directory (tests/golden/fork of the repo)")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    files = sorted(glob.glob(os.path.join(args.cases, "raster_case_*.npz")))
    if not files:
        raise SystemExit(f"no raster_case_*.npz under {args.cases}")
    import diff_gaussian_rasterization
    origin = getattr(diff_gaussian_rasterization, "__file__", "?")
    for f in files:
        name = os.path.basename(f)[len("raster_"):-len(".npz")]
        res = run_case(f, args.device)
        res["origin"] = np.array(origin)
        np.savez_compressed(os.path.join(args.out, f"fork_{name}.npz"), **res)
        print(f"{name}: radii>0 {int((res['radii'] > 0).sum())}, color range [{res['out_color'].min():.4f}, "
              f"{res['out_color'].max():.4f}] -> fork_{name}.npz")


if __name__ == "__main__":
    main()
