"""Shared helpers for the parity tests (test infrastructure; may use oracle/)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OUT_NAMES = ["color", "depth", "cov_quat", "cov_scale", "alpha"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"raster_{name}.npz"))
    return {k: z[k] for k in z.files}


def case_inputs(c):
    """The 7 tensor kwargs of the op (numpy), in the reference's naming."""
    pre = "in_colors_precomp" in c
    return dict(
        means3D=c["in_means3D"],
        shs=None if pre else c["in_shs"],
        colors_precomp=c["in_colors_precomp"] if pre else None,
        opacities=c["in_opacities"],
        scales=None if pre else c["in_scales"],
        rotations=None if pre else c["in_rotations"],
        cov3D_precomp=c["in_cov3D_precomp"] if pre else None,
    )


def oracle_cam_from_case(c):
    from oracle import oracle as orc
    P, W, H, deg = (int(v) for v in c["meta"])
    return orc.make_cam(H, W, c["tanfov"][0], c["tanfov"][1], c["bg"], float(c["scale_modifier"]),
                        c["viewmatrix"], c["projmatrix"], c["campos"], deg, 16)


def oracle_cam(cam, bg, sh_degree, scale_modifier=1.0, M=16):
    from oracle import oracle as orc
    return orc.make_cam(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, bg, scale_modifier,
                        cam.world_view_transform, cam.full_proj_transform, cam.camera_center, sh_degree, M)


def rel_err(a, b):
    """max |a-b| / max(|b|, tiny): gradient comparison metric (tensor-level relative error)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def normal_guidance_loss(cov_quat, cov_scale, normal, R_cam2world):
    """torch restatement of the consumer loss (reference loss/normal_guidance.py:3-22), used by the
    end-to-end gradient tests; pinned against ref_normal_guidance.npz."""
    import torch
    cs = cov_scale.permute(1, 2, 0).reshape(-1, 1, 3)
    q = cov_quat.permute(1, 2, 0).reshape(-1, 4)
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    rot = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                       two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                       two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1).reshape(-1, 3, 3)
    rs = rot.detach() * cs
    Rm = torch.as_tensor(R_cam2world, dtype=normal.dtype, device=normal.device)
    nw = (Rm @ normal.reshape(3, -1)).t()               # [npix,3] world normals
    nw = nw[:, :, None].expand(-1, 3, 3)
    return 0.8 * (rot * nw).sum(dim=-2).abs().mean() + 0.2 * (rs * nw).sum(dim=-2).abs().mean()
