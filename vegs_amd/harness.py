"""Device-agnostic counterpart of the reference's render() glue, for tests and bench.py.

Mirrors the contract of reference gaussian_renderer/__init__.py:20-119: a zeros
`screenspace_points` tensor with retained grad is passed as means2D, the 12 settings are built from
the camera, the op is called with the 8 keyword arguments, and the same result dict is returned.
The reference file itself is not shipped or imported; a user who has it runs it unmodified on top of
the `diff_gaussian_rasterization` package of this repo.
"""
import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def cam_tensors(cam, device):
    return dict(viewmatrix=torch.as_tensor(cam.world_view_transform, device=device),
                projmatrix=torch.as_tensor(cam.full_proj_transform, device=device),
                campos=torch.as_tensor(cam.camera_center, device=device))


def render(cam, tensors, sh_degree, bg_color, scaling_modifier=1.0, debug=False, colors_precomp=None,
           cov3D_precomp=None, cam_t=None, sh_color_grad=None):
    """tensors: dict with means3D, opacities, shs, scales, rotations (torch, on the GPU)."""
    means3D = tensors["means3D"]
    device = means3D.device
    screenspace_points = torch.zeros_like(means3D, dtype=means3D.dtype, requires_grad=True, device=device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    ct = cam_t if cam_t is not None else cam_tensors(cam, device)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=cam.tanfovx,
        tanfovy=cam.tanfovy, bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=ct["viewmatrix"],
        projmatrix=ct["projmatrix"], sh_degree=sh_degree, campos=ct["campos"], prefiltered=False, debug=debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    shs = None if colors_precomp is not None else tensors["shs"]
    scales = None if cov3D_precomp is not None else tensors["scales"]
    rotations = None if cov3D_precomp is not None else tensors["rotations"]
    rendered_image, depth_image, cov_quat, cov_scale, alpha, radii = rasterizer(
        means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
        opacities=tensors["opacities"], scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp,
        **({} if sh_color_grad is None else {"sh_color_grad": sh_color_grad}))
    return {"render": rendered_image, "render_depth": depth_image, "render_cov_quat": cov_quat,
            "render_cov_scale": cov_scale, "alpha": alpha, "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0, "radii": radii}


# ---------------------------------------------------------------------------------------------
# render_all-shaped driver: static Gaussians + dynamic box instances (reference
# gaussian_renderer/__init__.py:121-186 prepare_rasterization / merge_kwargs, :263-333 render_all).
# Every op input is a torch.cat of per-model tensors after a differentiable box2world transform,
# so gradients w.r.t. means3D, scales AND rotations must be right (they flow on into
# model/boxmodel.py:30-42).  Device-agnostic on purpose: the tests run the same graph on the CPU to
# carry the oracle's op-input gradients back to the parameters.

def quaternion_to_matrix(q):
    """(w,x,y,z) -> rotation matrix, normalising by |q|^2 (convention of utils/graphics_utils.py:204-248)."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def matrix_to_quaternion(m):
    """Rotation matrix -> (w,x,y,z): of the four algebraically equivalent candidates (each is the
    quaternion scaled by one of its own components) take the best conditioned one, i.e. the one whose
    defining component sqrt(1 +- m00 +- m11 +- m22)/2 is largest (utils/graphics_utils.py:140-201)."""
    m00, m01, m02 = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    m10, m11, m12 = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    m20, m21, m22 = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    sq = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1)
    q_abs = torch.sqrt(torch.clamp(sq, min=0.0) + (sq <= 0) * 1e-30) * (sq > 0)
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * torch.clamp(q_abs, min=0.1)[..., None])
    best = q_abs.argmax(-1)
    return torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4))).squeeze(-2)


def prepare_rasterization(t, box2world=None):
    """Op inputs of one Gaussian model (dict means3D, shs, opacities, scales, rotations); with a 4x4
    `box2world` the means, rotations and scales are carried into the world frame, differentiably."""
    means3D, scales, rotations = t["means3D"], t["scales"], t["rotations"]
    if box2world is not None:
        ones = torch.ones(means3D.shape[0], 1, dtype=means3D.dtype, device=means3D.device)
        hom = (box2world @ torch.cat((means3D, ones), 1).t()).t()
        means3D = hom[:, :3] / hom[:, 3:]
        lin = box2world[:3, :3]
        box_scale = torch.norm(lin, dim=0, keepdim=True)
        box_rot = lin / box_scale
        rotations = matrix_to_quaternion(box_rot[None] @ quaternion_to_matrix(rotations))
        scales = scales * box_scale
    return {"means3D": means3D, "shs": t["shs"], "opacities": t["opacities"], "scales": scales, "rotations": rotations}


def merge_kwargs(a, b):
    return {k: torch.cat((a[k], b[k]), 0).contiguous() for k in a}


def render_all(cam, static, boxes, box2worlds, sh_degree, bg_color, scaling_modifier=1.0, cam_t=None, fused=False,
               sh_color_grad=None, static_raw=None):
    """fused=False: the reference's op-by-op composition (device-agnostic; what the CPU checks run);
    fused=True: vegs_amd.instances.prepare_and_merge (one HIP launch for all instances, GPU only).
    static_raw (fused only): {"opacities", "scales", "rotations"} = the static model's RAW parameters, used instead of
    static's activated ones: the op activates the rows in front of the instances itself (VR_FLAG_RAW_PARAMS; the
    instances' rows behind the SH tail's boundary arrive activated and transformed) -- no activation launches over
    the static model.  Falls back to vegs_amd.instances.activate when the frame cannot use an SH tail."""
    if fused:
        from . import instances, rasterizer
        # (RAW only when the static model HAS rows: with an empty head the op would take every row -- the instances',
        # which arrive activated -- for raw parameters and activate them a second time)
        if static_raw is not None and static is not None and boxes and static["means3D"].shape[0] > 0:
            kw = instances.prepare_and_merge({**static, **static_raw}, boxes, box2worlds)
            if isinstance(kw["shs"], (tuple, list)) and len(kw["shs"]) == 3:       # the tail marks where the raw rows end
                with rasterizer.flags(rasterizer.get_flags() | rasterizer.FLAG_RAW_PARAMS):
                    pkg = render(cam, kw, sh_degree, bg_color, scaling_modifier, cam_t=cam_t, sh_color_grad=sh_color_grad)
                pkg["op_inputs"] = kw
                return pkg
            o, s_, r = instances.activate(static_raw["opacities"], static_raw["scales"], static_raw["rotations"])
            static = {**static, "opacities": o, "scales": s_, "rotations": r}
        kw = instances.prepare_and_merge(static, boxes, box2worlds)
    else:
        kw = prepare_rasterization(static)
        for t, b2w in zip(boxes, box2worlds):
            kw = merge_kwargs(kw, prepare_rasterization(t, b2w))
    pkg = render(cam, kw, sh_degree, bg_color, scaling_modifier, cam_t=cam_t, sh_color_grad=sh_color_grad)
    pkg["op_inputs"] = kw
    return pkg
