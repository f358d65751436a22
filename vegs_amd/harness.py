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

from .so3 import matrix_to_quaternion, quaternion_to_matrix  # noqa: E402,F401  (re-exported: tests and tools use harness.*)


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
        # static=None: only the dynamic instances are in frame (the reference's render_dyn, gaussian_renderer/__init__.py:188-260)
        kw = prepare_rasterization(static) if static is not None else None
        for t, b2w in zip(boxes, box2worlds):
            one = prepare_rasterization(t, b2w)
            kw = one if kw is None else merge_kwargs(kw, one)
        if kw is None:
            raise ValueError("render_all: neither a static model nor a box instance to render")
    pkg = render(cam, kw, sh_degree, bg_color, scaling_modifier, cam_t=cam_t, sh_color_grad=sh_color_grad)
    pkg["op_inputs"] = kw
    return pkg
