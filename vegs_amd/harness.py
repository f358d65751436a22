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
           cov3D_precomp=None, cam_t=None):
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
        opacities=tensors["opacities"], scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "render_depth": depth_image, "render_cov_quat": cov_quat,
            "render_cov_scale": cov_scale, "alpha": alpha, "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0, "radii": radii}
