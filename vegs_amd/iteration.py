"""Device-side counterpart of ONE VEGS training iteration, for tests and benchmarks (not a trainer).

What the reference does per iteration, in its own words (none of that Python is shipped or imported here):
  train.py:143-150   render_all(viewpoint_cam, gaussians, boxmodels, ...)       -> harness.render / render_all
  train.py:152-168   Ll1, ssim, loss_normal_guidance                            -> photometric + normal guidance
  train.py:196       loss.backward()
  train.py:299-301   max_radii2D / add_densification_stats                      -> densification statistics
  train.py:319-320   optimizer.step(); zero_grad(set_to_none=True)              -> Adam over the six named groups
with the model's activations of scene/gaussian_model.py:100-120 (sigmoid opacity, exp scaling, normalised rotation,
cat(features_dc, features_rest)) and its optimizer groups (:159-168).

Two variants of the same step are built from the same pieces so that tests can compare them:
  fused=False  the reference's composition: op-by-op prepare_rasterization/merge_kwargs, ATen loss code as
               utils/loss_utils.py:18-79 and loss/normal_guidance.py:3-22 write it, torch.optim.Adam
  fused=True   rows N1/N2/N4 of DESIGN.md section 10: vegs_amd.instances, vegs_amd.losses, vegs_amd.optim
Both go through the same rasterizer (there is only one).
"""
import types

import numpy as np
import torch
import torch.nn.functional as F

from . import harness, scenes

LRS = {"xyz": 1.6e-6, "f_dc": 2.5e-4, "f_rest": 2.5e-4 / 20, "opacity": 5e-3, "scaling": 5e-4, "rotation": 1e-4}
LAMBDA_DSSIM = 0.2          # arguments/__init__.py (lambda_dssim)
LAMBDA_NORMAL = 1e-3


# the reference's own learning rates (arguments/__init__.py:80-88; position_lr_init x spatial_lr_scale, the camera extent)
REFERENCE_LRS = {"xyz": 1.6e-5 * 10.0, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 5e-2, "scaling": 1e-3, "rotation": 1e-3}


def make_model(sc, device, lrs=None):
    """Raw parameters + optimizer groups as scene/gaussian_model.py:145-168 builds them from a scene dict."""
    lrs = lrs or LRS
    t = {k: torch.as_tensor(v, device=device) for k, v in sc.items()}
    p = {"xyz": t["means3D"].clone(), "f_dc": t["shs"][:, :1].contiguous(), "f_rest": t["shs"][:, 1:].contiguous(),
         "opacity": torch.logit(t["opacities"].clamp(1e-4, 1 - 1e-4)), "scaling": torch.log(t["scales"]),
         "rotation": t["rotations"].clone()}
    p = {k: torch.nn.Parameter(v.requires_grad_(True)) for k, v in p.items()}
    return p, [{"params": [p[k]], "lr": lrs[k], "name": k} for k in p]


def make_boxes(n, device, points=8196, seed=5, spacing=12.0):
    """n dynamic box instances of `points` Gaussians (scene/gaussian_model.py:462) with a random similarity box2world."""
    out = []
    brng = np.random.default_rng(seed)
    for i in range(n):
        b, _ = scenes.scene_random(P=points, sh_degree=3, seed=100 + i, extent=1.0, scale=0.05)
        B = np.eye(4, dtype=np.float32)
        ang = brng.uniform(0, 6.28)
        B[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32) * 1.5
        B[:3, 3] = [10.0 + spacing * i, brng.uniform(-3, 3), -0.8]
        out.append(({k: torch.tensor(v, device=device, requires_grad=True) for k, v in b.items()},
                    torch.tensor(B, device=device, requires_grad=True)))
    return out


def ssim_window(device):
    g1 = torch.tensor([np.exp(-(i - 5) ** 2 / 4.5) for i in range(11)], dtype=torch.float32)
    g1 = g1 / g1.sum()
    return (g1[:, None] @ g1[None, :]).expand(3, 1, 11, 11).contiguous().to(device)


def render_model(p, boxes, cam, cam_t, deg, bg, fused, sh_sink=None):
    if fused and not boxes:
        # the op takes the RAW parameters and activates them in its preprocess kernel (VR_FLAG_RAW_PARAMS): no activation
        # launches, gradients straight to _opacity / _scaling / _rotation
        from . import rasterizer
        t = {"means3D": p["xyz"], "opacities": p["opacity"], "scales": p["scaling"], "rotations": p["rotation"],
             "shs": (p["f_dc"], p["f_rest"])}
        with rasterizer.flags(rasterizer.get_flags() | rasterizer.FLAG_RAW_PARAMS):
            return harness.render(cam, t, deg, bg, cam_t=cam_t, sh_color_grad=sh_sink)
    if fused:
        # with instances in frame: the static model's raw parameters go into the concatenated inputs as they are and the op
        # activates the rows in front of the instances (harness.render_all: static_raw)
        t = {"means3D": p["xyz"], "shs": (p["f_dc"], p["f_rest"])}
        raw = {"opacities": p["opacity"], "scales": p["scaling"], "rotations": p["rotation"]}
        return harness.render_all(cam, t, [b for b, _ in boxes], [w for _, w in boxes], deg, bg, cam_t=cam_t, fused=True,
                                  sh_color_grad=sh_sink, static_raw=raw)
    # the reference's op-by-op composition (ATen activations, torch.cat of the SH tensors)
    t = {"means3D": p["xyz"], "opacities": torch.sigmoid(p["opacity"]), "scales": torch.exp(p["scaling"]),
         "rotations": F.normalize(p["rotation"]), "shs": torch.cat((p["f_dc"], p["f_rest"]), dim=1)}
    if not boxes:
        return harness.render(cam, t, deg, bg, cam_t=cam_t, sh_color_grad=sh_sink)
    return harness.render_all(cam, t, [b for b, _ in boxes], [w for _, w in boxes], deg, bg, cam_t=cam_t, fused=False,
                              sh_color_grad=sh_sink)


def aten_loss(pkg, gt, normal, win, R_c2w):
    """The reference's loss block op by op (utils/loss_utils.py:18-79 l1 + ssim, loss/normal_guidance.py:3-22)."""
    x, q, s = pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]
    l1 = (x - gt).abs().mean()
    mu1, mu2 = F.conv2d(x, win, padding=5, groups=3), F.conv2d(gt, win, padding=5, groups=3)
    s1 = F.conv2d(x * x, win, padding=5, groups=3) - mu1 * mu1
    s2 = F.conv2d(gt * gt, win, padding=5, groups=3) - mu2 * mu2
    s12 = F.conv2d(x * gt, win, padding=5, groups=3) - mu1 * mu2
    ss = (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
    Rm = harness.quaternion_to_matrix(q.permute(1, 2, 0).reshape(-1, 4))
    Rw = torch.as_tensor(R_c2w, dtype=torch.float32, device=x.device)
    nw = (Rw @ normal.reshape(3, -1)).t()[:, :, None].repeat(1, 1, 3)
    ng = 0.8 * (Rm * nw).sum(-2).abs().mean() + 0.2 * (Rm.detach() * s.permute(1, 2, 0).reshape(-1, 1, 3) * nw).sum(-2).abs().mean()
    return (1.0 - LAMBDA_DSSIM) * l1 + LAMBDA_DSSIM * (1 - ss) + LAMBDA_NORMAL * ng


def fused_loss(pkg, gt, normal, R_c2w):
    """train.py:162-168 as one autograd node (losses.training_loss), the uncovered-pixel guard inside it"""
    from . import losses
    cam = types.SimpleNamespace(original_normal=normal, R=R_c2w)
    loss, _ = losses.training_loss(pkg["render"], gt, cam, pkg["render_cov_quat_raw"], pkg["render_cov_scale"], LAMBDA_DSSIM,
                                   LAMBDA_NORMAL, guard_empty=True)
    return loss


def op_inputs(p):
    """The rasterizer's inputs (activated, concatenated) of raw parameters p, as detached numpy-free tensors."""
    with torch.no_grad():
        return {"means3D": p["xyz"].detach().clone(), "shs": torch.cat((p["f_dc"], p["f_rest"]), dim=1).contiguous(),
                "opacities": torch.sigmoid(p["opacity"]).detach(), "scales": torch.exp(p["scaling"]).detach(),
                "rotations": F.normalize(p["rotation"]).detach()}


class Trainer:
    """State of one variant: parameters, optimizer, densification statistics."""

    def __init__(self, sc, device, n_boxes=0, fused=True, box_points=8196, factored_sh=False, lrs=None):
        """factored_sh (fused variant): the op returns the 3-float factor of the SH gradient and Adam consumes it directly
        (optim.adam_step_sh_factored) -- the static model's dense [P,16,3] gradient is never written.  With box instances
        in frame the factor covers the concatenated op inputs: the static model's rows feed Adam, the instances' few
        thousand rows are rebuilt densely (optim.sh_grad_from_factors on their world-space means)."""
        from . import optim
        self.device, self.fused = device, fused
        self.factored_sh = bool(factored_sh and fused)
        self.p, groups = make_model(sc, device, lrs)
        self.boxes = make_boxes(n_boxes, device, box_points) if n_boxes else []
        self.opt = (optim.Adam if fused else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)
        P = self.p["xyz"].shape[0] + sum(b["means3D"].shape[0] for b, _ in self.boxes)   # statistics over the op inputs
        self.accum = torch.zeros(P, 1, device=device)
        self.denom = torch.zeros(P, 1, device=device)
        self.max_radii = torch.zeros(P, device=device)
        self.win = ssim_window(device)

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, percent_dense=0.01, noise=None):
        """train.py:303-312 for the static model (fused variant): optim.densify_and_prune on the optimizer's tensors, the
        statistics restarted at the new length.  Returns the number of Gaussians after the step."""
        from . import optim
        if not self.fused:
            raise NotImplementedError("the ATen variant exists for timing comparisons of the iteration only")
        P0 = self.p["xyz"].shape[0]
        rows_boxes = self.accum.shape[0] - P0
        new, (accum, denom, max_radii) = optim.densify_and_prune(
            self.opt, self.accum[:P0].contiguous(), self.denom[:P0].contiguous(), max_grad, min_opacity, extent,
            max_screen_size, percent_dense, noise=noise)
        self.p = new
        if rows_boxes:                        # the instances' rows of the statistics follow the static model's
            pad = torch.zeros(rows_boxes, 1, device=self.device)
            accum, denom = torch.cat((accum, pad)), torch.cat((denom, pad))
            max_radii = torch.cat((max_radii, pad[:, 0]))
        self.accum, self.denom, self.max_radii = accum, denom, max_radii
        return new["xyz"].shape[0]

    def forward_loss(self, cam, cam_t, deg, bg, gt, normal):
        sink = None
        if self.factored_sh and torch.is_grad_enabled():
            rows = self.p["xyz"].shape[0] + sum(b["means3D"].shape[0] for b, _ in self.boxes)
            # (only its .grad is used: no fill)
            sink = torch.empty((rows, 3), dtype=torch.float32, device=self.device, requires_grad=True)
        pkg = render_model(self.p, self.boxes, cam, cam_t, deg, bg, self.fused, sh_sink=sink)
        pkg["sh_sink"] = sink
        # NaN guard for pixels no Gaussian covers (A-5: exact zeros; the reference's 2/|q|^2 is NaN there) -- same in
        # both variants
        q = pkg["render_cov_quat"]
        pkg["render_cov_quat_raw"] = q
        if self.fused:
            loss = fused_loss(pkg, gt, normal, cam.R)        # (the guard is part of the fused loss block)
        else:
            pkg["render_cov_quat"] = torch.where((q.detach() * q.detach()).sum(0, keepdim=True) > 0, q, torch.ones_like(q))
            loss = aten_loss(pkg, gt, normal, self.win, cam.R)
        return loss, pkg

    def step(self, cam, cam_t, deg, bg, gt, normal, keep_grads=False):
        from . import optim
        loss, pkg = self.forward_loss(cam, cam_t, deg, bg, gt, normal)
        loss.backward()
        with torch.no_grad():
            vis, radii, vsp = pkg["visibility_filter"], pkg["radii"], pkg["viewspace_points"]
            if self.fused:
                optim.add_densification_stats(vsp.grad, radii, self.accum, self.denom, self.max_radii)
            else:
                self.max_radii[vis] = torch.max(self.max_radii[vis], radii[vis].float())             # train.py:299
                self.accum[vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)                # gaussian_model.py:411-413
                self.denom[vis] += 1
        grads = {k: v.grad.detach().clone() for k, v in self.p.items() if v.grad is not None} if keep_grads else None
        if pkg.get("sh_sink") is not None:
            ct = cam_t if cam_t is not None else harness.cam_tensors(cam, self.device)
            campos = ct["campos"].reshape(1, 3)
            factors = pkg["sh_sink"].grad
            P0 = self.p["xyz"].shape[0]
            optim.adam_step_sh_factored(self.opt, self.p["f_dc"], self.p["f_rest"], self.p["xyz"].detach(), campos,
                                        factors[:P0][None], deg, 1.0)
            if self.boxes:      # the instances' SH gradients, densely, from their rows of the factor and their WORLD-space means
                means = pkg["op_inputs"]["means3D"].detach()
                g_all = optim.sh_grad_from_factors(means[P0:], campos, factors[P0:][None], deg, self.boxes[0][0]["shs"].shape[1])
                row = 0
                for b, _ in self.boxes:       # (one launch for all instances; the gradients are row slices of its result)
                    n = b["means3D"].shape[0]
                    b["shs"].grad = g_all[row:row + n]
                    row += n
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        if not keep_grads:
            for b, w in self.boxes:                  # box tensors are not optimizer parameters here
                w.grad = None
                for t in b.values():
                    t.grad = None
        return loss.detach(), pkg, grads
