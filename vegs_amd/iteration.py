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


def _boxes_and_poses(boxes):
    """boxes: [(tensors, box2world), ...] or (list of tensors, ONE pose tensor [n,4,4])."""
    if isinstance(boxes, tuple) and len(boxes) == 2 and isinstance(boxes[1], torch.Tensor):
        return boxes
    return [b for b, _ in boxes], [w for _, w in boxes]


def render_model(p, boxes, cam, cam_t, deg, bg, fused, sh_sink=None):
    b_in, poses = _boxes_and_poses(boxes) if boxes else ([], [])
    boxes = b_in
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
        return harness.render_all(cam, t, b_in, poses, deg, bg, cam_t=cam_t, fused=True, sh_color_grad=sh_sink, static_raw=raw)
    # the reference's op-by-op composition (ATen activations, torch.cat of the SH tensors)
    t = {"means3D": p["xyz"], "opacities": torch.sigmoid(p["opacity"]), "scales": torch.exp(p["scaling"]),
         "rotations": F.normalize(p["rotation"]), "shs": torch.cat((p["f_dc"], p["f_rest"]), dim=1)}
    if not boxes:
        return harness.render(cam, t, deg, bg, cam_t=cam_t, sh_color_grad=sh_sink)
    return harness.render_all(cam, t, b_in, poses, deg, bg, cam_t=cam_t, fused=False, sh_color_grad=sh_sink)


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


class Schedule:
    """The densification / reset schedule of densification_and_optimization (train.py:283-320) with the reference's
    defaults (arguments/__init__.py:89-97); `extent` = scene.cameras_extent."""

    def __init__(self, extent, densify_from_iter=500, densify_until_iter=15_000, densify_until_iter_box=50_000,
                 densification_interval=100, opacity_reset_interval=3000, densify_grad_threshold=0.0002, percent_dense=0.01,
                 min_opacity=0.005, white_background=False):
        self.extent = extent
        self.densify_from_iter, self.densify_until_iter = densify_from_iter, densify_until_iter
        self.densify_until_iter_box, self.densification_interval = densify_until_iter_box, densification_interval
        self.opacity_reset_interval, self.densify_grad_threshold = opacity_reset_interval, densify_grad_threshold
        self.percent_dense, self.min_opacity, self.white_background = percent_dense, min_opacity, white_background

    def actions(self, iteration, box):
        """(collect statistics?, densify: (grad threshold, size threshold) | None, reset opacity?) at `iteration`."""
        if not iteration < (self.densify_until_iter_box if box else self.densify_until_iter):
            return False, None, False
        densify = None
        if iteration > self.densify_from_iter and iteration % self.densification_interval == 0:
            size = 20 if iteration > self.opacity_reset_interval else None
            thr = self.densify_grad_threshold
            if box:
                thr *= 0.5
                size = size * 0.5 if size is not None else None
            densify = (thr, size)
        reset = iteration % self.opacity_reset_interval == 0 or (self.white_background and iteration == self.densify_from_iter)
        return True, densify, reset


class InstanceModel:
    """One dynamic object's Gaussian model as the training loop holds it (scene.gaussian_box_models[instanceId]: a
    GaussianModel with its own optimizer, scene/gaussian_model.py:154-168): raw parameters + the six named groups."""

    def __init__(self, sc, device, fused, lrs=None):
        from . import optim
        self.p, groups = make_model(sc, device, lrs)
        self.opt = (optim.Adam if fused else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)

    @property
    def rows(self):
        return self.p["xyz"].shape[0]


def make_instance_models(n, device, fused, points=8196, seed=5, spacing=12.0, lrs=None, box_lr=0.005, lambda_reg=0.001,
                         box_model_cls=None):
    """n dynamic objects: (InstanceModel, BoxModel) pairs -- the Gaussians of make_boxes as raw parameters with optimizers,
    the pose as the reference's BoxModel (annotated box2world + the learnable deltas).  box_model_cls: the pose class
    (default vegs_amd.boxmodel.BoxModel, the HIP path; the op-by-op ATen variant of an iteration -- a comparison aid -- passes
    the checker's class, oracle.boxmodel_oracle.BoxModelOpByOp, which the product package does not import)."""
    from . import boxmodel
    if box_model_cls is None:
        if not fused:
            raise ValueError("the op-by-op iteration needs box_model_cls=oracle.boxmodel_oracle.BoxModelOpByOp "
                             "(test infrastructure; vegs_amd.boxmodel.BoxModel is the HIP path)")
        box_model_cls = boxmodel.BoxModel
    models, boxes = [], []
    brng = np.random.default_rng(seed)
    for i in range(n):
        b, _ = scenes.scene_random(P=points, sh_degree=3, seed=100 + i, extent=1.0, scale=0.05)
        B = np.eye(4, dtype=np.float32)
        ang = brng.uniform(0, 6.28)
        B[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32) * 1.5
        B[:3, 3] = [10.0 + spacing * i, brng.uniform(-3, 3), -0.8]
        models.append(InstanceModel(b, device, fused, lrs))
        boxes.append(box_model_cls(torch.tensor(B), lr=box_lr, lambda_reg=lambda_reg, device=device, fused=fused))
    return models, boxes


class Trainer:
    """State of one variant: parameters, optimizer(s), densification statistics -- on one GPU, or replicated on the
    `world` ranks of a view-sharded job (one process per GPU, SURVEY 8e / BASELINE C4, C5).

    step_views(views, ...) is ONE iteration over a batch of views with  loss = mean over the views of the per-view loss
    (one view per iteration reproduces the reference's loop exactly):
      * single process: `views` is the whole batch;
      * distributed (world > 1): `views` is THIS rank's share (normally one view: vegs_amd.dist.view_for_rank); the
        gradients are exchanged (vegs_amd.dist: factored + overlapped, or dense) so that every rank ends the iteration with
        the same parameters, optimizer state, statistics and -- every `schedule.densification_interval` iterations -- the
        same densified model (same statistics -> same plan; same generator seed -> same split draw,
        scene/gaussian_model.py:365-367).
    N ranks x 1 view == 1 process x N views: tests/test_gpu_dist_train.py."""

    def __init__(self, sc, device, n_boxes=0, fused=True, box_points=8196, factored_sh=False, lrs=None,
                 optimise_boxes=False, world=1, rank=0, group=None, exchange="factored", schedule=None, seed=0,
                 box_model_cls=None):
        """factored_sh (fused variant): the op returns the 3-float factor of the SH gradient and Adam consumes it directly
        (optim.adam_step_sh_factored) -- the static model's dense [P,16,3] gradient is never written.  With box instances
        in frame the factor covers the concatenated op inputs: the static model's rows feed Adam, the instances' few
        thousand rows are rebuilt densely (optim.sh_grad_from_factors on their world-space means).
        optimise_boxes: the instances are InstanceModels with their own optimizers and BoxModels with learnable pose
        corrections, all stepped every iteration (train.py:254-275); False keeps them as plain leaf tensors (gradients only).
        world / rank / group: the view-sharded job; exchange: "factored" (default: forces factored_sh), "dense", or
        "direct" (vegs_amd.xgmi: hand-written peer-to-peer exchange)."""
        from . import optim
        self.device, self.fused = device, fused
        self.world, self.rank, self.group, self.exchange = int(world), int(rank), group, exchange
        if exchange not in ("factored", "dense", "direct"):
            raise ValueError(f"exchange must be 'factored', 'dense' or 'direct' (got {exchange!r})")
        if self.world > 1 and exchange in ("factored", "direct"):
            # both schemes exchange the 3-float SH factor the FUSED operator returns; the op-by-op reference composition
            # (fused=False) has no such output -- it would arm an exchange that never receives a factor
            if not fused:
                raise ValueError(f"exchange={exchange!r} needs fused=True (the factored SH gradient is an output of the fused "
                                 "operator); use exchange='dense' with the op-by-op composition")
            factored_sh = True
        self.factored_sh = bool(factored_sh and fused)
        self.p, groups = make_model(sc, device, lrs)
        self.optimise_boxes = bool(optimise_boxes and n_boxes)
        self.instances, self.box_models, self.boxes = [], [], []
        if self.optimise_boxes:
            self.instances, self.box_models = make_instance_models(n_boxes, device, fused, box_points, lrs=lrs,
                                                                   box_model_cls=box_model_cls)
        elif n_boxes:
            self.boxes = make_boxes(n_boxes, device, box_points)
        self.opt = (optim.Adam if fused else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)
        P = self._rows()                                        # statistics over the op inputs
        self.accum = torch.zeros(P, 1, device=device)
        self.denom = torch.zeros(P, 1, device=device)
        self.max_radii = torch.zeros(P, device=device)
        self.win = ssim_window(device)
        self.schedule, self.iteration = schedule, 0
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(int(seed))
        self.xch = None
        if self.world > 1 and self.factored_sh and exchange == "factored":
            from . import dist as vdist
            self.xch = vdist.FactorExchange(self.world, group)
        self.direct = None
        if self.world > 1 and exchange == "direct":
            from . import xgmi
            self.direct = xgmi.DirectExchange(self.rank, self.world, device, group)
        self.last = {}
        self.sh_adam_events = None

    # ---- layout of the op inputs: static rows, then the instances' in order
    def _instance_rows(self):
        if self.optimise_boxes:
            return [m.rows for m in self.instances]
        return [b["means3D"].shape[0] for b, _ in self.boxes]

    def _rows(self):
        return self.p["xyz"].shape[0] + sum(self._instance_rows())

    def _box_inputs(self):
        """(boxes, box2worlds) for render_model: activated instance tensors + poses."""
        if not self.optimise_boxes:
            return [b for b, _ in self.boxes], [w for _, w in self.boxes]
        from . import boxmodel, instances
        poses = boxmodel.adjust_all(self.box_models)                  # [n,4,4]: one launch (fused) / op by op
        out = []
        for m in self.instances:
            p = m.p
            if self.fused:
                o, s_, r = instances.activate(p["opacity"], p["scaling"], p["rotation"])
            else:
                o, s_, r = torch.sigmoid(p["opacity"]), torch.exp(p["scaling"]), F.normalize(p["rotation"])
            out.append({"means3D": p["xyz"], "shs": torch.cat((p["f_dc"], p["f_rest"]), dim=1), "opacities": o, "scales": s_,
                        "rotations": r})
        return out, poses

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
            max_screen_size, percent_dense, noise=noise, generator=self.gen)
        self.p = new
        if rows_boxes:                        # the instances' rows of the statistics are kept behind the static model's
            accum, denom = torch.cat((accum, self.accum[P0:])), torch.cat((denom, self.denom[P0:]))
            max_radii = torch.cat((max_radii, self.max_radii[P0:]))
        self.accum, self.denom, self.max_radii = accum, denom, max_radii
        return new["xyz"].shape[0]

    def _scheduled(self):
        """densification_and_optimization's schedule part (train.py:283-316) for the static model and, with box=True
        thresholds, for every instance model (:254-268).  Returns the set of models whose parameters were replaced (the
        reference's optimizer.step() then finds no gradients on them: new nn.Parameter objects)."""
        from . import optim
        replaced = set()
        sch = self.schedule
        if sch is None or not self.fused:
            return replaced
        it = self.iteration
        P0 = self.p["xyz"].shape[0]
        _, dens, reset = sch.actions(it, box=False)
        if dens is not None:
            self.densify_and_prune(dens[0], sch.min_opacity, sch.extent, dens[1], sch.percent_dense)
            replaced.add("static")
        if reset:
            self.p["opacity"] = optim.reset_opacity(self.opt)
            replaced.add("static-opacity")
        if self.optimise_boxes:
            _, dens, reset = sch.actions(it, box=True)
            if dens is not None or reset:
                P0 = self.p["xyz"].shape[0]
                acc, den, mr, row = [self.accum[:P0]], [self.denom[:P0]], [self.max_radii[:P0]], P0
                for i, m in enumerate(self.instances):
                    n = m.rows
                    a, d, r = self.accum[row:row + n], self.denom[row:row + n], self.max_radii[row:row + n]
                    row += n
                    if dens is not None:
                        new, (a, d, r) = optim.densify_and_prune(m.opt, a.contiguous(), d.contiguous(), dens[0], sch.min_opacity,
                                                                  sch.extent, dens[1], sch.percent_dense, generator=self.gen)
                        m.p = new
                        replaced.add(("instance", i))
                    if reset:
                        m.p["opacity"] = optim.reset_opacity(m.opt)
                    acc.append(a); den.append(d); mr.append(r)
                self.accum, self.denom, self.max_radii = torch.cat(acc), torch.cat(den), torch.cat(mr)
        return replaced

    def forward_loss(self, cam, cam_t, deg, bg, gt, normal):
        sink = None
        if self.factored_sh and torch.is_grad_enabled():
            # (only its .grad is used: no fill)
            sink = torch.empty((self._rows(), 3), dtype=torch.float32, device=self.device, requires_grad=True)
        if self.optimise_boxes:
            b_in, poses = self._box_inputs()
            boxes = list(zip(b_in, poses)) if not isinstance(poses, torch.Tensor) else (b_in, poses)
        else:
            boxes = self.boxes
        pkg = render_model(self.p, boxes, cam, cam_t, deg, bg, self.fused, sh_sink=sink)
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
        """One iteration on one view (the reference's loop; in a distributed job: this rank's view)."""
        losses, pkgs, grads = self.step_views([dict(cam=cam, cam_t=cam_t, gt=gt, normal=normal)], deg, bg, keep_grads)
        return losses[0], pkgs[0], grads

    # ---- the per-view statistics (scene/gaussian_model.py:411-413, train.py:299), into `dst` = (accum, denom, max_radii)
    def _view_stats(self, pkg, dst):
        vis, radii, vsp = pkg["visibility_filter"], pkg["radii"], pkg["viewspace_points"]
        accum, denom, max_radii = dst
        if self.fused:
            from . import optim
            optim.add_densification_stats(vsp.grad, radii, accum, denom, max_radii)
        else:
            max_radii[vis] = torch.max(max_radii[vis], radii[vis].float())
            accum[vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)
            denom[vis] += 1

    def _small_params(self, with_sh):
        """Everything optimised besides the static model: instance parameters (+ box tensors / pose corrections)."""
        out = []
        for m in self.instances:
            out += [m.p[k] for k in ("xyz", "opacity", "scaling", "rotation")] + ([m.p["f_dc"], m.p["f_rest"]] if with_sh else [])
        for bm in self.box_models:
            out += [bm.delta_r, bm.delta_s, bm.delta_t]
        for b, w in self.boxes:
            out += [t for k, t in b.items() if with_sh or k != "shs"] + [w]
        return out

    def step_views(self, views, deg, bg, keep_grads=False):
        """One iteration over `views` (dicts cam, cam_t, gt, normal): this process's share of the iteration's view batch."""
        from . import boxmodel, optim, rasterizer
        from . import dist as vdist
        self.iteration += 1
        world, n_local = self.world, len(views)
        n_total = n_local * world
        rows, P0 = self._rows(), self.p["xyz"].shape[0]
        collect = True if self.schedule is None else self.schedule.actions(self.iteration, box=False)[0] or \
            (self.optimise_boxes and self.schedule.actions(self.iteration, box=True)[0])
        single = n_total == 1
        # statistics of THIS iteration's view batch: summed over the views first (and over the ranks), then added -- so that
        # N ranks x 1 view and 1 process x N views add the same numbers in the same order
        tmp = None if single else torch.zeros(3, rows, device=self.device)
        overlap = (self.xch is not None or self.direct is not None) and n_local == 1
        if self.direct is not None:      # (collective; a no-op while the model fits the window)
            self.direct.reserve(11 * P0 + 64, 3 * rows * n_local + 3 * n_local + 64)
        factors, campos, losses, pkgs = [], [], [], []
        for v in views:
            ct = v.get("cam_t") or harness.cam_tensors(v["cam"], self.device)
            loss, pkg = self.forward_loss(v["cam"], ct, deg, bg, v["gt"], v["normal"])
            # several local views: from the second one on the fused operator adds its rows straight into the leaves' .grad
            # (rasterizer.accumulate_grads: the same fp32 adds autograd would make, in the same order, without the dense
            # write + read-read-write per view; applies where the op's inputs ARE the leaf parameters -- no instances in frame)
            old_acc = rasterizer.accumulate_grads(self.fused and n_local > 1)
            try:
                if overlap:          # the factors start travelling between the backward's two halves
                    with (self.xch if self.xch is not None else self.direct).armed(ct["campos"]):
                        loss.backward()
                else:
                    loss.backward()
            finally:
                rasterizer.accumulate_grads(old_acc)
            with torch.no_grad():
                if collect:
                    self._view_stats(pkg, (self.accum, self.denom, self.max_radii) if single else
                                     (tmp[0].unsqueeze(1), tmp[1].unsqueeze(1), tmp[2]))
            if pkg.get("sh_sink") is not None:
                factors.append(pkg["sh_sink"].grad)
                campos.append(ct["campos"].reshape(3))
            losses.append(loss.detach())
            pkgs.append(pkg)

        # ---- the exchange: mean over ALL views of the iteration
        static_sh = [] if self.factored_sh else [self.p["f_dc"], self.p["f_rest"]]
        static_others = [self.p[k] for k in ("xyz", "opacity", "scaling", "rotation")]
        small = self._small_params(with_sh=not self.factored_sh)
        Fv = Cv = None
        with torch.no_grad():
            if self.factored_sh:
                Fv = factors[0][None] if n_local == 1 else torch.stack(factors)          # [n_local, rows, 3]
                Cv = torch.stack(campos).to(self.device, torch.float32)
            if world > 1:
                if self.direct is not None:
                    Fv, Cv = self.direct.finish(static_others) if overlap else self.direct.exchange(static_others, Fv, Cv, n_local)
                elif overlap:
                    Fv, Cv = self.xch.finish(static_others)
                elif self.factored_sh:
                    vdist.allreduce_grads(static_others, world, self.group)
                    Fv, Cv = vdist.all_gather_views(Fv, Cv, world, self.group)
                else:
                    vdist.allreduce_grads(static_others + static_sh, world, self.group)
                vdist.allreduce_grads(small, world, self.group, flat_bucket_bytes=1 << 40)
                if collect:
                    vdist._all_reduce(tmp[:2], torch.distributed.ReduceOp.SUM, self.group)
                    vdist._all_reduce(tmp[2], torch.distributed.ReduceOp.MAX, self.group)
            if n_local > 1:          # (the ranks' mean rides in the collective; the local views' is taken here)
                gs = [t.grad for t in static_others + static_sh + small if t.grad is not None]
                if gs:
                    torch._foreach_mul_(gs, 1.0 / n_local)
            if tmp is not None and collect:
                self.accum += tmp[0].unsqueeze(1)
                self.denom += tmp[1].unsqueeze(1)
                self.max_radii = torch.maximum(self.max_radii, tmp[2])
            grads = {k: v.grad.detach().clone() for k, v in self.p.items() if v.grad is not None} if keep_grads else None

            # the instances' SH gradients, densely, from their rows of the factors and their WORLD-space means (identical
            # for every view: the poses are replicated) -- already the mean over all views
            if self.factored_sh and rows > P0:
                means = pkgs[0]["op_inputs"]["means3D"].detach()
                M = (self.instances[0].p["f_dc"].shape[1] + self.instances[0].p["f_rest"].shape[1]) if self.optimise_boxes \
                    else self.boxes[0][0]["shs"].shape[1]
                g_all = optim.sh_grad_from_factors(means[P0:], Cv, Fv[:, P0:], deg, M, 1.0 / n_total, split=self.optimise_boxes)
                row = 0                   # (one launch for all instances; the gradients are row slices of its result)
                for i, n in enumerate(self._instance_rows()):
                    if self.optimise_boxes:
                        self.instances[i].p["f_dc"].grad = g_all[0][row:row + n]
                        self.instances[i].p["f_rest"].grad = g_all[1][row:row + n]
                    else:
                        self.boxes[i][0]["shs"].grad = g_all[row:row + n]
                    row += n

            # A timed-out wait of the direct exchange makes the NEXT exchange call raise (include/vegs_xgmi.h, FAILURE
            # CONTRACT); on iterations that densify -- the ranks must not plan a different model from diverged statistics --
            # the trainer synchronises and asks before it acts (one host sync per densification interval).
            if self.direct is not None and world > 1 and self.schedule is not None and \
                    self.schedule.actions(self.iteration, box=False)[1]:
                self.direct.check()
            # ---- densification / opacity reset on schedule, then the optimizers (train.py:283-320, :254-275)
            replaced = self._scheduled()
            if self.factored_sh and "static" not in replaced:
                ev = self.sh_adam_events          # (benchmarks: a HIP event pair around the SH Adam launch, when armed)
                if ev is not None:
                    ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                    ev[-1][0].record()
                optim.adam_step_sh_factored(self.opt, self.p["f_dc"], self.p["f_rest"], self.p["xyz"].detach(), Cv,
                                            Fv[:, :P0], deg, 1.0 / n_total)
                if ev is not None:
                    ev[-1][1].record()
            opts = [self.opt] + [m.opt for m in self.instances] + [bm.optimizer for bm in self.box_models]
            if self.fused:
                optim.step_many(opts)               # ONE multi-tensor launch: static model, instance models, pose corrections
            else:
                for o in opts:
                    o.step()
            for o in opts:
                o.zero_grad(set_to_none=True)
            if self.box_models:
                boxmodel.regularize_all(self.box_models)                    # train.py:274
        if not keep_grads:
            for b, w in self.boxes:                  # plain box tensors are not optimizer parameters
                w.grad = None
                for t in b.values():
                    t.grad = None
        self.last = {"factors": Fv, "campos": Cv, "replaced": replaced}
        return losses, pkgs, grads

    # ---- what must be identical on every rank after an iteration
    def state_tensors(self):
        out = {"static." + k: v for k, v in self.p.items()}
        for name, opt, p in [("static", self.opt, self.p)] + [(f"inst{i}", m.opt, m.p) for i, m in enumerate(self.instances)]:
            for k, v in p.items():
                out[f"{name}.{k}"] = v
                st = opt.state.get(v, {})
                for sk in ("exp_avg", "exp_avg_sq", "step"):
                    if sk in st:
                        out[f"{name}.{k}.{sk}"] = st[sk]
        for i, bm in enumerate(self.box_models):
            for k in ("delta_r", "delta_s", "delta_t"):
                t = getattr(bm, k)
                out[f"box{i}.{k}"] = t
                st = bm.optimizer.state.get(t, {})
                for sk in ("exp_avg", "exp_avg_sq", "step"):
                    if sk in st:
                        out[f"box{i}.{k}.{sk}"] = st[sk]
        out["accum"], out["denom"], out["max_radii"] = self.accum, self.denom, self.max_radii
        return out
