"""The scenario of tests/test_gpu_dist_train.py, importable by the test (one process averaging the views) and by its rank
workers (one view each): a small street scene with box instances, stereo views, fixed targets, a schedule that densifies
inside the run.  Returns the trainer after `steps` iterations."""
import numpy as np
import torch

from vegs_amd import dist as vdist, harness, iteration, rasterizer, scenes

H, W = 188, 688


def make(world, rank, exchange, device, n_boxes=2, P=40000, seed=31, deterministic=True, schedule=True):
    sc, deg = scenes.scene_street(P=P, length=60.0, sh_degree=3, seed=seed)
    cams = [scenes.kitti_camera(3.0 * (i // 2), 0.3 if i % 2 == 0 else -0.3, W, H) for i in range(6)]
    rng = np.random.default_rng(17)
    gts = [torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32), device=device) for _ in cams]
    normals = [torch.tensor(rng.normal(size=(3, H, W)).astype(np.float32), device=device) for _ in cams]
    sch = None
    if schedule:           # densify at iteration 2 (thresholds that select a few thousand Gaussians), reset opacity at 3
        sch = iteration.Schedule(extent=20.0, densify_from_iter=1, densification_interval=2, opacity_reset_interval=3,
                                 densify_grad_threshold=2e-5)
    if deterministic:
        rasterizer.set_flags(rasterizer.get_flags() | rasterizer.FLAG_DETERMINISTIC)
    tr = iteration.Trainer(sc, device, n_boxes=n_boxes, fused=True, box_points=1500, factored_sh=(exchange != "dense"),
                           lrs=iteration.REFERENCE_LRS, optimise_boxes=True, world=world, rank=rank,
                           exchange=exchange if world > 1 or exchange == "dense" else "factored",
                           schedule=sch, seed=5)
    return tr, deg, cams, gts, normals


def run(tr, deg, cams, gts, normals, steps, n_views, world, rank, on_step=None):
    """`steps` iterations of `n_views` views each; a rank renders the views vdist.view_for_rank gives it."""
    bg = torch.zeros(3, device=tr.device)
    per_rank = n_views // world
    losses = []
    for it in range(steps):
        mine = [vdist.view_for_rank(it * per_rank + k, rank, world, len(cams)) for k in range(per_rank)] if world > 1 \
            else [(it * n_views + k) % len(cams) for k in range(n_views)]
        views = [dict(cam=cams[v], cam_t=harness.cam_tensors(cams[v], tr.device), gt=gts[v], normal=normals[v]) for v in mine]
        ls, _, _ = tr.step_views(views, deg, bg)
        losses.append([float(x) for x in ls])
        if on_step is not None:
            on_step(it, tr)
    return losses


def state_numpy(tr):
    return {k: v.detach().cpu().numpy() for k, v in tr.state_tensors().items()}
