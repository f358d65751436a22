"""GPU (-m gpu): the operator is usable for what VEGS uses it for -- gradient-based fitting.  A small
scene is optimised with Adam (the optimiser of scene/gaussian_model.py:154-172, same activations as
scene/gaussian_model.py:38-46,100-120) against images rendered from a ground-truth scene; the
photometric loss must fall substantially and stay finite, the densification statistics must be filled."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fitting_a_small_scene_reduces_the_loss():
    from vegs_amd import harness, scenes
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    gt, deg = scenes.scene_random(P=1500, sh_degree=1, seed=5, scale=0.05)
    cams = [scenes.lookat_camera([2.0 * np.cos(a), 2.0 * np.sin(a), 0.3], [0, 0, 0], [0, 0, 1.0], 96, 80, 45.0)
            for a in np.linspace(0, 2 * np.pi, 6, endpoint=False)]
    bg = torch.zeros(3, device=dev)
    gt_t = {k: torch.tensor(v, device=dev) for k, v in gt.items()}
    with torch.no_grad():
        targets = [harness.render(c, gt_t, deg, bg)["render"].clone() for c in cams]

    # trainable copy, perturbed; raw parameters + the reference's activations
    g = torch.Generator(device="cpu").manual_seed(0)
    xyz = (gt_t["means3D"] + 0.03 * torch.randn(1500, 3, generator=g).to(dev)).requires_grad_(True)
    shs = (gt_t["shs"] * 0.0).requires_grad_(True)                       # start grey
    scal = torch.log(gt_t["scales"] * 1.3).requires_grad_(True)           # exp activation
    rot = (gt_t["rotations"] + 0.1 * torch.randn(1500, 4, generator=g).to(dev)).requires_grad_(True)
    opa = torch.logit(gt_t["opacities"].clamp(0.05, 0.95)).requires_grad_(True)
    opt = torch.optim.Adam([{"params": [xyz], "lr": 1e-3}, {"params": [shs], "lr": 2e-2}, {"params": [scal], "lr": 5e-3},
                            {"params": [rot], "lr": 1e-3}, {"params": [opa], "lr": 2e-2}], eps=1e-15)
    grad_accum = torch.zeros(1500, 1, device=dev)
    denom = torch.zeros(1500, 1, device=dev)
    losses = []
    for it in range(120):
        v = it % len(cams)
        t = {"means3D": xyz, "shs": shs, "scales": torch.exp(scal),
             "rotations": torch.nn.functional.normalize(rot), "opacities": torch.sigmoid(opa)}
        pkg = harness.render(cams[v], t, deg, bg)
        loss = (pkg["render"] - targets[v]).abs().mean()
        loss.backward()
        vis = pkg["visibility_filter"]
        grad_accum[vis] += torch.norm(pkg["viewspace_points"].grad[vis, :2], dim=-1, keepdim=True)   # gaussian_model.py:411-413
        denom[vis] += 1
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(loss.item())
    first, last = np.mean(losses[:6]), np.mean(losses[-6:])
    assert np.isfinite(losses).all()
    assert last < 0.5 * first, (first, last)
    assert (denom > 0).sum() > 1000 and torch.isfinite(grad_accum).all() and grad_accum.sum() > 0


def test_iteration_with_factored_sh_equals_dense_iteration():
    """vegs_amd.iteration.Trainer: the fused step with the FACTORED SH gradient (op -> 3-float factor -> Adam) tracks
    the fused step with the dense gradient over a few iterations."""
    from vegs_amd import harness, iteration, scenes
    dev = torch.device("cuda:0")
    sc, deg = scenes.scene_street(P=50000, length=80.0, sh_degree=3, seed=21)
    cams = [scenes.kitti_camera(2.0 * i, 0.3, 688, 188) for i in range(3)]
    rng = np.random.default_rng(0)
    gt = torch.tensor(rng.uniform(0, 1, (3, 188, 688)).astype(np.float32), device=dev)
    normal = torch.tensor(rng.normal(size=(3, 188, 688)).astype(np.float32), device=dev)
    bg = torch.zeros(3, device=dev)
    out = []
    for factored in (False, True):
        tr = iteration.Trainer(sc, dev, fused=True, factored_sh=factored)
        losses = []
        for it in range(6):
            cam = cams[it % 3]
            losses.append(float(tr.step(cam, harness.cam_tensors(cam, dev), deg, bg, gt, normal)[0]))
        out.append((losses, {k: v.detach().cpu().numpy() for k, v in tr.p.items()}))
    (la, pa), (lb, pb) = out
    assert np.allclose(la, lb, rtol=2e-5)
    for k in pa:
        # Adam's first steps move by ~lr * sign(g): differences come from noise-level gradients changing sign
        tol = 1e-4 * np.abs(pa[k]).max()
        assert float((np.abs(pa[k] - pb[k]) > tol).mean()) < 5e-3, k


def test_factored_sh_with_box_instances_equals_the_dense_step():
    """Trainer(n_boxes > 0, factored_sh=True): the op's 3-float factor covers the concatenated op inputs (static model +
    instances, SH rows read through the SH tail); the static model's rows feed Adam directly, the instances' SH gradients
    are rebuilt densely from their rows of the factor and their world-space means.  Against the same step with the dense
    gradient: losses, parameters after three iterations, and every instance's SH / means / box2world gradient."""
    from helpers import assert_grad_close
    from vegs_amd import harness, iteration, scenes
    dev = torch.device("cuda:0")
    sc, deg = scenes.scene_street(P=60000, length=60.0, sh_degree=3, seed=33)
    cams = [scenes.kitti_camera(2.0 * i, 0.3, 688, 188) for i in range(3)]
    rng = np.random.default_rng(1)
    gt = torch.tensor(rng.uniform(0, 1, (3, 188, 688)).astype(np.float32), device=dev)
    normal = torch.tensor(rng.normal(size=(3, 188, 688)).astype(np.float32), device=dev)
    bg = torch.zeros(3, device=dev)
    out = []
    for factored in (False, True):
        tr = iteration.Trainer(sc, dev, n_boxes=3, fused=True, box_points=2000, factored_sh=factored)
        assert tr.factored_sh == factored
        losses, box_grads = [], None
        for it in range(3):
            cam = cams[it % 3]
            loss, pkg, _ = tr.step(cam, harness.cam_tensors(cam, dev), deg, bg, gt, normal, keep_grads=True)
            losses.append(float(loss))
            if it == 0:
                box_grads = [({k: v.grad.detach().cpu().numpy() for k, v in b.items() if v.grad is not None}, w.grad.cpu().numpy())
                             for b, w in tr.boxes]
            for b, w in tr.boxes:
                w.grad = None
                for t in b.values():
                    t.grad = None
        out.append((losses, {k: v.detach().cpu().numpy() for k, v in tr.p.items()}, box_grads))
    (la, pa, ga), (lb, pb, gb) = out
    assert np.allclose(la, lb, rtol=2e-5)
    for k in pa:
        tol = 1e-4 * np.abs(pa[k]).max()
        assert float((np.abs(pa[k] - pb[k]) > tol).mean()) < 5e-3, k
    for (da, wa), (db, wb) in zip(ga, gb):
        assert set(da) == set(db) and "shs" in db
        for k in da:
            assert_grad_close(f"box {k}: factored vs dense", db[k], da[k], rtol=1e-3, floor=2e-6)
        assert np.abs(wa - wb).max() <= 1e-3 * max(np.abs(wa).max(), 1e-12)
