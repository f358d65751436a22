"""GPU (-m gpu): the fused per-pixel losses (vegs_amd/losses.py -> csrc/losses.hip)
against (1) outputs of the reference's own functions (tests/golden/ref_photometric.npz,
ref_normal_guidance.npz: values + autograd gradients) and (2) the float64 oracle at the full KITTI-360 frame
size.  fp32 kernels vs float64 / torch-fp32 conv: relative tolerances written at each assert."""
import os
import types

import numpy as np
import pytest
import torch

from helpers import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_photometric_losses_match_reference_outputs(tag):
    from vegs_amd import losses
    z = np.load(os.path.join(GOLDEN, "ref_photometric.npz"))
    img = torch.tensor(z[f"img_{tag}"], device=DEV, requires_grad=True)
    gt = torch.tensor(z[f"gt_{tag}"], device=DEV)
    # exactly the three lines of train.py:162-164
    Ll1 = losses.l1_loss(img, gt)
    lambda_dssim = 0.2
    loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - losses.ssim(img, gt))
    assert abs(Ll1.item() - z[f"l1_{tag}"]) < 1e-6
    assert abs(loss.item() - z[f"loss_{tag}"]) < 1e-6
    loss.backward()
    assert rel_err(img.grad.cpu().numpy(), z[f"grad_loss_{tag}"]) < 2e-5
    # separately
    for fn, key in ((losses.l1_loss, "l1"), (losses.ssim, "ssim")):
        x = torch.tensor(z[f"img_{tag}"], device=DEV, requires_grad=True)
        v = fn(x, gt)
        assert abs(v.item() - z[f"{key}_{tag}"]) < 2e-6
        v.backward()
        assert rel_err(x.grad.cpu().numpy(), z[f"grad_{key}_{tag}"]) < 2e-5, key
    # fused convenience call
    x = torch.tensor(z[f"img_{tag}"], device=DEV, requires_grad=True)
    l2, l1v = losses.photometric_loss(x, gt, 0.2)
    l2.backward()
    assert abs(l2.item() - z[f"loss_{tag}"]) < 1e-6 and abs(l1v.item() - z[f"l1_{tag}"]) < 1e-6
    assert rel_err(x.grad.cpu().numpy(), z[f"grad_loss_{tag}"]) < 2e-5


def test_photometric_full_frame_against_oracle_and_no_grad():
    from oracle import loss_oracle as lo
    from vegs_amd import losses
    rng = np.random.default_rng(3)
    H, W = 376, 1376
    x = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    y = np.clip(x + rng.normal(0, 0.1, x.shape), 0, 1).astype(np.float32)
    l1, ss, grad = lo.photometric(x, y, 0.8, -0.2)
    xt = torch.tensor(x, device=DEV, requires_grad=True)
    yt = torch.tensor(y, device=DEV)
    loss, _ = losses.photometric_loss(xt, yt, 0.2)
    loss.backward()
    assert abs(loss.item() - (0.8 * l1 + 0.2 * (1 - ss))) < 1e-6
    assert rel_err(xt.grad.cpu().numpy(), grad) < 2e-5
    with torch.no_grad():                                   # evaluation path (train.py:557): no derivative maps kept
        assert abs(losses.l1_loss(xt, yt).item() - l1) < 1e-6 and abs(losses.ssim(xt, yt).item() - ss) < 1e-6
    # deterministic: the loss sums do not use floating-point atomics
    a = losses.photometric_loss(xt, yt, 0.2)[0].item()
    assert a == losses.photometric_loss(xt, yt, 0.2)[0].item()


def _cam(normal, R):
    return types.SimpleNamespace(original_normal=normal, R=R)


def test_normal_guidance_matches_reference_outputs():
    from vegs_amd import losses
    z = np.load(os.path.join(GOLDEN, "ref_normal_guidance.npz"))
    cq = torch.tensor(z["cov_quat"], device=DEV, requires_grad=True)
    cs = torch.tensor(z["cov_scale"], device=DEV, requires_grad=True)
    cam = _cam(torch.tensor(z["normal"], device=DEV), z["R"])
    loss = losses.loss_normal_guidance(cam, cq, cs)
    assert abs(loss.item() - z["loss"]) < 1e-6
    (1e-3 * loss).backward()                                 # train.py:168: loss += lambda_dnormal * Lng
    assert rel_err(cq.grad.cpu().numpy(), 1e-3 * z["grad_cov_quat"]) < 1e-4
    assert rel_err(cs.grad.cpu().numpy(), 1e-3 * z["grad_cov_scale"]) < 1e-5


def test_normal_guidance_full_frame_against_oracle_and_nan_propagation():
    from oracle import loss_oracle as lo
    from vegs_amd import losses, scenes
    rng = np.random.default_rng(5)
    H, W = 376, 1408
    q = rng.normal(size=(4, H, W)).astype(np.float32)
    s = rng.uniform(1e-4, 0.3, (3, H, W)).astype(np.float32)
    n = rng.normal(size=(3, H, W)).astype(np.float32)
    n /= np.linalg.norm(n, axis=0, keepdims=True)
    want, dq, ds = lo.normal_guidance(q, s, n, scenes.R_KITTI)
    cq = torch.tensor(q, device=DEV, requires_grad=True)
    cs = torch.tensor(s, device=DEV, requires_grad=True)
    loss = losses.loss_normal_guidance(_cam(torch.tensor(n, device=DEV), scenes.R_KITTI), cq, cs)
    loss.backward()
    assert abs(loss.item() - want) < 2e-6
    assert rel_err(cq.grad.cpu().numpy(), dq) < 1e-4 and rel_err(cs.grad.cpu().numpy(), ds) < 1e-5
    # uncovered pixel (cov_quat = 0): the reference's 2/|q|^2 gives inf and the loss turns NaN -- same here
    q[:, 10, 10] = 0
    bad = losses.loss_normal_guidance(_cam(torch.tensor(n, device=DEV), scenes.R_KITTI), torch.tensor(q, device=DEV),
                                      torch.tensor(s, device=DEV))
    assert torch.isnan(bad).item()


@pytest.mark.parametrize("guard", [False, True])
def test_training_loss_block_equals_its_parts(guard):
    """losses.training_loss (train.py:162-168 as one node: 3 launches forward, 2 backward) against the reference outputs the
    individual losses are pinned by (ref_photometric / ref_normal_guidance goldens) and, at the full frame, against the
    composition of the separately tested losses -- with the uncovered-pixel guard against torch.where in front of them."""
    from vegs_amd import losses, scenes
    zp = np.load(os.path.join(GOLDEN, "ref_photometric.npz"))
    zn = np.load(os.path.join(GOLDEN, "ref_normal_guidance.npz"))
    lam, lam_n = 0.2, 0.03
    if not guard:
        # the goldens' own sizes differ between the two files: check the terms of each at its size through the block
        x, y = zp["img_a"], zp["gt_a"]
        C, H, W = x.shape
        rng = np.random.default_rng(1)
        q = rng.normal(size=(4, H, W)).astype(np.float32)
        sc = rng.uniform(1e-3, 0.3, (3, H, W)).astype(np.float32)
        n = rng.normal(size=(3, H, W)).astype(np.float32)
        loss, aux = losses.training_loss(torch.tensor(x, device=DEV), torch.tensor(y, device=DEV),
                                         _cam(torch.tensor(n, device=DEV), scenes.R_KITTI), torch.tensor(q, device=DEV),
                                         torch.tensor(sc, device=DEV), lam, lam_n)
        assert abs(aux[0].item() - float(zp["l1_a"])) < 1e-6 and abs(aux[1].item() - float(zp["ssim_a"])) < 1e-6
        cq, cs = torch.tensor(zn["cov_quat"], device=DEV), torch.tensor(zn["cov_scale"], device=DEV)
        Hn, Wn = cq.shape[1:]
        img = torch.rand(3, Hn, Wn, device=DEV)
        _, aux_n = losses.training_loss(img, img.clone(), _cam(torch.tensor(zn["normal"], device=DEV), zn["R"]), cq, cs, lam, lam_n)
        assert abs(aux_n[2].item() - float(zn["loss"])) < 1e-6 and abs(aux_n[0].item()) == 0.0
    rng = np.random.default_rng(8)
    H, W = 376, 1408
    x = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    y = np.clip(x + rng.normal(0, 0.1, x.shape), 0, 1).astype(np.float32)
    q = rng.normal(size=(4, H, W)).astype(np.float32)
    sc = rng.uniform(1e-4, 0.3, (3, H, W)).astype(np.float32)
    n = rng.normal(size=(3, H, W)).astype(np.float32)
    n /= np.linalg.norm(n, axis=0, keepdims=True)
    if guard:
        hole = rng.random((H, W)) < 0.15                      # sky: no Gaussian covers these pixels
        q[:, hole] = 0.0
        sc[:, hole] = 0.0
    cam = _cam(torch.tensor(n, device=DEV), scenes.R_KITTI)
    yt = torch.tensor(y, device=DEV)

    def leaves():
        return [torch.tensor(a, device=DEV, requires_grad=True) for a in (x, q, sc)]
    xt, qt, st = leaves()
    qq = torch.where((qt.detach() * qt.detach()).sum(0, keepdim=True) > 0, qt, torch.ones_like(qt)) if guard else qt
    want = losses.photometric_loss(xt, yt, lam)[0] + lam_n * losses.loss_normal_guidance(cam, qq, st)
    (2.5 * want).backward()
    x2, q2, s2 = leaves()
    got, aux = losses.training_loss(x2, yt, cam, q2, s2, lam, lam_n, guard_empty=guard)
    assert not aux.requires_grad and got.requires_grad
    (2.5 * got).backward()
    assert torch.isfinite(got).item() and abs(got.item() - want.item()) < 2e-6 * max(1.0, abs(want.item()))
    for a, b, name in ((x2, xt, "image"), (q2, qt, "cov_quat"), (s2, st, "cov_scale")):
        ga, gb = a.grad.cpu().numpy(), b.grad.cpu().numpy()
        assert np.isfinite(ga).all(), name
        assert np.abs(ga - gb).max() <= 2e-6 * np.abs(gb).max() + 1e-12, name      # (weights rounded in a different order)
    if guard:
        assert float(q2.grad[:, torch.tensor(hole, device=DEV)].abs().max()) == 0.0
    with torch.no_grad():                                       # evaluation: no derivative maps, same value
        assert abs(losses.training_loss(x2, yt, cam, q2, s2, lam, lam_n, guard_empty=guard)[0].item() - got.item()) == 0.0
    with pytest.raises(ValueError, match="GPU"):
        losses.training_loss(x2.cpu(), yt.cpu(), cam, q2, s2, lam, lam_n)
    with pytest.raises(ValueError, match="cov_quat"):
        losses.training_loss(x2, yt, cam, q2[:3], s2, lam, lam_n)


def test_losses_feed_the_rasterizer_backward():
    """The whole loss block of train.py:162-168 on a rendered frame: gradients reach the Gaussians."""
    from vegs_amd import harness, losses, scenes
    sc, deg = scenes.scene_random(P=4000, sh_degree=1, seed=9, scale=0.05)
    cam = scenes.camera_c1(160, 96)
    t = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in sc.items()}
    pkg = harness.render(cam, t, deg, torch.zeros(3, device=DEV))
    rng = np.random.default_rng(1)
    gt = torch.tensor(rng.uniform(0, 1, (3, 96, 160)).astype(np.float32), device=DEV)
    nrm = torch.tensor(rng.normal(size=(3, 96, 160)).astype(np.float32), device=DEV)
    loss, _ = losses.photometric_loss(pkg["render"], gt, 0.2)
    covered = (pkg["render_cov_quat"].detach().abs().sum(0, keepdim=True) > 0)
    quat = torch.where(covered, pkg["render_cov_quat"], torch.ones_like(pkg["render_cov_quat"]))
    loss = loss + 1e-3 * losses.loss_normal_guidance(_cam(nrm, scenes.R_KITTI), quat, pkg["render_cov_scale"])
    loss.backward()
    for k in ("means3D", "shs", "scales", "rotations", "opacities"):
        assert t[k].grad is not None and torch.isfinite(t[k].grad).all() and t[k].grad.abs().sum() > 0, k


def test_loss_argument_checks():
    from vegs_amd import losses
    a = torch.zeros(3, 8, 8)
    with pytest.raises(ValueError):
        losses.l1_loss(a, a)
    with pytest.raises(ValueError):
        losses.ssim(a.to(DEV), torch.zeros(3, 8, 9, device=DEV))
    with pytest.raises(NotImplementedError):
        losses.ssim(a.to(DEV), a.to(DEV), window_size=7)
