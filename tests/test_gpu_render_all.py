"""GPU (-m gpu): render_all-shaped call (reference gaussian_renderer/__init__.py:263-333): the op's inputs
are torch.cat's of a static model and box instances carried into the world frame by differentiable
box2world transforms, so the gradients of means3D, scales and rotations have to flow correctly into
the box parameters (model/boxmodel.py:30-42).  Checked against the oracle by running the SAME torch
graph on the CPU in float64 and feeding it the oracle's op-input gradients."""
import numpy as np
import pytest
import torch

from helpers import assert_grad_close, oracle_cam, rel_err

pytestmark = pytest.mark.gpu


def _box2world(params, dtype, device):
    """similarity transform from learnable (angle, log-scale, translation) -- a BoxModel-like delta"""
    ang, logs, tr = params
    c, s = torch.cos(ang), torch.sin(ang)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    Rz = torch.stack([torch.stack([c, -s, z]), torch.stack([s, c, z]), torch.stack([z, z, o])])
    top = torch.cat([Rz * torch.exp(logs), tr[:, None]], 1)
    return torch.cat([top, torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=dtype, device=device)], 0)


def _build(dev, dtype, sc_static, sc_boxes, box_params_np):
    static = {k: torch.tensor(v, dtype=dtype, device=dev, requires_grad=True) for k, v in sc_static.items()}
    boxes = [{k: torch.tensor(v, dtype=dtype, device=dev, requires_grad=True) for k, v in b.items()} for b in sc_boxes]
    bparams = [[torch.tensor(p, dtype=dtype, device=dev, requires_grad=True) for p in bp] for bp in box_params_np]
    b2ws = [_box2world(bp, dtype, dev) for bp in bparams]
    return static, boxes, bparams, b2ws


def test_render_all_gradients_reach_box_parameters():
    from oracle import oracle as orc
    from vegs_amd import harness, scenes
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    sc_static, deg = scenes.scene_random(P=3000, sh_degree=2, seed=31, scale=0.04)
    sc_boxes = [scenes.scene_random(P=400, sh_degree=2, seed=40 + i, extent=0.15, scale=0.03)[0] for i in range(2)]
    box_params = [(np.float64(0.6), np.float64(0.2), np.array([0.25, 0.1, -0.1])),
                  (np.float64(-1.1), np.float64(-0.3), np.array([-0.3, -0.05, 0.2]))]
    cam = scenes.camera_c1(160, 112)
    rng = np.random.default_rng(2)
    gc, gq, gs = (rng.normal(size=s).astype(np.float32) for s in [(3, 112, 160), (4, 112, 160), (3, 112, 160)])

    # ---- HIP path (fp32, GPU)
    static, boxes, bparams, b2ws = _build(dev, torch.float32, sc_static, sc_boxes, box_params)
    pkg = harness.render_all(cam, static, boxes, b2ws, deg, torch.zeros(3, device=dev))
    torch.autograd.backward([pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]],
                            [torch.tensor(g, device=dev) for g in (gc, gq, gs)])
    P_all = 3000 + 800
    assert pkg["viewspace_points"].grad.shape == (P_all, 3) and pkg["radii"].shape == (P_all,)

    # ---- oracle on the concatenated op inputs, then the same graph on the CPU in float64
    kw32 = {k: v.detach().cpu().numpy() for k, v in pkg["op_inputs"].items()}
    oc = oracle_cam(cam, [0, 0, 0], deg)
    o_out, st = orc.forward(oc, kw32["means3D"], kw32["shs"], None, kw32["opacities"], kw32["scales"], kw32["rotations"], None)
    assert np.array_equal(pkg["radii"].cpu().numpy(), o_out["radii"])
    assert np.array_equal(pkg["render"].detach().cpu().numpy(), o_out["color"])
    og = orc.backward(oc, st, gc, None, gq, gs, None)
    cpu = torch.device("cpu")
    c_static, c_boxes, c_bparams, c_b2ws = _build(cpu, torch.float64, sc_static, sc_boxes, box_params)
    kw = harness.prepare_rasterization(c_static)
    for t, b in zip(c_boxes, c_b2ws):
        kw = harness.merge_kwargs(kw, harness.prepare_rasterization(t, b))
    keys = ["means3D", "shs", "opacities", "scales", "rotations"]
    torch.autograd.backward([kw[k] for k in keys], [torch.tensor(og[k], dtype=torch.float64) for k in keys])

    for k in keys:
        assert_grad_close(k, static[k].grad.cpu().numpy(), c_static[k].grad.numpy())
        for b, cb in zip(boxes, c_boxes):
            assert_grad_close("box " + k, b[k].grad.cpu().numpy(), cb[k].grad.numpy())
    for bp, cbp in zip(bparams, c_bparams):           # the BoxModel-like deltas: angle, log-scale, translation
        for p, cp in zip(bp, cbp):
            assert p.grad is not None and torch.isfinite(p.grad).all()
            assert rel_err(p.grad.cpu().numpy(), cp.grad.numpy()) < 2e-3
