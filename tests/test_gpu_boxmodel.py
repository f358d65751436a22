"""GPU (-m gpu): the BoxModel kernels (vegs_amd/csrc/instances.hip: k_box_fwd / k_box_bwd / k_box_reg, C ABI
include/vegs_instances.h) against outputs of the reference's own class (tests/golden/ref_boxmodel.npz) and against the
op-by-op ATen composition; the multi-optimizer Adam launch (vegs_amd.optim.step_many) against torch.optim.Adam."""
import numpy as np
import pytest
import torch

from test_boxmodel import REF, _model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_all_instances_in_one_launch_match_the_reference_class():
    from vegs_amd import boxmodel
    n = REF["box2world"].shape[0]
    bms = [_model(i, fused=True, device=DEV) for i in range(n)]
    adj = boxmodel.adjust_all(bms)
    assert adj.shape == (n, 4, 4)
    for i in range(n):
        assert np.allclose(adj[i].detach().cpu().numpy(), REF["adjusted"][i], rtol=0, atol=2e-6 * np.abs(REF["adjusted"][i]).max())
    adj.backward(torch.tensor(REF["g_adjusted"], device=DEV))
    for i, bm in enumerate(bms):
        for name, t in (("grad_delta_r", bm.delta_r), ("grad_delta_s", bm.delta_s), ("grad_delta_t", bm.delta_t)):
            want = REF[name][i]
            assert np.allclose(t.grad.cpu().numpy(), want, rtol=0, atol=3e-6 * max(np.abs(want).max(), 1.0)), (i, name)
    # single-model surface, as return_gaussians_boxes_and_box2worlds calls it (gaussian_renderer/__init__.py:343)
    one = bms[2].adjustbox2world()
    assert one.shape == (4, 4) and torch.equal(one, adj[2].detach())


def test_step_and_regularize_rounds_match_the_reference_class():
    from vegs_amd import boxmodel, optim
    n = REF["box2world"].shape[0]
    bms = [_model(i, fused=True, device=DEV) for i in range(n)]
    for it in range(REF["after_r"].shape[1]):
        for i, bm in enumerate(bms):
            bm.delta_r.grad = torch.tensor(REF["step_g_r"][i, it], device=DEV)
            bm.delta_s.grad = torch.tensor(REF["step_g_s"][i, it], device=DEV)
            bm.delta_t.grad = torch.tensor(REF["step_g_t"][i, it], device=DEV)
        optim.step_many([bm.optimizer for bm in bms])            # train.py:272-273 for every model: ONE launch
        for bm in bms:
            bm.optimizer.zero_grad()
        boxmodel.regularize_all(bms)                             # :274
        for i, bm in enumerate(bms):
            for name, t in (("after_r", bm.delta_r), ("after_s", bm.delta_s), ("after_t", bm.delta_t)):
                assert np.allclose(t.detach().cpu().numpy(), REF[name][i, it], rtol=0, atol=2e-6), (i, it, name)
            assert bm.delta_r.grad is None


def test_nan_guard_of_the_training_loop():
    """train.py:199-205: a NaN in delta_r.grad or delta_s.grad zeroes all three gradients of that box model."""
    from vegs_amd import boxmodel
    bms = [_model(i, fused=True, device=DEV) for i in range(3)]
    g = torch.tensor(REF["g_adjusted"][:3], device=DEV).clone()
    g[1, 0, 0] = float("nan")                                   # reaches delta_r / delta_s of instance 1 only
    boxmodel.adjust_all(bms).backward(g)
    for t in (bms[1].delta_r, bms[1].delta_s, bms[1].delta_t):
        assert torch.equal(t.grad, torch.zeros_like(t))
    assert np.allclose(bms[0].delta_r.grad.cpu().numpy(), REF["grad_delta_r"][0], atol=1e-5)
    assert torch.isfinite(bms[2].delta_s.grad).all() and bms[2].delta_s.grad.abs().sum() > 0
    bms = [_model(i, fused=True, device=DEV) for i in range(3)]
    boxmodel.adjust_all(bms, nan_guard=False).backward(g)
    assert torch.isnan(bms[1].delta_r.grad).any()


def test_step_many_equals_one_torch_adam_per_optimizer():
    """64 tensors of three optimizers with different learning rates, eps and step counts in one launch."""
    from vegs_amd import optim
    rng = np.random.default_rng(3)
    shapes = [(5000, 3), (5000, 1, 3), (5000, 15, 3), (5000, 1), (5000, 3), (5000, 4), (4,), (3,), (3,), (0, 3), (1025,)]
    mine, theirs = [], []
    for k in range(7):                                           # 77 tensors: two launches' worth
        vals = [rng.normal(size=s).astype(np.float32) for s in shapes]
        eps = 1e-15 if k % 2 == 0 else 1e-8
        pa = [torch.nn.Parameter(torch.tensor(v, device=DEV)) for v in vals]
        pb = [torch.nn.Parameter(torch.tensor(v)) for v in vals]
        mk = lambda ps: [{"params": [p], "lr": 1e-3 * (1 + j), "name": str(j)} for j, p in enumerate(ps)]
        mine.append((optim.Adam(mk(pa), lr=0.0, eps=eps), pa))
        theirs.append((torch.optim.Adam(mk(pb), lr=0.0, eps=eps), pb))
    for it in range(4):
        active = [k for k in range(7) if (it + k) % 3 != 0]      # optimizers skip iterations: their step counts diverge
        for k in active:
            for pa, pb in zip(mine[k][1], theirs[k][1]):
                if pa.numel() and rng.random() < 0.9:
                    g = rng.normal(size=tuple(pa.shape)).astype(np.float32) * 1e-2
                    pa.grad, pb.grad = torch.tensor(g, device=DEV), torch.tensor(g)
        optim.step_many([mine[k][0] for k in active])
        for k in active:
            theirs[k][0].step()
            mine[k][0].zero_grad(set_to_none=True)
            theirs[k][0].zero_grad(set_to_none=True)
    for (oa, pa), (ob, pb) in zip(mine, theirs):
        for a, b in zip(pa, pb):
            assert np.allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-6, atol=2e-7)
            if b in ob.state:
                assert float(oa.state[a]["step"]) == float(ob.state[b]["step"])
