"""CPU: the op-by-op BoxModel statement (oracle/boxmodel_oracle.py: BoxModelOpByOp) against outputs of the reference's OWN class
(model/boxmodel.py, run by tests/golden/make_golden.py part_e -> ref_boxmodel.npz): adjustbox2world(), its gradients, and
three rounds of train.py:270-274 (optimizer.step, zero_grad, regularize).  The HIP kernels are compared with both in
tests/test_gpu_boxmodel.py."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = np.load(os.path.join(HERE, "golden", "ref_boxmodel.npz"))


def _model(i, fused=False, device="cpu"):
    if fused:
        from vegs_amd.boxmodel import BoxModel
    else:
        from oracle.boxmodel_oracle import BoxModelOpByOp as BoxModel
    bm = BoxModel(torch.tensor(REF["box2world"][i]), lr=float(REF["lr"]), lambda_reg=float(REF["lambda_reg"]), device=device,
                  fused=fused)
    with torch.no_grad():
        bm.delta_r.copy_(torch.tensor(REF["delta_r"][i]))
        bm.delta_s.copy_(torch.tensor(REF["delta_s"][i]))
        bm.delta_t.copy_(torch.tensor(REF["delta_t"][i]))
    return bm


def test_adjustbox2world_and_gradients_match_the_reference_class():
    for i in range(REF["box2world"].shape[0]):
        bm = _model(i)
        a = bm.adjustbox2world()
        assert np.allclose(a.detach().numpy(), REF["adjusted"][i], rtol=0, atol=2e-6 * np.abs(REF["adjusted"][i]).max())
        a.backward(torch.tensor(REF["g_adjusted"][i]))
        for name, t in (("grad_delta_r", bm.delta_r), ("grad_delta_s", bm.delta_s), ("grad_delta_t", bm.delta_t)):
            want = REF[name][i]
            assert np.allclose(t.grad.numpy(), want, rtol=0, atol=3e-6 * max(np.abs(want).max(), 1.0)), (i, name)


def test_step_and_regularize_rounds_match_the_reference_class():
    for i in range(REF["box2world"].shape[0]):
        bm = _model(i)
        for it in range(REF["after_r"].shape[1]):
            bm.delta_r.grad = torch.tensor(REF["step_g_r"][i, it])
            bm.delta_s.grad = torch.tensor(REF["step_g_s"][i, it])
            bm.delta_t.grad = torch.tensor(REF["step_g_t"][i, it])
            bm.optimizer.step()
            bm.optimizer.zero_grad()
            bm.regularize(it + 1)
            for name, t in (("after_r", bm.delta_r), ("after_s", bm.delta_s), ("after_t", bm.delta_t)):
                assert np.allclose(t.detach().numpy(), REF[name][i, it], rtol=0, atol=2e-6), (i, it, name)
    assert np.array_equal(REF["delta_r"][0], [1, 0, 0, 0])      # the 0/0 case of the norms is in the fixture
