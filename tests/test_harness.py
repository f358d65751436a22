"""CPU: the device-agnostic counterpart of the reference's renderer glue (vegs_amd/harness.py) is pinned
against outputs of the reference's own functions (tests/golden/ref_camera.npz)."""
import os

import numpy as np
import torch

from helpers import GOLDEN
from vegs_amd import harness


def test_quaternion_matrix_round_trip_matches_reference():
    z = np.load(os.path.join(GOLDEN, "ref_camera.npz"))
    got = harness.quaternion_to_matrix(torch.tensor(z["quat"])).numpy()
    assert np.abs(got - z["quat_matrix"]).max() < 1e-6          # utils/graphics_utils.py:204-248
    got_q = harness.matrix_to_quaternion(torch.tensor(z["rotmat"])).numpy()
    assert np.abs(got_q - z["rotmat_quat"]).max() < 1e-6        # utils/graphics_utils.py:140-201 (same sign choice)


def test_prepare_rasterization_box_transform_is_differentiable():
    """box2world = similarity transform: means move rigidly, scales scale, rotations compose."""
    torch.manual_seed(0)
    n = 20
    t = dict(means3D=torch.randn(n, 3, dtype=torch.float64), shs=torch.randn(n, 16, 3, dtype=torch.float64),
             opacities=torch.rand(n, 1, dtype=torch.float64), scales=torch.rand(n, 3, dtype=torch.float64) + 0.1,
             rotations=torch.nn.functional.normalize(torch.randn(n, 4, dtype=torch.float64), dim=1))
    ang = torch.tensor(0.7, dtype=torch.float64, requires_grad=True)
    s = torch.tensor(1.8, dtype=torch.float64, requires_grad=True)
    c, sn = torch.cos(ang), torch.sin(ang)
    Rz = torch.stack([torch.stack([c, -sn, torch.zeros_like(c)]), torch.stack([sn, c, torch.zeros_like(c)]),
                      torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)])
    b2w = torch.eye(4, dtype=torch.float64)
    b2w = torch.cat([torch.cat([Rz * s, torch.tensor([[1.0], [2.0], [3.0]], dtype=torch.float64)], 1),
                     torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=torch.float64)], 0)
    out = harness.prepare_rasterization(t, b2w)
    want_means = (Rz.detach() * s.detach()) @ t["means3D"].t()
    assert torch.allclose(out["means3D"], want_means.t() + torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64))
    assert torch.allclose(out["scales"], t["scales"] * s.detach())
    Rm = harness.quaternion_to_matrix(out["rotations"])
    assert torch.allclose(Rm, Rz.detach()[None] @ harness.quaternion_to_matrix(t["rotations"]), atol=1e-9)
    (out["means3D"].sum() + out["scales"].sum() + out["rotations"][:, 0].sum()).backward()
    assert ang.grad is not None and s.grad is not None and torch.isfinite(ang.grad) and s.grad.abs() > 0
