"""CPU: every product entry point refuses CPU tensors loudly -- there is no CPU / PyTorch fallback behind any of
them (the oracle is test infrastructure and is never reached from the product path)."""
import types

import numpy as np
import pytest
import torch


def test_losses_refuse_cpu_tensors():
    from vegs_amd import losses
    img = torch.rand(3, 16, 16)
    with pytest.raises(ValueError):
        losses.l1_loss(img, img)
    with pytest.raises(ValueError):
        losses.ssim(img, img)
    with pytest.raises(ValueError):
        losses.photometric_loss(img, img, 0.2)
    cam = types.SimpleNamespace(original_normal=torch.rand(3, 16, 16), R=np.eye(3))
    with pytest.raises(ValueError):
        losses.loss_normal_guidance(cam, torch.rand(4, 16, 16), torch.rand(3, 16, 16))


def test_adam_and_statistics_refuse_cpu_tensors():
    from vegs_amd import optim
    p = torch.nn.Parameter(torch.zeros(10, 3))
    opt = optim.Adam([p], lr=1e-3)
    p.grad = torch.ones_like(p)
    with pytest.raises(ValueError):
        opt.step()
    with pytest.raises(NotImplementedError):
        optim.Adam([p], lr=1e-3, weight_decay=0.1)
    with pytest.raises(ValueError):
        optim.add_densification_stats(torch.zeros(10, 3), torch.zeros(10, dtype=torch.int32), torch.zeros(10, 1),
                                      torch.zeros(10, 1), torch.zeros(10))


def test_instances_knn_and_rasterizer_refuse_cpu_tensors():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from simple_knn._C import distCUDA2
    from vegs_amd import instances
    t = {"means3D": torch.zeros(4, 3), "scales": torch.ones(4, 3), "rotations": torch.ones(4, 4),
         "shs": torch.zeros(4, 1, 3), "opacities": torch.ones(4, 1)}
    with pytest.raises(ValueError):
        instances.prepare_and_merge(t, [], [])
    with pytest.raises(ValueError):
        distCUDA2(torch.zeros(10, 3))
    st = GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3),
                                       False, False)
    with pytest.raises(ValueError):
        GaussianRasterizer(raster_settings=st)(means3D=t["means3D"], means2D=torch.zeros(4, 3), shs=t["shs"],
                                               opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])


def test_trainer_rejects_a_factored_exchange_without_the_fused_operator():
    """exchange='factored' / 'direct' move the 3-float SH factor, an output of the FUSED operator: with fused=False the
    trainer used to arm an exchange that never got a factor (advisor finding, round 4)."""
    import pytest
    from vegs_amd import iteration
    for ex in ("factored", "direct"):
        with pytest.raises(ValueError, match="fused=True"):
            iteration.Trainer({}, "cpu", fused=False, world=2, rank=0, exchange=ex)
    with pytest.raises(ValueError, match="exchange must be"):
        iteration.Trainer({}, "cpu", exchange="ring")
