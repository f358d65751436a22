"""GPU (-m gpu): the FACTORED SH gradient (include/vegs_rast.h VrInGrads.dL_dcolors_sh; vegs_optim.h
vr_sh_grad_from_factors / vr_sh_adam_step).  dL/dshs of a view is basis(dir) x (clamp-masked dL/dcolour): the op can
hand out the 3-float factor instead of the 48-float row, a view-sharded job exchanges the factors (vegs_amd.dist
.exchange_factored), and the optimizer consumes them without the dense gradient ever existing.  Everything here is
checked against the dense path of the same operator (which the parity tests pin against the oracle)."""
import numpy as np
import pytest
import torch

from helpers import assert_grad_close, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(P=30000, deg=3, seed=3):
    from vegs_amd import scenes
    sc, deg = scenes.scene_street(P=P, length=60.0, sh_degree=deg, seed=seed)
    sc["shs"][:, 0, :] -= 1.2 * (np.arange(P) % 5 == 0)[:, None]       # some colours clamp at zero
    return sc, deg


def _render(sc, deg, cam, g, factored, split=False, M=16):
    from vegs_amd import harness, rasterizer
    from vegs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    T = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in sc.items()}
    dc = T["shs"][:, :1].detach().clone().contiguous().requires_grad_(True)
    rest = T["shs"][:, 1:M].detach().clone().contiguous().requires_grad_(True)
    whole = T["shs"][:, :M].detach().clone().contiguous().requires_grad_(True)
    ct = harness.cam_tensors(cam, torch.device(DEV))
    rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy,
                                       torch.zeros(3, device=DEV), 1.0, ct["viewmatrix"], ct["projmatrix"], deg,
                                       ct["campos"], False, False)
    m2d = torch.zeros(T["means3D"].shape[0], 3, device=DEV, requires_grad=True)
    sink = torch.zeros(T["means3D"].shape[0], 3, device=DEV, requires_grad=True) if factored else None
    with rasterizer.flags(rasterizer.FLAG_DETERMINISTIC):          # bit-reproducible gradients: exact comparisons below
        out = GaussianRasterizer(rs)(means3D=T["means3D"], means2D=m2d, opacities=T["opacities"],
                                     shs=(dc, rest) if split else whole, scales=T["scales"], rotations=T["rotations"],
                                     sh_color_grad=sink)
        torch.autograd.backward([out[0], out[2], out[3]], [torch.tensor(x, device=DEV) for x in g])
    return dict(T=T, dc=dc, rest=rest, whole=whole, m2d=m2d, sink=sink, campos=ct["campos"], radii=out[5])


@pytest.mark.parametrize("deg,M,split", [(3, 16, False), (3, 16, True), (1, 16, True), (2, 9, False), (0, 1, False)])
def test_factor_times_basis_equals_the_dense_sh_gradient(deg, M, split):
    from vegs_amd import optim, scenes
    sc, _ = _scene(deg=deg)
    cam = scenes.kitti_camera(0.0, 0.3, 688, 188)
    rng = np.random.default_rng(1)
    g = [rng.normal(size=(k, 188, 688)).astype(np.float32) for k in (3, 4, 3)]
    if M == 1:
        split = False
    a = _render(sc, deg, cam, g, factored=False, split=split, M=M)
    b = _render(sc, deg, cam, g, factored=True, split=split, M=M)
    assert b["whole"].grad is None and b["dc"].grad is None and b["rest"].grad is None      # no dense SH gradient
    f = b["sink"].grad
    assert f.shape == (sc["means3D"].shape[0], 3) and torch.all(f[a["radii"] == 0] == 0)
    for k in ("means3D", "opacities", "scales", "rotations"):                               # everything else unchanged
        assert torch.equal(a["T"][k].grad, b["T"][k].grad), k
    assert torch.equal(a["m2d"].grad, b["m2d"].grad)
    got = optim.sh_grad_from_factors(b["T"]["means3D"].detach(), b["campos"][None], f[None], deg, M, 1.0, split=split)
    if split:
        assert torch.equal(got[0], a["dc"].grad) and torch.equal(got[1], a["rest"].grad)     # one view: bit-exact
    else:
        assert torch.equal(got, a["whole"].grad)
    assert (f != 0).any() and float((f == 0).float().mean()) > 0.05


@pytest.mark.parametrize("split", [True, False])
def test_adam_from_factors_equals_adam_on_the_dense_gradient(split):
    """Three views (as three ranks would contribute), Adam on the SH tensors: fused from the factors vs. the reference's
    optimizer (torch.optim.Adam, CPU) on the dense mean gradient."""
    from vegs_amd import optim, scenes
    sc, deg = _scene(P=20000)
    cams = [scenes.kitti_camera(0.0, 0.3, 688, 188), scenes.kitti_camera(0.0, -0.3, 688, 188), scenes.kitti_camera(6.0, 0.3, 688, 188)]
    rng = np.random.default_rng(2)
    dense, factors, campos = None, [], []
    for cam in cams:
        g = [rng.normal(size=(k, 188, 688)).astype(np.float32) for k in (3, 4, 3)]
        a = _render(sc, deg, cam, g, factored=False)
        b = _render(sc, deg, cam, g, factored=True)
        dense = a["whole"].grad if dense is None else dense + a["whole"].grad
        factors.append(b["sink"].grad)
        campos.append(b["campos"])
    dense = (dense / 3).cpu()
    F, Cc = torch.stack(factors), torch.stack(campos)
    means = torch.tensor(sc["means3D"], device=DEV)
    rebuilt = optim.sh_grad_from_factors(means, Cc, F, deg, 16, 1.0 / 3)
    assert rel_err(rebuilt.cpu().numpy(), dense.numpy()) < 2e-6
    assert_grad_close("rebuilt", rebuilt.cpu().numpy(), dense.numpy(), rtol=1e-5, floor=1e-7, outliers=0, near=0)

    shs = torch.tensor(sc["shs"])
    if split:
        ref_p = [torch.nn.Parameter(shs[:, :1].clone().contiguous()), torch.nn.Parameter(shs[:, 1:].clone().contiguous())]
    else:
        ref_p = [torch.nn.Parameter(shs.clone())]
    our_p = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref_p]
    lrs = [2.5e-3, 2.5e-3 / 20]
    ref_opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref_p, lrs)], lr=0.0, eps=1e-15)
    our_opt = optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(our_p, lrs)], lr=0.0, eps=1e-15)
    for step in range(3):
        if split:
            ref_p[0].grad, ref_p[1].grad = dense[:, :1].contiguous() * (step + 1), dense[:, 1:].contiguous() * (step + 1)
        else:
            ref_p[0].grad = dense * (step + 1)
        ref_opt.step()
        optim.adam_step_sh_factored(our_opt, our_p[0], our_p[1] if split else None, means, Cc, F, deg, (step + 1) / 3.0)
    for p, q in zip(ref_p, our_p):
        assert rel_err(q.detach().cpu().numpy(), p.detach().numpy()) < 3e-6
        st, sr = our_opt.state[q], ref_opt.state[p]
        assert float(st["step"]) == 3.0
        assert rel_err(st["exp_avg"].cpu().numpy(), sr["exp_avg"].numpy()) < 3e-6
        assert rel_err(st["exp_avg_sq"].cpu().numpy(), sr["exp_avg_sq"].numpy()) < 3e-6
