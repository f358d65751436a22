"""GPU (-m gpu): fused instance transform + concatenation (row N4) against (1) outputs of the reference's own
functions + autograd (tests/golden/ref_instances.npz) and (2) the float64 oracle on a render_all-sized frame."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_instance_transform_matches_reference_outputs():
    from vegs_amd.instances import prepare_and_merge
    z = np.load(os.path.join(GOLDEN, "ref_instances.npz"))
    boxes, b2ws = [], []
    for b in range(4):
        t = {"means3D": z[f"xyz_{b}"], "scales": z[f"scales_{b}"], "rotations": z[f"rot_{b}"],
             "shs": np.zeros((60, 4, 3), np.float32), "opacities": np.full((60, 1), 0.5, np.float32)}
        boxes.append({k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in t.items()})
        b2ws.append(torch.tensor(z[f"box2world_{b}"], device=DEV, requires_grad=True))
    kw = prepare_and_merge(None, boxes, b2ws)                       # render_dyn: instances only
    gm = np.concatenate([z[f"gout_means_{b}"] for b in range(4)])
    gs = np.concatenate([z[f"gout_scales_{b}"] for b in range(4)])
    gr = np.concatenate([z[f"gout_rot_{b}"] for b in range(4)])
    torch.autograd.backward([kw["means3D"], kw["scales"], kw["rotations"]],
                            [torch.tensor(g, device=DEV) for g in (gm, gs, gr)])
    for b in range(4):
        sl = slice(60 * b, 60 * b + 60)
        assert np.abs(kw["means3D"][sl].detach().cpu().numpy() - z[f"out_means_{b}"]).max() < 5e-6
        assert np.abs(kw["scales"][sl].detach().cpu().numpy() - z[f"out_scales_{b}"]).max() < 1e-6
        assert np.abs(kw["rotations"][sl].detach().cpu().numpy() - z[f"out_rot_{b}"]).max() < 2e-6
        assert rel_err(boxes[b]["means3D"].grad.cpu().numpy(), z[f"grad_xyz_{b}"]) < 2e-5
        assert rel_err(boxes[b]["scales"].grad.cpu().numpy(), z[f"grad_scales_{b}"]) < 2e-5
        assert rel_err(boxes[b]["rotations"].grad.cpu().numpy(), z[f"grad_rot_{b}"]) < 5e-5
        assert rel_err(b2ws[b].grad.cpu().numpy(), z[f"grad_box2world_{b}"]) < 5e-5
    assert kw["shs"].shape == (240, 4, 3) and kw["opacities"].shape == (240, 1)


def test_render_dyn_op_by_op_equals_fused():
    """render_dyn (gaussian_renderer/__init__.py:188-260: only the dynamic instances, no static model): the op-by-op
    composition (harness.render_all(static=None, fused=False)) and the fused one render the same frame -- images equal to
    the rounding of the transformed inputs, gradients of every instance leaf and pose equal per tensor."""
    from vegs_amd import harness, scenes
    rng = np.random.default_rng(16)
    nb, n = 3, 4000
    sc_boxes = [scenes.scene_random(P=n, sh_degree=1, seed=70 + i, extent=0.3, scale=0.03)[0] for i in range(nb)]
    Bs = []
    for i in range(nb):
        B = np.eye(4)
        B[:3, :3] = harness.quaternion_to_matrix(torch.tensor(rng.normal(size=4))).numpy() * rng.uniform(0.8, 1.3)
        B[:3, 3] = rng.uniform(-0.3, 0.3, 3)
        Bs.append(B.astype(np.float32))
    cam = scenes.camera_c1(192, 128)
    gouts = [torch.tensor(rng.normal(size=s).astype(np.float32), device=DEV) for s in [(3, 128, 192), (4, 128, 192), (3, 128, 192)]]
    res = []
    for fused in (False, True):
        bx = [{k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in b.items()} for b in sc_boxes]
        bw = [torch.tensor(B, device=DEV, requires_grad=True) for B in Bs]
        pkg = harness.render_all(cam, None, bx, bw, 1, torch.zeros(3, device=DEV), fused=fused)
        torch.autograd.backward([pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]], gouts)
        res.append((pkg, bx, bw))
    (pa, bxa, bwa), (pb, bxb, bwb) = res
    assert pa["op_inputs"]["means3D"].shape[0] == nb * n and torch.equal(pa["radii"], pb["radii"])
    for k in ("render", "render_depth", "render_cov_quat", "render_cov_scale", "alpha"):
        assert rel_err(pb[k].detach().cpu().numpy(), pa[k].detach().cpu().numpy()) < 2e-5, k
    for a, b in zip(bxa, bxb):
        for k in ("means3D", "scales", "rotations", "opacities", "shs"):
            assert rel_err(b[k].grad.cpu().numpy(), a[k].grad.cpu().numpy()) < 2e-3, k
    for a, b in zip(bwa, bwb):
        assert rel_err(b.grad.cpu().numpy(), a.grad.cpu().numpy()) < 2e-3
    with pytest.raises(ValueError):
        harness.render_all(cam, None, [], [], 1, torch.zeros(3, device=DEV), fused=False)


def test_fused_render_all_equals_the_op_by_op_composition():
    """C5-shaped: a static model + 20 instances of 8196 Gaussians (more than one launch table), through the
    rasterizer; forward images bit-identical inputs apart, gradients of every leaf vs the float64 oracle."""
    from oracle import instance_oracle as io
    from vegs_amd import harness, scenes
    sc_static, deg = scenes.scene_random(P=30_000, sh_degree=1, seed=3, scale=0.03)
    rng = np.random.default_rng(6)
    nb, n = 20, 8196
    sc_boxes = [scenes.scene_random(P=n, sh_degree=1, seed=50 + i, extent=0.1, scale=0.02)[0] for i in range(nb)]
    Bs = []
    for i in range(nb):
        q = rng.normal(size=4)
        R = np.asarray(harness.quaternion_to_matrix(torch.tensor(q)).numpy())
        B = np.eye(4)
        B[:3, :3] = R * rng.uniform(0.7, 1.5)
        B[:3, 3] = rng.uniform(-0.4, 0.4, 3)
        Bs.append(B.astype(np.float32))
    cam = scenes.camera_c1(256, 160)

    def leaves():
        st = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in sc_static.items()}
        bx = [{k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in b.items()} for b in sc_boxes]
        bw = [torch.tensor(B, device=DEV, requires_grad=True) for B in Bs]
        return st, bx, bw

    gouts = [torch.tensor(rng.normal(size=s).astype(np.float32), device=DEV) for s in [(3, 160, 256), (4, 160, 256), (3, 160, 256)]]
    res = []
    for fused in (False, True):
        st, bx, bw = leaves()
        pkg = harness.render_all(cam, st, bx, bw, deg, torch.zeros(3, device=DEV), fused=fused)
        torch.autograd.backward([pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]], gouts)
        res.append((pkg, st, bx, bw))
    (pa, sta, bxa, bwa), (pb, stb, bxb, bwb) = res
    for k in ("means3D", "scales", "rotations"):
        assert rel_err(pb["op_inputs"][k].detach().cpu().numpy(), pa["op_inputs"][k].detach().cpu().numpy()) < 1e-6, k
    # the fused path hands the SH rows over in parts -- the static model's tensor where it is + the instances' rows as an
    # SH tail (ABI v6) -- instead of one concatenated copy
    shs_b = pb["op_inputs"]["shs"]
    assert isinstance(shs_b, tuple) and shs_b[0].data_ptr() == stb["shs"].data_ptr() and shs_b[1] is None
    assert torch.equal(torch.cat((shs_b[0], shs_b[2]), 0), pa["op_inputs"]["shs"])
    assert torch.equal(pb["render"], pa["render"]) or rel_err(pb["render"].detach().cpu().numpy(), pa["render"].detach().cpu().numpy()) < 1e-5
    # oracle: the fused op's own inputs and upstream gradients
    kwb = pb["op_inputs"]
    # (gradients of the op inputs are not retained; recompute them on detached inputs)
    det = {k: (tuple(None if x is None else x.detach().clone().requires_grad_(True) for x in v) if isinstance(v, tuple)
               else v.detach().clone().requires_grad_(True)) for k, v in kwb.items()}
    p2 = harness.render(cam, det, deg, torch.zeros(3, device=DEV))
    torch.autograd.backward([p2["render"], p2["render_cov_quat"], p2["render_cov_scale"]], gouts)
    gm, gs, gr = (det[k].grad.cpu().numpy() for k in ("means3D", "scales", "rotations"))
    off = 30_000
    for i in (0, 7, 19):
        sl = slice(off + i * n, off + (i + 1) * n)
        want = io.backward(sc_boxes[i]["means3D"], sc_boxes[i]["scales"], sc_boxes[i]["rotations"], Bs[i], gm[sl], gs[sl], gr[sl])
        got = (bxb[i]["means3D"].grad, bxb[i]["scales"].grad, bxb[i]["rotations"].grad, bwb[i].grad)
        for g, w, name in zip(got, want, ("means3D", "scales", "rotations", "box2world")):
            assert rel_err(g.cpu().numpy(), w) < 2e-4, (i, name)
    for k in ("means3D", "scales", "rotations", "shs", "opacities"):     # static model: plain slices
        assert rel_err(stb[k].grad.cpu().numpy(), sta[k].grad.cpu().numpy()) < 1e-4, k
    for ba, bb in zip(bwa, bwb):                                         # and the op-by-op composition agrees
        assert rel_err(bb.grad.cpu().numpy(), ba.grad.cpu().numpy()) < 2e-3


def test_instance_argument_checks():
    from vegs_amd.instances import prepare_and_merge
    t = {"means3D": torch.zeros(4, 3), "scales": torch.zeros(4, 3), "rotations": torch.zeros(4, 4),
         "shs": torch.zeros(4, 1, 3), "opacities": torch.zeros(4, 1)}
    with pytest.raises(ValueError):
        prepare_and_merge(t, [], [])
    g = {k: v.to(DEV) for k, v in t.items()}
    with pytest.raises(ValueError):
        prepare_and_merge(g, [g], [torch.eye(3, device=DEV)])
    with pytest.raises(ValueError):
        prepare_and_merge(g, [g], [])


def test_activations_match_the_reference_functions():
    """vegs_amd.instances.activate == torch.sigmoid / torch.exp / F.normalize as scene/gaussian_model.py:37-45 assigns
    them, against outputs and autograd gradients of those functions (tests/golden/ref_activations.npz, float32 CPU) incl.
    quaternions below F.normalize's eps, and against the same ATen ops on the GPU at 2 M rows."""
    import torch.nn.functional as F
    from vegs_amd.instances import activate
    z = np.load(os.path.join(GOLDEN, "ref_activations.npz"))
    raw = [torch.tensor(z[k], device=DEV, requires_grad=True) for k in ("raw_opacity", "raw_scaling", "raw_rotation")]
    out = activate(*raw)
    for y, k in zip(out, ("opacity", "scales", "rotations")):
        assert y.shape == z[k].shape
        assert rel_err(y.detach().cpu().numpy(), z[k]) < 3e-7, k
        assert np.abs(y.detach().cpu().numpy() - z[k]).max() <= 4e-7 * max(1.0, np.abs(z[k]).max()), k
    torch.autograd.backward(list(out), [torch.tensor(z[k], device=DEV) for k in ("g_opacity", "g_scales", "g_rotations")])
    for t, k in zip(raw, ("d_opacity", "d_scaling", "d_rotation")):
        got, want = t.grad.cpu().numpy(), z[k]
        # (the quaternion gradient (g - y <y,g>) / |x| cancels: its error scales with the row, 1.1e-6 of it in fp32
        # whichever way the terms are grouped)
        bad = np.abs(got - want) > 2e-6 * np.abs(want) + 3e-6 * np.abs(want).max(axis=-1, keepdims=True) + 1e-30
        assert not bad.any(), (k, np.argwhere(bad)[:5], got[bad][:5], want[bad][:5])
    # only some outputs used / only some inputs requiring grad
    a = torch.tensor(z["raw_opacity"], device=DEV, requires_grad=True)
    b = torch.tensor(z["raw_scaling"], device=DEV)
    c = torch.tensor(z["raw_rotation"], device=DEV, requires_grad=True)
    o, s, r = activate(a, b, c)
    (o.sum() * 2.0).backward()
    assert c.grad is None or float(c.grad.abs().max()) == 0.0
    assert rel_err(a.grad.cpu().numpy(), (2.0 * z["opacity"] * (1 - z["opacity"]))) < 1e-6
    # at size, against ATen on the same device
    g = torch.Generator(device=DEV).manual_seed(5)
    P = 2_000_000
    ro = torch.randn(P, 1, device=DEV, generator=g) * 3
    rs = torch.randn(P, 3, device=DEV, generator=g) * 2 - 3
    rq = torch.randn(P, 4, device=DEV, generator=g)
    gs = [torch.randn(P, n, device=DEV, generator=g) for n in (1, 3, 4)]
    x1 = [t.clone().requires_grad_(True) for t in (ro, rs, rq)]
    x2 = [t.clone().requires_grad_(True) for t in (ro, rs, rq)]
    y1 = activate(*x1)
    y2 = (torch.sigmoid(x2[0]), torch.exp(x2[1]), F.normalize(x2[2]))
    torch.autograd.backward(list(y1), gs)
    torch.autograd.backward(list(y2), gs)
    for u, v in zip(y1, y2):
        assert float((u - v).detach().abs().max()) <= 3e-7 * max(1.0, float(v.detach().abs().max()))
    for name, u, v in zip(("opacity", "scaling"), x1[:2], x2[:2]):
        worst = float(((u.grad - v.grad).abs() / (v.grad.abs() + 1e-30)).max())
        assert worst < 2e-6, (name, worst)
    # quaternion: (g - y <y,g>) / |x| cancels when g is nearly parallel to x, so fp32 results agree to a few ulp of the
    # TERMS (|g| / |x|), not of the result; measured against float64: this kernel must be as close as ATen's fp32 is
    x64 = rq.double().requires_grad_(True)
    F.normalize(x64).backward(gs[2].double())
    scale = gs[2].abs().amax(dim=-1, keepdim=True).double() / rq.double().norm(dim=-1, keepdim=True)
    e_mine = float(((x1[2].grad.double() - x64.grad).abs() / scale).max())
    e_aten = float(((x2[2].grad.double() - x64.grad).abs() / scale).max())
    assert e_mine < 1e-6 and e_mine < 2.0 * e_aten + 2e-7, (e_mine, e_aten)


def test_activation_argument_checks():
    from vegs_amd.instances import activate
    o, s, q = torch.zeros(5, 1, device=DEV), torch.zeros(5, 3, device=DEV), torch.ones(5, 4, device=DEV)
    with pytest.raises(ValueError):
        activate(o, s, q[:, :3])
    with pytest.raises(ValueError):
        activate(o[:4], s, q)
    with pytest.raises(ValueError):
        activate(o.cpu(), s.cpu(), q.cpu())
    e = activate(o[:0], s[:0], q[:0])
    assert e[0].shape == (0, 1) and e[2].shape == (0, 4)


@pytest.mark.parametrize("sh_degree", [3, 2])
def test_raw_static_model_with_instances_equals_activations_in_front(sh_degree):
    """harness.render_all(static_raw=...): the static model's RAW _opacity / _scaling / _rotation in the concatenated
    inputs, activated by the op for the rows in front of the instances (VR_FLAG_RAW_PARAMS + the SH tail's boundary) ==
    vegs_amd.instances.activate in front of prepare_and_merge: images bit-identical, every gradient equal."""
    import numpy as np
    from vegs_amd import harness, instances, iteration, rasterizer, scenes
    dev = torch.device("cuda", 0)
    # (degree 2: M = 9, rows of 27 floats cannot form an SH tail -> the concatenated fallback, activations in front)
    sc, deg = scenes.scene_street(P=30000, length=40.0, sh_degree=sh_degree, seed=12)
    M = sc["shs"].shape[1]
    cam = scenes.kitti_camera(2.0, 0.3, 688, 188)
    cam_t = harness.cam_tensors(cam, dev)
    bg = torch.zeros(3, device=dev)
    rng = np.random.default_rng(2)
    gouts = [torch.tensor(rng.normal(size=s).astype(np.float32), device=dev) for s in [(3, 188, 688), (4, 188, 688), (3, 188, 688)]]

    def run(raw):
        p, _ = iteration.make_model(sc, dev)
        with torch.no_grad():
            p["rotation"].mul_(torch.tensor(rng2.uniform(0.5, 2.0, (p["rotation"].shape[0], 1)).astype(np.float32), device=dev))
        boxes = iteration.make_boxes(3, dev, points=700)
        for b, _ in boxes:
            b["shs"] = b["shs"].detach()[:, :M].contiguous().requires_grad_(True)
        t = {"means3D": p["xyz"], "shs": (p["f_dc"], p["f_rest"]) if M == 16 else torch.cat((p["f_dc"], p["f_rest"]), 1)}
        rawd = {"opacities": p["opacity"], "scales": p["scaling"], "rotations": p["rotation"]}
        with rasterizer.flags(rasterizer.FLAG_DETERMINISTIC):
            if raw:
                pkg = harness.render_all(cam, t, [b for b, _ in boxes], [w for _, w in boxes], deg, bg, cam_t=cam_t, fused=True,
                                         static_raw=rawd)
            else:
                o, s_, r = instances.activate(rawd["opacities"], rawd["scales"], rawd["rotations"])
                pkg = harness.render_all(cam, {**t, "opacities": o, "scales": s_, "rotations": r}, [b for b, _ in boxes],
                                         [w for _, w in boxes], deg, bg, cam_t=cam_t, fused=True)
            torch.autograd.backward([pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]], gouts)
        return pkg, p, boxes

    rng2 = np.random.default_rng(5)
    pa, ma, ba = run(False)
    rng2 = np.random.default_rng(5)
    pb, mb, bb = run(True)
    for k in ("render", "render_depth", "render_cov_quat", "render_cov_scale", "alpha", "radii"):
        assert torch.equal(pa[k], pb[k]), k
    assert int((pb["radii"][30000:] > 0).sum()) > 200                      # the instances are in the frame
    for k in ma:
        ga, gb = ma[k].grad.cpu().numpy(), mb[k].grad.cpu().numpy()
        assert np.abs(ga - gb).max() <= 2e-6 * np.abs(ga).max(), (k, np.abs(ga - gb).max(), np.abs(ga).max())
    for (b1, w1), (b2, w2) in zip(ba, bb):
        assert torch.equal(w1.grad, w2.grad)
        for k in b1:
            assert torch.equal(b1[k].grad, b2[k].grad), k
