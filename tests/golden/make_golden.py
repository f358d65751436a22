"""Generates the golden fixtures under tests/golden/ (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Part A imports the reference's own Python functions from /root/reference (read-only;
nothing of it is copied -- only inputs and outputs are stored):
  * utils/sh_utils.py:57-112 eval_sh            -> ref_sh.npz
  * utils/graphics_utils.py:204-248 quaternion_to_matrix, :266-277 getWorld2View2,
    :305-337 getProjectionMatrixwithPrincipalPointOffset, :342-343 focal2fov -> ref_camera.npz
  * loss/normal_guidance.py:3-22 loss_normal_guidance (value + gradients w.r.t.
    cov_quat / cov_scale)                        -> ref_normal_guidance.npz
  * utils/loss_utils.py:18-22 l1_loss, :39-79 ssim (values + gradients)  -> ref_photometric.npz
  * utils/graphics_utils.py:49-53 decompose_T_to_RS, :140-201 matrix_to_quaternion, :204-248
    quaternion_to_matrix composed as gaussian_renderer/__init__.py:122-153 does -> ref_instances.npz
  * utils/general_utils.py:83-129 build_scaling_rotation / strip_symmetric composed as
    scene/gaussian_model.py:32-36 build_covariance_from_scaling_rotation does (their hard-coded device='cuda'
    is redirected to the CPU for the duration of the call)                         -> ref_cov3d.npz
Part B runs the independent float64 autograd restatement oracle/torch_ref.py on tiny
seeded scenes (vegs_amd/scenes.py) and stores inputs, forward images and input gradients
-> raster_case*.npz.  These pin vr_oracle.c and, on the GPU box, the HIP kernels.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def part_a():
    sys.path.insert(0, REF)
    import utils.graphics_utils as gu
    import utils.sh_utils as shu
    spec = importlib.util.spec_from_file_location("ref_normal_guidance", os.path.join(REF, "loss/normal_guidance.py"))
    ng = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ng)

    rng = np.random.default_rng(100)
    n = 64
    sh = rng.normal(0, 0.5, (n, 3, 16)).astype(np.float32)          # [n, C, K] as eval_sh wants
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    out = {"sh": sh, "dirs": d}
    for deg in range(4):
        out[f"rgb_deg{deg}"] = shu.eval_sh(deg, torch.tensor(sh), torch.tensor(d)).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_sh.npz"), **out)

    # camera matrices at KITTI-360-like intrinsics, built as scene/cameras.py:70-88 does
    from vegs_amd import scenes
    cams = {}
    for tag, (W, H) in {"1408": (1408, 376), "1376": (1376, 376)}.items():
        s = W / 1408
        fx, fy, cx, cy = scenes.KITTI360_FX, scenes.KITTI360_FY, scenes.KITTI360_CX * s, scenes.KITTI360_CY
        R = scenes.R_KITTI
        eye = np.array([3.0, -0.3, 0.1])
        T = -R.T @ eye
        fovx, fovy = gu.focal2fov(fx, W), gu.focal2fov(fy, H)
        wv = torch.tensor(gu.getWorld2View2(R, T)).transpose(0, 1)
        pm = gu.getProjectionMatrixwithPrincipalPointOffset(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy, fx=fx, fy=fy,
                                                            cx=cx, cy=cy, w=W, h=H).transpose(0, 1)
        full = (wv.unsqueeze(0).bmm(pm.unsqueeze(0))).squeeze(0)
        center = wv.inverse()[3, :3]
        cams.update({f"R_{tag}": R, f"T_{tag}": T, f"K_{tag}": np.array([fx, fy, cx, cy, W, H]),
                     f"fov_{tag}": np.array([fovx, fovy]), f"view_{tag}": wv.numpy(), f"full_{tag}": full.numpy(),
                     f"center_{tag}": center.numpy()})
    q = rng.normal(size=(32, 4)).astype(np.float32)
    cams["quat"] = q
    cams["quat_matrix"] = gu.quaternion_to_matrix(torch.tensor(q)).numpy()
    # matrix_to_quaternion (utils/graphics_utils.py:140-201) on proper rotations, incl. near-180-degree ones
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    qn[:4] = np.array([[1e-4, 1, 0, 0], [1e-4, 0, 1, 0], [1e-4, 0, 0, 1], [0.5, 0.5, 0.5, 0.5]], np.float32)
    qn = (qn / np.linalg.norm(qn, axis=1, keepdims=True)).astype(np.float32)
    rm = gu.quaternion_to_matrix(torch.tensor(qn))
    cams["rotmat"] = rm.numpy()
    cams["rotmat_quat"] = gu.matrix_to_quaternion(rm).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_camera.npz"), **cams)

    # normal-guidance loss (the consumer of cov_quat / cov_scale): value and gradients
    H, W = 12, 20

    class Cam:
        pass
    cam = Cam()
    cam.original_normal = torch.tensor(rng.normal(size=(3, H, W)).astype(np.float32))
    cam.R = scenes.R_KITTI.copy()
    cq = torch.tensor(rng.normal(size=(4, H, W)).astype(np.float32), requires_grad=True)
    cs = torch.tensor(rng.uniform(0.01, 0.2, (3, H, W)).astype(np.float32), requires_grad=True)
    loss = ng.loss_normal_guidance(cam, cq, cs)
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "ref_normal_guidance.npz"), normal=cam.original_normal.numpy(), R=cam.R,
                        cov_quat=cq.detach().numpy(), cov_scale=cs.detach().numpy(), loss=loss.item(),
                        grad_cov_quat=cq.grad.numpy(), grad_cov_scale=cs.grad.numpy())

    # photometric losses (utils/loss_utils.py:18-22 l1_loss, :39-79 ssim) exactly as train.py:162-164 combines
    # them: values and d/d(image) by torch autograd, on small images with all border cases
    import utils.loss_utils as lu
    blob = {}
    for tag, (H, W) in {"a": (37, 53), "b": (16, 16), "c": (9, 40)}.items():     # c: image smaller than the window
        img = torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32), requires_grad=True)
        gt = torch.tensor(np.clip(img.detach().numpy() + rng.normal(0, 0.15, (3, H, W)), 0, 1).astype(np.float32))
        gt[:, :3, :5] = img.detach()[:, :3, :5]                                   # exact zeros of |x-y|
        l1 = lu.l1_loss(img, gt)
        ss = lu.ssim(img, gt)
        g_l1, = torch.autograd.grad(l1, img, retain_graph=True)
        g_ss, = torch.autograd.grad(ss, img, retain_graph=True)
        lam = 0.2                                                                 # arguments/__init__.py:90
        loss = (1.0 - lam) * l1 + lam * (1.0 - ss)
        g_loss, = torch.autograd.grad(loss, img)
        blob.update({f"img_{tag}": img.detach().numpy(), f"gt_{tag}": gt.numpy(), f"l1_{tag}": l1.item(),
                     f"ssim_{tag}": ss.item(), f"grad_l1_{tag}": g_l1.numpy(), f"grad_ssim_{tag}": g_ss.numpy(),
                     f"loss_{tag}": loss.item(), f"grad_loss_{tag}": g_loss.numpy()})
    blob["window"] = lu.gaussian(11, 1.5).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_photometric.npz"), **blob)

    # box-instance branch of prepare_rasterization (gaussian_renderer/__init__.py:122-126,140-153), composed from
    # the reference's own decompose_T_to_RS / quaternion_to_matrix / matrix_to_quaternion; values + autograd
    # gradients w.r.t. the instance's Gaussians and the 4x4 box2world
    blob = {}
    for b in range(4):
        n = 60
        ang = rng.normal(size=4)
        ang = ang / np.linalg.norm(ang)
        if b == 1:
            ang = np.array([1e-3, 0.0, 1.0, 0.0])                      # ~180 degrees about y: another candidate wins
        Rb = gu.quaternion_to_matrix(torch.tensor(ang)).numpy()
        sc3 = np.array([1.7, 1.7, 1.7]) if b % 2 == 0 else np.array([0.6, 1.1, 2.3])   # uniform / per-axis scale
        B = np.eye(4)
        B[:3, :3] = Rb * sc3[None, :]
        B[:3, 3] = rng.normal(size=3) * 3
        box2world = torch.tensor(B.astype(np.float32), requires_grad=True)
        xyz = torch.tensor(rng.normal(size=(n, 3)).astype(np.float32), requires_grad=True)
        scales = torch.tensor(rng.uniform(0.01, 0.3, (n, 3)).astype(np.float32), requires_grad=True)
        q = rng.normal(size=(n, 4))
        q[:8] = np.eye(4)[[0, 1, 2, 3, 0, 1, 2, 3]] + 1e-3 * rng.normal(size=(8, 4))       # axis-aligned half-turns
        rot = torch.tensor((q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32), requires_grad=True)
        means3D = torch.cat((xyz, torch.ones(n, 1)), dim=1)
        means3D = torch.matmul(box2world, means3D.transpose(1, 0).contiguous()).transpose(1, 0).contiguous()
        means3D = means3D[:, :3] / means3D[:, 3:]
        b_scale, b_rot = gu.decompose_T_to_RS(box2world)
        rotations = gu.matrix_to_quaternion(torch.matmul(b_rot[None, ...], gu.quaternion_to_matrix(rot)))
        scales_o = scales * b_scale
        gm, gr, gsc = (torch.tensor(rng.normal(size=t.shape).astype(np.float32)) for t in (means3D, rotations, scales_o))
        ((means3D * gm).sum() + (rotations * gr).sum() + (scales_o * gsc).sum()).backward()
        blob.update({f"box2world_{b}": box2world.detach().numpy(), f"xyz_{b}": xyz.detach().numpy(),
                     f"scales_{b}": scales.detach().numpy(), f"rot_{b}": rot.detach().numpy(),
                     f"out_means_{b}": means3D.detach().numpy(), f"out_rot_{b}": rotations.detach().numpy(),
                     f"out_scales_{b}": scales_o.detach().numpy(), f"gout_means_{b}": gm.numpy(),
                     f"gout_rot_{b}": gr.numpy(), f"gout_scales_{b}": gsc.numpy(),
                     f"grad_xyz_{b}": xyz.grad.numpy(), f"grad_scales_{b}": scales.grad.numpy(),
                     f"grad_rot_{b}": rot.grad.numpy(), f"grad_box2world_{b}": box2world.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, "ref_instances.npz"), **blob)

    # cov3D: the in-repo definition the rasterizer's scale/rotation path must reproduce (scene/gaussian_model.py:32-36
    # = strip_symmetric(L L^T), L = build_scaling_rotation(modifier * scaling, rotation), utils/general_utils.py:83-129).
    # Those helpers allocate with device='cuda'; there is no GPU in the build container, so torch.zeros is wrapped to
    # ignore the device while they run -- the arithmetic is the reference's own.
    import types
    if "torchvision" not in sys.modules:          # general_utils imports torchvision for an image helper only; absent here
        tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
        tvt.functional = types.ModuleType("torchvision.transforms.functional")
        tv.transforms = tvt
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                            "torchvision.transforms.functional": tvt.functional})
    import utils.general_utils as gen
    real_zeros = torch.zeros

    def zeros_cpu(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)
    n = 200
    s = np.exp(rng.normal(np.log(0.05), 1.0, (n, 3))).astype(np.float32)
    s[:20, 0] = 1e-5                                                  # VEGS discs (utils/norminit_utils.py:217-219)
    q = rng.normal(size=(n, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    blob = {"scales": s, "rotations": q}
    torch.zeros = zeros_cpu
    try:
        for mod in (1.0, 0.37):
            L = gen.build_scaling_rotation(mod * torch.tensor(s), torch.tensor(q))
            blob[f"cov6_mod{mod}"] = gen.strip_symmetric(L @ L.transpose(1, 2)).numpy()
    finally:
        torch.zeros = real_zeros
    np.savez_compressed(os.path.join(HERE, "ref_cov3d.npz"), **blob)


CASES = {
    # name: (scene kwargs, camera (W,H), sh_degree, bg, mode)
    "case_sh3": dict(P=300, seed=11, scale=0.05, W=64, H=48, deg=3, bg=(0.1, 0.2, 0.3), mode="sh_sr", mod=1.0),
    "case_precomp": dict(P=250, seed=12, scale=0.04, W=48, H=48, deg=0, bg=(0.0, 0.0, 0.0), mode="precomp", mod=1.0),
    "case_cull_deg1": dict(P=400, seed=13, scale=0.08, W=80, H=40, deg=1, bg=(1.0, 1.0, 1.0), mode="sh_sr", mod=1.3,
                           extent=2.5, opaque=True),
    # the fork assumptions as switches (include/vegs_rast.h VrFlags): one case per flag + all together
    "case_flag_scale": dict(P=260, seed=21, scale=0.05, W=64, H=48, deg=2, bg=(0.1, 0.2, 0.3), mode="sh_sr", mod=1.4, flags=1),
    "case_flag_depthnorm": dict(P=260, seed=22, scale=0.05, W=64, H=48, deg=1, bg=(0.3, 0.1, 0.0), mode="sh_sr", mod=1.0, flags=2),
    "case_flag_noalpha": dict(P=260, seed=23, scale=0.05, W=64, H=48, deg=1, bg=(0.0, 0.2, 0.1), mode="sh_sr", mod=1.0, flags=4),
    "case_flag_fill": dict(P=150, seed=24, scale=0.04, W=64, H=48, deg=0, bg=(0.2, 0.2, 0.2), mode="sh_sr", mod=1.0, flags=8),
    "case_flag_all": dict(P=260, seed=25, scale=0.05, W=64, H=48, deg=3, bg=(0.5, 0.4, 0.3), mode="sh_sr", mod=0.8, flags=15),
    "case_flag_dnorm_fill": dict(P=200, seed=26, scale=0.05, W=48, H=48, deg=1, bg=(0.0, 0.0, 0.0), mode="sh_sr", mod=1.2, flags=10),
    # ---- round 6: cases that cross what the segmented kernels are built around (round-5 verdict: every case above has
    # tile lists of at most 159 entries -- one 256-entry segment -- and rectangles of at most a few tiles)
    # DEEP TILES: 6,000 translucent Gaussians on 24 tiles: lists of up to ~2,000 entries (8 segments), n_contrib up to
    # ~1,300, ~3,900 pixels that STOP (T < 1e-4) in a later segment: segment carries, late stops, the backward's suffix sums
    "case_deep_tiles": dict(P=6000, seed=41, scale=0.06, W=96, H=64, deg=2, bg=(0.1, 0.2, 0.3), mode="sh_sr", mod=1.0,
                            opmean=-2.0, gout16=True),
    # HUGE RECTANGLES: a 12 x 10-tile frame with splats whose rectangles cover more than 64 tiles (cell masks, the whole-wave /
    # listed emission), some of 9 ... 64 tiles, among small ones
    "case_huge_rect": dict(P=500, seed=42, scale=0.03, W=192, H=160, deg=1, bg=(0.0, 0.1, 0.0), mode="sh_sr", mod=1.0,
                           giants=((0, 30.0), (1, 18.0), (2, 45.0), (3, 9.0), (4, 6.0), (5, 7.5), (6, 12.0), (7, 5.0)), gout16=True),
    # the reference's frame (gaussian_renderer/__init__.py:109: torch.Size([3, 376, 1408])) with the KITTI-360 principal
    # point offset: a street of discs rendered at 1408 x 376; the fixture keeps a CROP of the images (upstream gradients are
    # zero outside it), so the file stays small while the tile grid (88 x 24), the projection and the lists are the frame's
    "case_kitti_crop": dict(P=3000, seed=43, W=1408, H=376, deg=3, bg=(0.0, 0.0, 0.0), mode="sh_sr", mod=1.0,
                            scene="street", length=40.0, cam="kitti", crop=(180, 276, 600, 792), gout16=True),
}


def build_case(c):
    from vegs_amd import scenes
    if c.get("scene") == "street":
        sc, deg = scenes.scene_street(P=c["P"], length=c["length"], sh_degree=c["deg"], seed=c["seed"])
    else:
        sc, deg = scenes.scene_random(P=c["P"], sh_degree=c["deg"], seed=c["seed"], scale=c["scale"],
                                      extent=c.get("extent", 0.5))
    if c.get("opmean") is not None:      # translucent: many splats per pixel before it saturates
        rng = np.random.default_rng(c["seed"] + 1000)
        sc["opacities"] = (1 / (1 + np.exp(-rng.normal(c["opmean"], 1.0, sc["opacities"].shape)))).astype(np.float32)
    for i, k in c.get("giants", ()):     # a few splats blown up: rectangles of tens to hundreds of tiles
        sc["scales"][i] = (sc["scales"][i] * k).astype(np.float32)
        sc["opacities"][i] = 0.35
    if c.get("opaque"):
        rng = np.random.default_rng(c["seed"] + 1000)
        sc["opacities"] = rng.uniform(0.5, 1.0, sc["opacities"].shape).astype(np.float32)
        sc["rotations"] = (sc["rotations"] * rng.uniform(0.8, 1.2, (c["P"], 1))).astype(np.float32)  # un-normalised q
    cam = scenes.kitti_camera(2.0, 0.3, c["W"], c["H"]) if c.get("cam") == "kitti" else scenes.camera_c1(c["W"], c["H"])
    return sc, deg, cam


def part_b(only_new=False):
    from oracle import torch_ref
    for name, c in CASES.items():
        if only_new and os.path.exists(os.path.join(HERE, f"raster_{name}.npz")):
            continue
        sc, deg, cam = build_case(c)
        P = c["P"]
        T = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sc.items()}
        m2d = torch.zeros(P, 3, dtype=torch.float64, requires_grad=True)
        kw = dict(H=cam.image_height, W=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                  bg=torch.tensor(c["bg"], dtype=torch.float64), scale_modifier=c["mod"],
                  viewmatrix=torch.tensor(cam.world_view_transform), projmatrix=torch.tensor(cam.full_proj_transform),
                  campos=torch.tensor(cam.camera_center), sh_degree=deg)
        extra = {}
        if c["mode"] == "precomp":
            with torch.no_grad():
                cov = torch_ref.build_cov3d(T["scales"], c["mod"], T["rotations"])
                cov6 = torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)
                col = torch.rand(P, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(c["seed"]))
            # round through float32 so stored inputs are exactly what was rendered
            cov6 = cov6.float().double().requires_grad_(True)
            col = col.float().double().requires_grad_(True)
            res = torch_ref.rasterize(T["means3D"], None, col, T["opacities"], None, None, cov6, means2D=m2d, **kw)
            extra = {"in_colors_precomp": col, "in_cov3D_precomp": cov6}
        else:
            res = torch_ref.rasterize(T["means3D"], T["shs"], None, T["opacities"], T["scales"], T["rotations"], None,
                                      means2D=m2d, flags=c.get("flags", 0), **kw)
        rng = np.random.default_rng(c["seed"] + 7)
        names = ["color", "depth", "cov_quat", "cov_scale", "alpha"]
        gouts = [rng.normal(size=tuple(r.shape)).astype(np.float32) for r in res[:5]]
        if c.get("gout16"):              # values a float16 holds: the stored arrays compress to half
            gouts = [g.astype(np.float16).astype(np.float32) for g in gouts]
        crop = c.get("crop")             # (y0, y1, x0, x1): upstream gradients vanish outside, only the crop is stored
        if crop:
            for g in gouts:
                keep = g[:, crop[0]:crop[1], crop[2]:crop[3]].copy()
                g[...] = 0.0
                g[:, crop[0]:crop[1], crop[2]:crop[3]] = keep
        loss = sum((r * torch.tensor(g, dtype=torch.float64)).sum() for r, g in zip(res[:5], gouts))
        loss.backward()
        blob = {"meta": np.array([P, cam.image_width, cam.image_height, deg], np.int64),
                "bg": np.array(c["bg"], np.float32), "scale_modifier": np.float32(c["mod"]),
                "tanfov": np.array([cam.tanfovx, cam.tanfovy], np.float64),
                "viewmatrix": cam.world_view_transform, "projmatrix": cam.full_proj_transform,
                "campos": cam.camera_center, "radii": res[5].numpy(), "flags": np.int64(c.get("flags", 0))}
        for k, v in sc.items():
            blob["in_" + k] = v
        for k, v in extra.items():
            blob[k] = v.detach().numpy().astype(np.float32)
        if crop:
            blob["crop"] = np.array(crop, np.int64)
        for n, r, g in zip(names, res[:5], gouts):
            o = r.detach().numpy().astype(np.float32)
            blob["out_" + n] = o[:, crop[0]:crop[1], crop[2]:crop[3]] if crop else o
            blob["gout_" + n] = g[:, crop[0]:crop[1], crop[2]:crop[3]] if crop else g
        blob["grad_means2D"] = m2d.grad.numpy().astype(np.float32)
        blob["grad_means3D"] = T["means3D"].grad.numpy().astype(np.float32)
        blob["grad_opacities"] = T["opacities"].grad.numpy().astype(np.float32)
        if c["mode"] == "precomp":
            blob["grad_colors_precomp"] = extra["in_colors_precomp"].grad.numpy().astype(np.float32)
            blob["grad_cov3D_precomp"] = extra["in_cov3D_precomp"].grad.numpy().astype(np.float32)
        else:
            blob["grad_shs"] = T["shs"].grad.numpy().astype(np.float32)
            blob["grad_scales"] = T["scales"].grad.numpy().astype(np.float32)
            blob["grad_rotations"] = T["rotations"].grad.numpy().astype(np.float32)
        np.savez_compressed(os.path.join(HERE, f"raster_{name}.npz"), **blob)
        print(name, "visible", int((res[5] > 0).sum()), "of", P)


def part_c():
    """ref_activations.npz: the model's activations as the reference assigns them (scene/gaussian_model.py:37-45:
    scaling_activation = torch.exp, opacity_activation = torch.sigmoid, rotation_activation =
    torch.nn.functional.normalize; read through get_scaling / get_opacity / get_rotation, :98-120), float32 on the CPU,
    with autograd's gradients for random upstream gradients.  The assignments are checked in the reference's source
    before the functions are called here."""
    src = open(os.path.join(REF, "scene", "gaussian_model.py")).read()
    for line in ("self.scaling_activation = torch.exp", "self.opacity_activation = torch.sigmoid",
                 "self.rotation_activation = torch.nn.functional.normalize"):
        assert line in src, line
    rng = np.random.default_rng(77)
    n = 512
    o = rng.normal(0, 3, (n, 1)).astype(np.float32)
    o[:4, 0] = [30.0, -30.0, 0.0, -88.0]
    sc = rng.normal(-3, 2, (n, 3)).astype(np.float32)
    sc[0] = [-11.5, 3.0, 0.0]
    q = rng.normal(size=(n, 4)).astype(np.float32)
    q[0] = 0.0                       # |q| below F.normalize's eps: divided by eps, gradient g / eps
    q[1] = [1e-13, 0, 0, 0]
    q[2] = [3e-7, -2e-7, 1e-7, 0]
    q[3] = [1e4, -2e4, 5e3, 1.0]
    to, ts, tq = (torch.tensor(a, requires_grad=True) for a in (o, sc, q))
    yo, ys, yq = torch.sigmoid(to), torch.exp(ts), torch.nn.functional.normalize(tq)
    go, gs, gq = (torch.tensor(rng.normal(size=a.shape).astype(np.float32)) for a in (o, sc, q))
    torch.autograd.backward([yo, ys, yq], [go, gs, gq])
    np.savez_compressed(os.path.join(HERE, "ref_activations.npz"), raw_opacity=o, raw_scaling=sc, raw_rotation=q,
                        opacity=yo.detach().numpy(), scales=ys.detach().numpy(), rotations=yq.detach().numpy(),
                        g_opacity=go.numpy(), g_scales=gs.numpy(), g_rotations=gq.numpy(), d_opacity=to.grad.numpy(),
                        d_scaling=ts.grad.numpy(), d_rotation=tq.grad.numpy())


def part_d():
    """ref_densify.npz: scene/gaussian_model.py:263-413 (densify_and_prune with its clone / split / prune and the
    optimizer-state surgery, reset_opacity) run by the reference's OWN GaussianModel methods on the CPU.  What had to be
    arranged for that (none of it arithmetic): the module's imports that are absent here (plyfile, the decoder package)
    are empty stand-ins, torch.zeros ignores device="cuda", and torch.normal -- the split's random draw,
    scene/gaussian_model.py:367 -- returns mean + std * noise with the unit-normal `noise` of the fixture, so that the
    draw is an INPUT of the case (torch.normal(mean, std) is defined as that with noise ~ N(0, 1))."""
    import types
    sys.path.insert(0, REF)
    for name, attrs in (("plyfile", ("PlyData", "PlyElement")), ("model", ("GaussianDecoder",))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, type(a, (), {}))
            sys.modules[name] = m
    if "torchvision" not in sys.modules:
        tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
        tvt.functional = types.ModuleType("torchvision.transforms.functional")
        tv.transforms = tvt
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                            "torchvision.transforms.functional": tvt.functional})
    knn = types.ModuleType("simple_knn._C")
    knn.distCUDA2 = None
    saved_knn = {k: sys.modules.get(k) for k in ("simple_knn", "simple_knn._C")}
    sys.modules["simple_knn"] = types.ModuleType("simple_knn")
    sys.modules["simple_knn._C"] = knn
    try:
        # the file itself, not the `scene` package (whose __init__ pulls in the dataset readers and their dependencies)
        spec = importlib.util.spec_from_file_location("ref_gaussian_model", os.path.join(REF, "scene", "gaussian_model.py"))
        gm = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gm)
        GaussianModel = gm.GaussianModel
    finally:
        for k, v in saved_knn.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    real_zeros, real_normal = torch.zeros, torch.normal

    def zeros_cpu(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)

    blob = {}
    cases = {"a": dict(P=1200, seed=5, max_grad=0.0002, min_opacity=0.005, extent=5.0, size_threshold=20, deg=3),
             "b": dict(P=700, seed=6, max_grad=0.0004, min_opacity=0.02, extent=3.0, size_threshold=None, deg=2),
             "c": dict(P=300, seed=7, max_grad=1e9, min_opacity=0.005, extent=4.0, size_threshold=20, deg=1)}  # nothing selected
    for tag, c in cases.items():
        rng = np.random.default_rng(c["seed"])
        P, M = c["P"], (c["deg"] + 1) ** 2
        g = GaussianModel(c["deg"])
        par = {"xyz": rng.normal(0, 2, (P, 3)), "f_dc": rng.normal(0, 1, (P, 1, 3)), "f_rest": rng.normal(0, .2, (P, M - 1, 3)),
               "opacity": rng.normal(-1.0, 3.0, (P, 1)),
               "scaling": np.log(np.exp(rng.normal(np.log(0.03), 1.2, (P, 3)))), "rotation": rng.normal(size=(P, 4))}
        par = {k: v.astype(np.float32) for k, v in par.items()}
        g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation = (
            torch.nn.Parameter(torch.tensor(par[k])) for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"))
        args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                                     position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=2.5e-3,
                                     opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3)
        g.spatial_lr_scale = 1.0
        noise_used = []

        def normal_from_noise(mean=None, std=None, **k):
            n = torch.tensor(rng.normal(size=tuple(std.shape)).astype(np.float32))
            noise_used.append(n.numpy().copy())
            return mean + std * n
        torch.zeros = zeros_cpu
        torch.normal = normal_from_noise
        try:
            g.training_setup(args)
            for _ in range(3):                     # the moments the surgery has to carry along
                for grp in g.optimizer.param_groups:
                    q = grp["params"][0]
                    q.grad = torch.tensor(rng.normal(0, 1e-3, tuple(q.shape)).astype(np.float32))
                g.optimizer.step()
            names = [grp["name"] for grp in g.optimizer.param_groups]
            for grp in g.optimizer.param_groups:
                st = g.optimizer.state[grp["params"][0]]
                blob[f"{tag}_in_{grp['name']}"] = grp["params"][0].detach().numpy().copy()
                blob[f"{tag}_in_m_{grp['name']}"] = st["exp_avg"].numpy().copy()
                blob[f"{tag}_in_v_{grp['name']}"] = st["exp_avg_sq"].numpy().copy()
            den = rng.integers(0, 40, (P, 1)).astype(np.float32)
            den[rng.random(P) < 0.1] = 0.0        # never visible: 0 / 0 = NaN -> 0 (scene/gaussian_model.py:391-392)
            acc = (den * np.exp(rng.normal(np.log(c["max_grad"] if c["max_grad"] < 1 else 2e-4), 1.0, (P, 1)))).astype(np.float32)
            g.xyz_gradient_accum, g.denom = torch.tensor(acc), torch.tensor(den)
            g.max_radii2D = torch.tensor(rng.uniform(0, 60, P).astype(np.float32))
            blob[f"{tag}_in_accum"], blob[f"{tag}_in_denom"], blob[f"{tag}_in_max_radii2D"] = acc, den, g.max_radii2D.numpy().copy()
            g.densify_and_prune(c["max_grad"], c["min_opacity"], c["extent"], c["size_threshold"])
            for grp in g.optimizer.param_groups:
                st = g.optimizer.state[grp["params"][0]]
                assert grp["params"][0] is {"xyz": g._xyz, "f_dc": g._features_dc, "f_rest": g._features_rest,
                                             "opacity": g._opacity, "scaling": g._scaling, "rotation": g._rotation}[grp["name"]]
                blob[f"{tag}_out_{grp['name']}"] = grp["params"][0].detach().numpy().copy()
                blob[f"{tag}_out_m_{grp['name']}"] = st["exp_avg"].numpy().copy()
                blob[f"{tag}_out_v_{grp['name']}"] = st["exp_avg_sq"].numpy().copy()
                blob[f"{tag}_out_step_{grp['name']}"] = np.float32(float(st["step"]))
            blob[f"{tag}_out_accum"], blob[f"{tag}_out_denom"] = g.xyz_gradient_accum.numpy().copy(), g.denom.numpy().copy()
            blob[f"{tag}_out_max_radii2D"] = g.max_radii2D.numpy().copy()
            assert len(noise_used) == 1
            blob[f"{tag}_noise"] = noise_used[0]
            blob[f"{tag}_settings"] = np.array([c["max_grad"], c["min_opacity"], c["extent"], 0.01,
                                                1.0 if c["size_threshold"] else 0.0], dtype=np.float64)
            # reset_opacity on the densified model (scene/gaussian_model.py:215-218)
            g.reset_opacity()
            st = g.optimizer.state[g._opacity]
            blob[f"{tag}_reset_opacity"] = g._opacity.detach().numpy().copy()
            assert float(st["exp_avg"].abs().max()) == 0.0 and float(st["exp_avg_sq"].abs().max()) == 0.0
            print(tag, "P", P, "->", g._xyz.shape[0], "split draws", noise_used[0].shape[0])
        finally:
            torch.zeros, torch.normal = real_zeros, real_normal
    np.savez_compressed(os.path.join(HERE, "ref_densify.npz"), **blob)


def part_e():
    """ref_boxmodel.npz: the reference's OWN BoxModel class (model/boxmodel.py:4-57) on the CPU -- adjustbox2world() and its
    autograd gradients w.r.t. delta_r / delta_s / delta_t for random upstream gradients, then three rounds of what
    train.py:270-274 does with it (optimizer.step(); zero_grad(); regularize(iteration)) from given gradients, the deltas
    recorded after every round.  Arranged for the CPU (none of it arithmetic): the file is loaded on its own (the `model`
    package's __init__ pulls in the StyleGAN decoder), torch.tensor / torch.eye ignore device='cuda' and Tensor.cuda()
    returns the tensor."""
    import types
    sys.path.insert(0, REF)
    import utils.graphics_utils  # noqa: F401  (quaternion_to_matrix: the reference's own)
    spec = importlib.util.spec_from_file_location("ref_boxmodel", os.path.join(REF, "model", "boxmodel.py"))
    bm_mod = importlib.util.module_from_spec(spec)
    real_tensor, real_cuda = torch.tensor, torch.Tensor.cuda

    def tensor_cpu(*a, **k):
        k.pop("device", None)
        return real_tensor(*a, **k)
    torch.tensor = tensor_cpu
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        spec.loader.exec_module(bm_mod)
        rng = np.random.default_rng(41)
        args = types.SimpleNamespace(boxmodel_lr=0.005, boxmodel_lambda_reg=0.001)
        n = 6
        blob = {"lr": np.float64(args.boxmodel_lr), "lambda_reg": np.float64(args.boxmodel_lambda_reg)}
        base, d_r, d_s, d_t, adj, G, g_r, g_s, g_t = ([] for _ in range(9))
        rounds = {k: [] for k in ("step_g_r", "step_g_s", "step_g_t", "after_r", "after_s", "after_t")}
        for i in range(n):
            ang = rng.uniform(0, 6.28)
            Rm = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
            obj = types.SimpleNamespace(R=Rm * rng.uniform(0.8, 2.0), T=rng.normal(0, 10, 3))
            bm = bm_mod.BoxModel(obj, args)
            if i > 0:                                   # instance 0 stays at the identity: the norms' 0/0 case
                with torch.no_grad():
                    bm.delta_r += torch.tensor(rng.normal(0, 0.2, 4).astype(np.float32))
                    bm.delta_s += torch.tensor(rng.normal(0, 0.1, 3).astype(np.float32))
                    bm.delta_t += torch.tensor(rng.normal(0, 0.3, 3).astype(np.float32))
            base.append(bm.box2world.numpy().copy())
            d_r.append(bm.delta_r.detach().numpy().copy()); d_s.append(bm.delta_s.detach().numpy().copy())
            d_t.append(bm.delta_t.detach().numpy().copy())
            a = bm.adjustbox2world()
            g = torch.tensor(rng.normal(size=(4, 4)).astype(np.float32))
            a.backward(g)
            adj.append(a.detach().numpy().copy()); G.append(g.numpy().copy())
            g_r.append(bm.delta_r.grad.numpy().copy()); g_s.append(bm.delta_s.grad.numpy().copy())
            g_t.append(bm.delta_t.grad.numpy().copy())
            bm.optimizer.zero_grad()
            per = {k: [] for k in rounds}
            for it in range(3):                         # train.py:270-274
                gr, gs, gt = (rng.normal(0, 1e-2, k).astype(np.float32) for k in (4, 3, 3))
                bm.delta_r.grad, bm.delta_s.grad, bm.delta_t.grad = torch.tensor(gr), torch.tensor(gs), torch.tensor(gt)
                bm.optimizer.step()
                bm.optimizer.zero_grad()
                bm.regularize(it + 1)
                per["step_g_r"].append(gr); per["step_g_s"].append(gs); per["step_g_t"].append(gt)
                per["after_r"].append(bm.delta_r.detach().numpy().copy())
                per["after_s"].append(bm.delta_s.detach().numpy().copy())
                per["after_t"].append(bm.delta_t.detach().numpy().copy())
            for k in rounds:
                rounds[k].append(np.stack(per[k]))
        for name, v in (("box2world", base), ("delta_r", d_r), ("delta_s", d_s), ("delta_t", d_t), ("adjusted", adj),
                        ("g_adjusted", G), ("grad_delta_r", g_r), ("grad_delta_s", g_s), ("grad_delta_t", g_t)):
            blob[name] = np.stack(v).astype(np.float32)
        for k, v in rounds.items():
            blob[k] = np.stack(v).astype(np.float32)            # [instance, round, k]
        np.savez_compressed(os.path.join(HERE, "ref_boxmodel.npz"), **blob)
        print("ref_boxmodel.npz:", n, "instances")
    finally:
        torch.tensor, torch.Tensor.cuda = real_tensor, real_cuda


if __name__ == "__main__":
    if "--densify" in sys.argv:
        part_d()
        sys.exit(0)
    if "--activations" in sys.argv:
        part_c()
        sys.exit(0)
    if "--boxmodel" in sys.argv:
        part_e()
        sys.exit(0)
    # --new: keep the committed fixtures (their random draws are part of the pins) and only add missing ones
    new = "--new" in sys.argv
    if not new or not os.path.exists(os.path.join(HERE, "ref_cov3d.npz")):
        part_a()
    part_b(only_new=new)
    if not new or not os.path.exists(os.path.join(HERE, "ref_activations.npz")):
        part_c()
    if not new or not os.path.exists(os.path.join(HERE, "ref_densify.npz")):
        part_d()
    if not new or not os.path.exists(os.path.join(HERE, "ref_boxmodel.npz")):
        part_e()
