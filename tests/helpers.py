"""Shared helpers for the parity tests (test infrastructure; may use oracle/)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OUT_NAMES = ["color", "depth", "cov_quat", "cov_scale", "alpha"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"raster_{name}.npz"))
    return {k: z[k] for k in z.files}


FLAG_CASES = ["case_flag_scale", "case_flag_depthnorm", "case_flag_noalpha", "case_flag_fill", "case_flag_all",
              "case_flag_dnorm_fill"]


# round 6: cases whose tile lists span several 256-entry segments (stops in later segments), whose rectangles exceed 64
# tiles, and the reference's 1408 x 376 frame (a crop of its images): tests/golden/make_golden.py CASES
DEEP_CASES = ["case_deep_tiles", "case_huge_rect", "case_kitti_crop"]
FLAG_FULL_TILE_LISTS = 32768     # include/vegs_rast.h VR_FLAG_FULL_TILE_LISTS: the reference's lists (every tile of the rectangle)


def assert_images_close(c, name, out, what=""):
    """Forward images against a float64 golden: 1e-5 (x the channel's magnitude when that exceeds 1) -- except on the
    KITTI-shaped street, whose 1e-5-thin discs seen edge-on amplify the fp32 rounding of the projection (conic
    conditioning in the hundreds): there the bar is the north star's 1e-4 with at most 2 % of the pixels beyond 1e-5."""
    street = name == "case_kitti_crop"
    for n in OUT_NAMES:
        scale = max(1.0, np.abs(c["out_" + n]).max())
        d = np.abs(case_view(c, out[n]) - c["out_" + n])
        assert d.max() < (1e-4 if street else 1e-5) * scale, (what, n, d.max())
        if street:
            assert (d > 1e-5 * scale).mean() < 0.02, (what, n, (d > 1e-5 * scale).mean())


def case_gouts(c):
    """Upstream gradients of the five images at full frame size (a cropped fixture holds them inside its crop only:
    they are zero outside)."""
    P, W, H, deg = (int(v) for v in c["meta"])
    gouts = []
    for n in OUT_NAMES:
        g = c["gout_" + n]
        if "crop" in c:
            y0, y1, x0, x1 = (int(v) for v in c["crop"])
            full = np.zeros((g.shape[0], H, W), np.float32)
            full[:, y0:y1, x0:x1] = g
            g = full
        gouts.append(g)
    return gouts


def case_view(c, img):
    """The part of a full-frame image the fixture stores."""
    if "crop" not in c:
        return img
    y0, y1, x0, x1 = (int(v) for v in c["crop"])
    return img[:, y0:y1, x0:x1]


def case_flags(c):
    """VrFlags (include/vegs_rast.h) the golden case was rendered with (0 for the cases of round 1)."""
    return int(c["flags"]) if "flags" in c else 0


def case_inputs(c):
    """The 7 tensor kwargs of the op (numpy), in the reference's naming."""
    pre = "in_colors_precomp" in c
    return dict(
        means3D=c["in_means3D"],
        shs=None if pre else c["in_shs"],
        colors_precomp=c["in_colors_precomp"] if pre else None,
        opacities=c["in_opacities"],
        scales=None if pre else c["in_scales"],
        rotations=None if pre else c["in_rotations"],
        cov3D_precomp=c["in_cov3D_precomp"] if pre else None,
    )


def oracle_cam_from_case(c):
    from oracle import oracle as orc
    P, W, H, deg = (int(v) for v in c["meta"])
    return orc.make_cam(H, W, c["tanfov"][0], c["tanfov"][1], c["bg"], float(c["scale_modifier"]),
                        c["viewmatrix"], c["projmatrix"], c["campos"], deg, 16, flags=case_flags(c))


def oracle_cam(cam, bg, sh_degree, scale_modifier=1.0, M=16, flags=0):
    from oracle import oracle as orc
    return orc.make_cam(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, bg, scale_modifier,
                        cam.world_view_transform, cam.full_proj_transform, cam.camera_center, sh_degree, M, flags=flags)


def rel_err(a, b):
    """max |a-b| / max(|b|, tiny): tensor-level relative error.  Kept for quantities that have no per-Gaussian
    rows (images, optimizer state); GRADIENTS are compared with assert_grad_close below."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def grad_mismatch(got, want, rtol=1e-3, floor=1e-6):
    """Per-row (= per-Gaussian) gradient comparison.  Row i passes when every element satisfies
        |got - want| <= rtol * max|want[i]| + floor * max|want|
    i.e. the error is measured against the size of THAT Gaussian's gradient (a Gaussian whose gradient is 1e-4 of
    the tensor maximum cannot hide a 100 % error under a tensor-level norm), with a floor tied to the tensor
    maximum for rows that are pure cancellation noise (fp32 sums of terms ~1e-6 * max).  Returns (failing row
    indices, per-row error in units of the allowance)."""
    g = np.asarray(got, np.float64)
    w = np.asarray(want, np.float64)
    assert g.shape == w.shape, (g.shape, w.shape)
    if g.size == 0:
        return np.zeros(0, np.int64), np.zeros(0)
    g2 = g.reshape(g.shape[0], -1) if g.ndim > 1 else g.reshape(-1, 1)
    w2 = w.reshape(g2.shape)
    rowmax = np.abs(w2).max(axis=1)
    allow = rtol * rowmax + floor * max(np.abs(w2).max(), 1e-30)
    err = np.abs(g2 - w2).max(axis=1)
    bad = ~np.isfinite(g2).all(axis=1) | (err > allow)
    return np.nonzero(bad)[0], err / allow


def assert_grad_close(name, got, want, rtol=1e-3, floor=1e-6, outliers=1e-4, near=1e-3, cap=10.0, explain=None, ill=None,
                      ill_quota=None):
    """Assert the per-row criterion of grad_mismatch.  fp32 atomics are order-dependent and the preprocess backward
    amplifies the noise of rows that are pure cancellation (1e-5-thin discs), so a few rows may leave the allowance:
    at most a fraction `near` of the rows by up to 3x, at most a fraction `outliers` (0.01 %) by more than that, and
    none by `cap` (10x).  The offenders are printed so that a systematic error is visible in the log; `explain(i)`
    (optional) returns a string printed next to offender i (e.g. the conditioning of its conic).  `ill` (optional bool
    per row): rows the caller has shown to be ill-conditioned (helpers.ill_conditioned) are exempt from `cap` -- they
    still count towards `near` and `outliers`, unless `ill_quota` is given: then they are held to that many allowances
    instead (a number, or one per row: e.g. a multiple of each row's MEASURED sensitivity) and leave the quotas to the
    well-conditioned rows.
    Returns the indices of the failing rows."""
    bad, ratio = grad_mismatch(got, want, rtol, floor)
    n = max(int(np.asarray(want).shape[0]) if np.asarray(want).ndim else 1, 1)
    if len(bad):
        w = np.asarray(want).reshape(n, -1)
        g = np.asarray(got).reshape(n, -1)
        worst = bad[np.argsort(-ratio[bad])][:8]
        print(f"[grad] {name}: {len(bad)} / {n} rows outside rtol={rtol} floor={floor}; worst rows:")
        for i in worst:
            print(f"    row {i}: err/allow {ratio[i]:.2f}  got {g[i][:4]}  want {w[i][:4]}" +
                  (f"  [{explain(int(i))}]" if explain else ""))
        assert np.isfinite(ratio[bad]).all(), f"{name}: non-finite gradient rows"
        if ill is not None and ill_quota is not None:
            # rows the caller has shown to be ill-conditioned do not use up the quotas of the well-conditioned ones (a
            # 600-seed fuzz campaign: one screen-filling edge-on disc, conic conditioning 128 against a median of 1.35,
            # measured summation sensitivity 12 ... 33 allowances, was off by 2 ... 4 -- in both backward modes); they
            # are held to `ill_quota` allowances instead
            is_ill = np.asarray(ill, bool)[bad]
            quota = np.broadcast_to(np.asarray(ill_quota, np.float64), ratio.shape)[bad]      # (a number, or one per row)
            over = is_ill & ~(ratio[bad] < quota)
            assert not over.any(), f"{name}: ill-conditioned row off by {ratio[bad][over].max():.1f}x its allowance"
            bad_q = bad[~is_ill]
        else:
            bad_q = bad
        far = int((ratio[bad_q] > 3.0).sum())
        assert far <= outliers * n, f"{name}: {far} of {n} rows are off by more than 3x the per-row allowance"
        # (small tensors: ONE row marginally outside -- under 1.5x -- is rounding noise of the atomic sums, not a finding:
        # a 300-row case was seen at 1.01x in one run of 400 with the same inputs; `near * n` would allow 0.3 rows)
        marginal_ok = len(bad_q) <= 1 and (len(bad_q) == 0 or ratio[bad_q].max() < 1.5)
        assert len(bad_q) <= near * n or marginal_ok, f"{name}: {len(bad_q)} of {n} rows fail the per-row gradient check"
        capped = bad if ill is None else bad[~np.asarray(ill, bool)[bad]]
        if len(capped):
            assert ratio[capped].max() < cap, f"{name}: outlier row off by {ratio[capped].max():.1f}x the allowance"
    return bad


def summation_sensitivity(cam, st, og, names=("means3D", "scales", "rotations"), rtol=2e-4, floor=2e-7, eps=2e-6, seeds=3, rows=None):
    """MEASURED conditioning of the dense gradients with respect to fp32 summation: the per-Gaussian sums the oracle's
    preprocess backward starts from (its double sums, og["_per_gaussian"], from orc.backward(abs_sums=True)) are
    perturbed by eps * (sum of the ABSOLUTE per-fragment terms) * N(0,1) -- eps = 2e-6 is what an fp32 sum of a few
    hundred to a few thousand rounded terms carries (sqrt(n) * 6e-8 per addition plus ~3e-7 per term from v_exp_f32 /
    v_rcp_f32) -- and the oracle's own fp32 chain re-run.  Returns {name: per-row movement in units of the comparison's
    allowance (rtol, floor)}: a row that moves by m cannot be held closer than ~m allowances by ANY fp32 implementation.
    The movement is the largest over `seeds` random directions in the space of the 17 sums.
    rows (round 6): measure these rows only (the chain is per Gaussian; returns arrays of len(rows)) -- cheap enough for many
    draws.  An edge-on disc amplifies ONE direction of that space and three draws sample it poorly (headline view 5, row
    1357233, conic conditioning 114: 8.9 allowances over three draws, 35.2 over sixteen; a build that adds the same terms in
    another order -- k_seg_bwd's row-packed tail chunks -- landed on that direction with 35.4): the full-size tests
    re-measure their offenders with 64 draws."""
    from oracle import oracle as orc
    pg = og["_per_gaussian"]
    assert pg.get("abs") is not None, "run orc.backward(..., abs_sums=True)"
    tmax = {k: max(np.abs(np.asarray(og[k], np.float64)).max(), 1e-30) for k in names if og.get(k) is not None}
    if rows is not None:
        rows = np.asarray(rows, np.int64)
        cut = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a)[rows])
        st = dict(st, inputs={k: cut(v) for k, v in st["inputs"].items()}, radii=cut(st["radii"]), cov3D=cut(st["cov3D"]),
                  clamped=cut(st["clamped"]))
        pg = {k: cut(v) for k, v in pg.items()}
        og = {k: (cut(og[k]) if og.get(k) is not None else None) for k in names}
    moved = {}
    for seed in range(seeds):
        g = orc.preprocess_backward(cam, st, pg["mean2D"], pg["conic"], pg["opacity"], pg["attr"],
                                    perturb=(eps, 100 + seed, pg["abs"]))
        for k in names:
            if og.get(k) is None:
                continue
            w = np.asarray(og[k], np.float64).reshape(og[k].shape[0], -1)
            d = np.abs(np.asarray(g[k], np.float64).reshape(w.shape) - w).max(axis=1)
            allow = rtol * np.abs(w).max(axis=1) + floor * tmax[k]
            moved[k] = np.maximum(moved.get(k, 0.0), d / allow)
    return moved


def ill_conditioned(st, factor=20.0):
    """(mask, explain): rows whose conic conditioning is more than `factor` times the median of the visible rows."""
    cond = conic_conditioning(st)
    vis = np.asarray(st["radii"]) > 0
    typical = float(np.median(cond[vis])) if vis.any() else 1.0
    return cond > factor * typical, (lambda i: f"conic conditioning {cond[i]:.3g} (median of the visible rows {typical:.3g})")


def conic_conditioning(st):
    """Per Gaussian, from the oracle's forward state: (A + C)^2 / (4 det) of the 2D conic -- 1 for a circle, large for
    the edge-on discs whose gradient chain conic -> cov2D -> cov3D -> (scale, rotation, mean) divides by det^2 and so
    amplifies the rounding of the per-Gaussian sums it starts from (inf for culled rows)."""
    con = np.asarray(st["conic_op"], np.float64)
    det = con[:, 0] * con[:, 2] - con[:, 1] ** 2
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(det > 0, (con[:, 0] + con[:, 2]) ** 2 / (4 * det), np.inf)


def normal_guidance_loss(cov_quat, cov_scale, normal, R_cam2world):
    """torch restatement of the consumer loss (reference loss/normal_guidance.py:3-22), used by the
    end-to-end gradient tests; pinned against ref_normal_guidance.npz."""
    import torch
    cs = cov_scale.permute(1, 2, 0).reshape(-1, 1, 3)
    q = cov_quat.permute(1, 2, 0).reshape(-1, 4)
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    rot = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                       two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                       two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1).reshape(-1, 3, 3)
    rs = rot.detach() * cs
    Rm = torch.as_tensor(R_cam2world, dtype=normal.dtype, device=normal.device)
    nw = (Rm @ normal.reshape(3, -1)).t()               # [npix,3] world normals
    nw = nw[:, :, None].expand(-1, 3, 3)
    return 0.8 * (rot * nw).sum(dim=-2).abs().mean() + 0.2 * (rs * nw).sum(dim=-2).abs().mean()
