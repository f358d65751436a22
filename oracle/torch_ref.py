"""Independent PyTorch (autograd, float64-capable) restatement of the rasterizer.

TEST INFRASTRUCTURE ONLY ("parity unpinned", see vr_oracle.c).  Written from the spec in
SURVEY.md Appendix A (A.1-A.7, A-1..A-6) in a deliberately different formulation from
vr_oracle.c -- vectorised per tile, transmittance by cumprod, gradients by autograd -- so
that agreement between the two is evidence, not tautology.  Used to (i) generate the
golden fixtures under tests/golden/ (make_golden.py) and (ii) check the hand-derived
backward of vr_oracle.c.

SH colour follows utils/sh_utils.py:57-112; cov3D follows scene/gaussian_model.py:32-36 +
utils/general_utils.py:97-129 (quaternion w,x,y,z; no renormalisation inside the op).
"""
import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
      -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def sh_to_rgb(deg, sh, dirs):
    """sh [P,M,3], dirs [P,3] unit -> [P,3] (before +0.5)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
               + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def build_cov3d(scales, mod, q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    L = R * (mod * scales)[:, None, :]
    return L @ L.transpose(1, 2)


def rasterize(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, *,
              H, W, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix, campos, sh_degree,
              means2D=None, flags=0):
    """Returns (color[3,H,W], depth[1,H,W], cov_quat[4,H,W], cov_scale[3,H,W], alpha[1,H,W], radii[P]).

    If `means2D` ([P,3], requires_grad) is given it is added (as zeros) to the NDC position so
    that its gradient is d loss / d NDC -- the quantity the op deposits on `viewspace_points`.
    `flags`: VrFlags bits 0-3 of include/vegs_rast.h (1 scale x modifier, 2 normalised depth, 4 extra channels give
    no gradient through alpha, 8 identity fill of cov_quat) -- the open questions of SURVEY.md A.8 as switches.
    """
    f_scale, f_dnorm, f_noalpha, f_fill = bool(flags & 1), bool(flags & 2), bool(flags & 4), bool(flags & 8)
    dt = means3D.dtype
    P = means3D.shape[0]
    V, PM = viewmatrix.to(dt), projmatrix.to(dt)
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    t = (ph @ V)[:, :3]
    hom = ph @ PM
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None]
    if means2D is not None:
        ndc = ndc + means2D[:, :2]
    front = t[:, 2] > 0.2

    if cov3D_precomp is not None:
        c = cov3D_precomp
        Sig = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(-1, 3, 3)
    else:
        Sig = build_cov3d(scales, scale_modifier, rotations)
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    tz = torch.where(front, t[:, 2], torch.ones_like(t[:, 2]))
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    rx, ry = t[:, 0] / tz, t[:, 1] / tz
    inx = (rx >= -limx) & (rx <= limx)
    iny = (ry >= -limy) & (ry <= limy)
    # upstream treats a clamped t.x / t.y as a constant (zero gradient through it, A.6)
    tx = torch.where(inx, t[:, 0], (torch.clamp(rx, -limx, limx) * tz).detach())
    ty = torch.where(iny, t[:, 1], (torch.clamp(ry, -limy, limy) * tz).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], 1).reshape(-1, 2, 3)
    Wv = V[:3, :3].t()
    M2 = J @ Wv
    cov = M2 @ Sig @ M2.transpose(1, 2)
    a, b, c_ = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c_ - b * b
    ok = front & (det != 0)
    det_s = torch.where(ok, det, torch.ones_like(det))
    conA, conB, conC = c_ / det_s, -b / det_s, a / det_s
    mid = 0.5 * (a + c_)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    rad = torch.ceil(3 * torch.sqrt(lam.detach().float())).to(torch.int64)
    px = ((ndc[:, 0] + 1) * W - 1) * 0.5
    py = ((ndc[:, 1] + 1) * H - 1) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pxf, pyf, rf = px.detach().float(), py.detach().float(), rad.float()
    x0 = torch.clamp(((pxf - rf) / 16).to(torch.int64), 0, gx)
    y0 = torch.clamp(((pyf - rf) / 16).to(torch.int64), 0, gy)
    x1 = torch.clamp(((pxf + rf + 15) / 16).to(torch.int64), 0, gx)
    y1 = torch.clamp(((pyf + rf + 15) / 16).to(torch.int64), 0, gy)
    ok = ok & ((x1 - x0) * (y1 - y0) > 0)
    radii = torch.where(ok, rad, torch.zeros_like(rad)).to(torch.int32)

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos.to(dt)[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(sh_to_rgb(sh_degree, shs, d) + 0.5, 0.0)
    depth = t[:, 2]
    quat = rotations if rotations is not None else torch.zeros(P, 4, dtype=dt)
    scl = scales if scales is not None else torch.zeros(P, 3, dtype=dt)
    if f_scale:
        scl = scl * scale_modifier
    attrs = torch.cat([rgb, depth[:, None], quat, scl], 1)  # [P,11]

    out = torch.zeros(11, H, W, dtype=dt)
    Tfin = torch.ones(H, W, dtype=dt)
    bgv = bg.to(dt)
    depth_key = depth.detach().float()
    vis = torch.nonzero(ok)[:, 0]
    for tyi in range(gy):
        for txi in range(gx):
            sel = vis[(x0[vis] <= txi) & (x1[vis] > txi) & (y0[vis] <= tyi) & (y1[vis] > tyi)]
            if sel.numel() == 0:
                continue
            # sort by (depth fp32 bits, id): depth > 0 so value order == bit order; stable keeps id order
            order = torch.sort(depth_key[sel], stable=True)[1]
            sel = sel[order]
            ys = torch.arange(tyi * 16, min(tyi * 16 + 16, H))
            xs = torch.arange(txi * 16, min(txi * 16 + 16, W))
            yy, xx = torch.meshgrid(ys, xs, indexing="ij")
            pxs, pys = xx.reshape(-1).to(dt), yy.reshape(-1).to(dt)
            dx = px[sel][:, None] - pxs[None]
            dy = py[sel][:, None] - pys[None]
            power = -0.5 * (conA[sel][:, None] * dx * dx + conC[sel][:, None] * dy * dy) - conB[sel][:, None] * dx * dy
            alpha = opacities[sel].reshape(-1, 1) * torch.exp(torch.clamp(power, max=0.0))
            # min(0.99, .) with a straight-through gradient, as the published backward does (A.5)
            alpha = alpha + (torch.clamp(alpha, max=0.99) - alpha).detach()
            valid = (power <= 0) & (alpha >= 1.0 / 255.0)
            a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
            om = 1 - a_eff
            T_incl = torch.cumprod(om, 0)
            T_excl = torch.cat([torch.ones(1, om.shape[1], dtype=dt), T_incl[:-1]], 0)
            stop = valid & (T_incl < 1e-4)          # T*(1-alpha) < 1e-4 -> that splat is NOT applied
            dead = torch.cumsum(stop.to(torch.int32), 0) > 0
            wgt = torch.where(dead, torch.zeros_like(a_eff), a_eff * T_excl)   # [n, npix]
            acc = attrs[sel].t() @ wgt                                          # [11, npix]
            if f_noalpha:   # extra channels: gradient to the attributes only (weights are constants for them)
                acc = torch.cat([acc[:3], attrs[sel][:, 3:].t() @ wgt.detach()], 0)
            # final T = product of (1-alpha) over applied splats
            om_applied = torch.where(dead, torch.ones_like(om), om)
            Tf = torch.prod(om_applied, 0)
            out[:, ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1] = acc.reshape(11, len(ys), len(xs))
            Tfin[ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1] = Tf.reshape(len(ys), len(xs))
    color = out[0:3] + Tfin[None] * bgv[:, None, None]
    Tx = Tfin.detach() if f_noalpha else Tfin      # transmittance as seen by the extra channels
    depth_img, quat_img = out[3:4], out[4:8]
    if f_dnorm:
        A = 1 - Tx
        depth_img = torch.where(A > 0, depth_img / torch.where(A > 0, A, torch.ones_like(A)), torch.zeros_like(depth_img))
    if f_fill:
        quat_img = torch.cat([quat_img[0:1] + Tx[None], quat_img[1:]], 0)
    return color, depth_img, quat_img, out[8:11], (1 - Tfin)[None], radii
