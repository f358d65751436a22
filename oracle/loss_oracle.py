"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float64) of the per-pixel losses that sit right
after the rasterizer in the reference's training iteration:

  * l1_loss                      reference utils/loss_utils.py:18-22 (mask=None branch)
  * ssim / _ssim                 reference utils/loss_utils.py:30-79 (window 11, sigma 1.5, zero padding,
                                 size_average=True, no mask -- the call of train.py:164)
  * loss_normal_guidance         reference loss/normal_guidance.py:3-22, with quaternion_to_matrix of
                                 utils/graphics_utils.py:204-248 and cam_normal_to_world_normal :362-368

Values AND analytic gradients (w.r.t. the rendered image / cov_quat / cov_scale).  Pinned against outputs
of the reference's own functions + torch autograd: tests/golden/ref_photometric.npz and
tests/golden/ref_normal_guidance.npz (tests/test_oracle_golden.py).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.
"""
from math import exp

import numpy as np

C1 = 0.01 ** 2
C2 = 0.03 ** 2


def gaussian_window(window_size=11, sigma=1.5):
    """utils/loss_utils.py:30-32: float32 tensor of exp(.) values, divided by its float32 sum."""
    g = np.array([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], np.float32)
    return (g / g.sum(dtype=np.float32)).astype(np.float32)


def _conv(img, g):
    """depthwise 2D correlation with the outer-product window, zero padding (F.conv2d(..., padding=ws//2,
    groups=C), utils/loss_utils.py:51-62); the window is symmetric so this is also its own adjoint."""
    g = g.astype(np.float64)
    r = len(g) // 2
    C, H, W = img.shape
    pad = np.zeros((C, H + 2 * r, W + 2 * r))
    pad[:, r:r + H, r:r + W] = img
    tmp = np.zeros((C, H + 2 * r, W))
    for k in range(len(g)):
        tmp += g[k] * pad[:, :, k:k + W]
    out = np.zeros((C, H, W))
    for k in range(len(g)):
        out += g[k] * tmp[:, k:k + H, :]
    return out


def l1_loss(x, y):
    return np.abs(x.astype(np.float64) - y.astype(np.float64)).mean()


def ssim_maps(x, y, window_size=11):
    x = x.astype(np.float64)
    y = y.astype(np.float64)
    g = gaussian_window(window_size)
    mu1, mu2 = _conv(x, g), _conv(y, g)
    s1 = _conv(x * x, g) - mu1 * mu1
    s2 = _conv(y * y, g) - mu2 * mu2
    s12 = _conv(x * y, g) - mu1 * mu2
    A1, A2 = 2 * mu1 * mu2 + C1, 2 * s12 + C2
    B1, B2 = mu1 * mu1 + mu2 * mu2 + C1, s1 + s2 + C2
    S = A1 * A2 / (B1 * B2)
    return S, (mu1, mu2, A1, A2, B1, B2, g)


def ssim(x, y, window_size=11):
    return ssim_maps(x, y, window_size)[0].mean()


def photometric(x, y, g_l1=1.0, g_ssim=1.0):
    """(l1 mean, ssim mean, d(g_l1*l1 + g_ssim*ssim)/dx)."""
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    S, (mu1, mu2, A1, A2, B1, B2, g) = ssim_maps(x, y)
    n = x.size
    # partial derivatives of S w.r.t. the three window moments that depend on x
    d_mu = 2 * mu2 * (A2 - A1) / (B1 * B2) - 2 * mu1 * S * (1 / B1 - 1 / B2)   # total, incl. through s1 and s12
    d_e11 = -S / B2
    d_e12 = 2 * A1 / (B1 * B2)
    grad_ssim = _conv(d_mu, g) + 2 * x64 * _conv(d_e11, g) + y64 * _conv(d_e12, g)
    grad = g_ssim * grad_ssim / n + g_l1 * np.sign(x64 - y64) / n
    return np.abs(x64 - y64).mean(), S.mean(), grad


def quaternion_to_matrix(q):
    """utils/graphics_utils.py:204-248; q [...,4] = (r,i,j,k), normalised by |q|^2 inside."""
    r, i, j, k = (q[..., a] for a in range(4))
    with np.errstate(divide="ignore", invalid="ignore"):
        two_s = 2.0 / (q * q).sum(-1)
        o = np.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                      two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                      two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def normal_guidance(cov_quat, cov_scale, normal, R_cam2world, g=1.0):
    """(loss, dL/dcov_quat [4,H,W], dL/dcov_scale [3,H,W]) of loss/normal_guidance.py:3-22."""
    _, H, W = normal.shape
    n_pix = H * W
    q = cov_quat.astype(np.float64).transpose(1, 2, 0).reshape(-1, 4)
    s = cov_scale.astype(np.float64).transpose(1, 2, 0).reshape(-1, 3)
    nw = (R_cam2world.astype(np.float32).astype(np.float64) @ normal.astype(np.float64).reshape(3, -1)).T  # [n,3]
    Rm = quaternion_to_matrix(q)                                 # [n,3,3]
    c = np.einsum("nij,ni->nj", Rm, nw)                          # column j of R . n
    t1 = np.abs(c)
    t2 = np.abs(c * s)
    loss = 0.8 * t1.mean() + 0.2 * t2.mean()
    k1, k2 = 0.8 / (3 * n_pix), 0.2 / (3 * n_pix)
    # scale gradient (rotation detached in the second term)
    ds = k2 * np.sign(c * s) * c
    # rotation gradient (first term only), then through quaternion_to_matrix
    G = k1 * np.sign(c)[:, None, :] * nw[:, :, None]            # dL/dR[i][j] = k1 sign(c_j) n_i
    r, i, j, k = (q[:, a] for a in range(4))
    with np.errstate(divide="ignore", invalid="ignore"):
        two_s = 2.0 / (q * q).sum(-1)
    M = np.stack([-(j * j + k * k), i * j - k * r, i * k + j * r, i * j + k * r, -(i * i + k * k), j * k - i * r,
                  i * k - j * r, j * k + i * r, -(i * i + j * j)], -1).reshape(-1, 3, 3)
    GM = (G * M).sum((1, 2))
    dr = -k * G[:, 0, 1] + j * G[:, 0, 2] + k * G[:, 1, 0] - i * G[:, 1, 2] - j * G[:, 2, 0] + i * G[:, 2, 1]
    di = j * (G[:, 0, 1] + G[:, 1, 0]) + k * (G[:, 0, 2] + G[:, 2, 0]) - 2 * i * (G[:, 1, 1] + G[:, 2, 2]) + r * (G[:, 2, 1] - G[:, 1, 2])
    dj = -2 * j * (G[:, 0, 0] + G[:, 2, 2]) + i * (G[:, 0, 1] + G[:, 1, 0]) + r * (G[:, 0, 2] - G[:, 2, 0]) + k * (G[:, 1, 2] + G[:, 2, 1])
    dk = -2 * k * (G[:, 0, 0] + G[:, 1, 1]) + r * (G[:, 1, 0] - G[:, 0, 1]) + i * (G[:, 0, 2] + G[:, 2, 0]) + j * (G[:, 1, 2] + G[:, 2, 1])
    dM = np.stack([dr, di, dj, dk], -1)
    with np.errstate(invalid="ignore", over="ignore"):
        dq = two_s[:, None] * dM - (two_s ** 2 * GM)[:, None] * q
    return (loss, g * dq.reshape(H, W, 4).transpose(2, 0, 1), g * ds.reshape(H, W, 3).transpose(2, 0, 1))
