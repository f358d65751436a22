"""TEST INFRASTRUCTURE (not shipped, never imported by vegs_amd/): numpy restatement of the reference's densification
step, scene/gaussian_model.py:263-413 -- densify_and_prune = densify_and_clone + densify_and_split + prune_points with
the optimizer-state surgery of cat_tensors_to_optimizer / _prune_optimizer -- and reset_opacity (:215-218).

Pinned by tests/golden/ref_densify.npz, which the reference's own GaussianModel methods produced on the CPU
(tests/golden/make_golden.py part_d; tests/test_densify_oracle.py).

What the reference's sequence of cat / mask operations amounts to, for P Gaussians in and N = 2 split copies:

  g[i]        = xyz_gradient_accum[i] / denom[i], NaN -> 0                                       (:391-392)
  big[i]      = max_k exp(scaling[i,k]) > percent_dense * extent
  clone[i]    = |g[i]| >= max_grad and not big[i]                                                 (:377-380)
  split[i]    = g[i] >= max_grad and big[i]     (the clones appended in between have gradient 0)  (:356-362)
  after clone + split + prune_points(split originals) the rows are, in this order,
      A: the originals that were not split          (parameters and Adam moments carried along)
      B: one copy of every cloned original          (moments 0)
      C1, C2: the split samples, copy-major (`repeat(N, 1)`): xyz = R(q) (noise * exp(scaling)) + xyz,
              scaling = log(exp(scaling) / (0.8 N)), everything else copied; moments 0                (:364-373)
  then one more prune_points over ALL rows with (:396-402)
      opacity:  sigmoid(opacity) < min_opacity
      world:    max_k exp(scaling_row) > 0.1 * extent            only if max_screen_size is truthy
      screen:   max_radii2D > max_screen_size                    NEVER fires: densification_postfix (:349-351) has just
                                                                 reset max_radii2D to zeros for every row
  and xyz_gradient_accum, denom, max_radii2D are zeros of the new length (:349-351 + the prune's masking).
"""
import numpy as np

NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
N_SPLIT = 2


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)


def build_rotation(r):
    """utils/general_utils.py:97-118 in float32"""
    r = r.astype(np.float32)
    norm = np.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = np.zeros((q.shape[0], 3, 3), np.float32)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - w * z)
    R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y)
    R[:, 2, 1] = 2 * (y * z + w * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def plan(opacity, scaling, accum, denom, max_grad, min_opacity, extent, percent_dense, prune_big):
    """-> (src [n_out] source row, kind [n_out] 0 = A, 1 = B, 2 = C1, 3 = C2, draw [n_out] row of the split's noise
    (kinds 2, 3; -1 otherwise), S = number of split originals = noise rows per copy)"""
    P = opacity.shape[0]
    with np.errstate(divide="ignore", invalid="ignore"):
        g = (accum.reshape(P).astype(np.float32) / denom.reshape(P).astype(np.float32)).astype(np.float32)
    g[np.isnan(g)] = 0.0
    s = np.exp(scaling.astype(np.float32)).astype(np.float32)
    smax = s.max(axis=1)
    big = smax > np.float32(percent_dense * extent)
    clone = (np.abs(g) >= np.float32(max_grad)) & ~big
    split = (g >= np.float32(max_grad)) & big
    op = _sigmoid(opacity.reshape(P))
    s_new = np.exp(np.log(s / np.float32(0.8 * N_SPLIT))).astype(np.float32)
    prune_orig = op < np.float32(min_opacity)
    prune_new = prune_orig.copy()
    if prune_big:
        prune_orig = prune_orig | (smax > np.float32(0.1 * extent))
        prune_new = prune_new | (s_new.max(axis=1) > np.float32(0.1 * extent))
    idx = np.arange(P)
    a = idx[~split & ~prune_orig]
    b = idx[clone & ~prune_orig]
    rank = np.cumsum(split) - 1                     # position among the split originals = row of the draw
    S = int(split.sum())
    c = idx[split & ~prune_new]
    src = np.concatenate([a, b, c, c])
    kind = np.concatenate([np.zeros(len(a)), np.ones(len(b)), np.full(len(c), 2), np.full(len(c), 3)]).astype(np.int32)
    draw = np.concatenate([np.full(len(a) + len(b), -1), rank[c], S + rank[c]]).astype(np.int64)
    return src, kind, draw, S


def densify_and_prune(params, m, v, accum, denom, noise, max_grad, min_opacity, extent, percent_dense, prune_big):
    """params / m / v: dicts over NAMES (m, v may be None: an optimizer without state).  noise [2 S, 3] unit normal."""
    src, kind, draw, S = plan(params["opacity"], params["scaling"], accum, denom, max_grad, min_opacity, extent,
                              percent_dense, prune_big)
    assert noise.shape[0] == N_SPLIT * S
    out, om, ov = {}, {}, {}
    new = kind >= 2
    for k in NAMES:
        out[k] = params[k][src].copy()
        if m is not None:
            keep = (kind == 0).reshape((-1,) + (1,) * (params[k].ndim - 1))
            om[k] = np.where(keep, m[k][src], np.float32(0.0)).astype(np.float32)
            ov[k] = np.where(keep, v[k][src], np.float32(0.0)).astype(np.float32)
    if new.any():
        i = src[new]
        std = np.exp(params["scaling"][i].astype(np.float32)).astype(np.float32)
        samples = (noise[draw[new]].astype(np.float32) * std).astype(np.float32)
        R = build_rotation(params["rotation"][i])
        out["xyz"][new] = (np.einsum("nij,nj->ni", R.astype(np.float64), samples.astype(np.float64)).astype(np.float32)
                           + params["xyz"][i])
        out["scaling"][new] = np.log(std / np.float32(0.8 * N_SPLIT)).astype(np.float32)
    n = len(src)
    stats = (np.zeros((n, 1), np.float32), np.zeros((n, 1), np.float32), np.zeros(n, np.float32))
    return out, (om if m is not None else None), (ov if m is not None else None), stats, (src, kind, draw, S)


def reset_opacity(opacity, cap=0.01):
    """scene/gaussian_model.py:215-218: inverse_sigmoid(min(sigmoid(o), 0.01)); the caller zeroes the moments"""
    x = np.minimum(_sigmoid(opacity), np.float32(cap)).astype(np.float32)
    return np.log(x / (np.float32(1.0) - x)).astype(np.float32)
