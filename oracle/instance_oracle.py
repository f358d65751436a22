"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float64) of the box-instance branch of the reference's
prepare_rasterization (gaussian_renderer/__init__.py:122-126 means, :140-153 rotations/scales), i.e.

    means'     = homogeneous(box2world @ [x;1])                                   (:123-126)
    S, Rb      = decompose_T_to_RS(box2world)     utils/graphics_utils.py:49-53   (column norms / normalised columns)
    rotations' = matrix_to_quaternion(Rb @ quaternion_to_matrix(rotations))       utils/graphics_utils.py:140-248
    scales'    = scales * S                                                       (:153)

with analytic gradients w.r.t. xyz, scales, rotations and the 4x4 box2world.  Pinned against outputs of the
reference's own functions + torch autograd (tests/golden/ref_instances.npz, tests/test_oracle_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

from .loss_oracle import quaternion_to_matrix

# quat_by_rijk rows (utils/graphics_utils.py:172-188): entry k of candidate c is  sgn * (m[a] + s2 * m[b])  or t_c
_OFF = {  # (c, k) -> ((row,col), (row,col), sign of the second term)
    (0, 1): ((2, 1), (1, 2), -1), (0, 2): ((0, 2), (2, 0), -1), (0, 3): ((1, 0), (0, 1), -1),
    (1, 0): ((2, 1), (1, 2), -1), (1, 2): ((1, 0), (0, 1), +1), (1, 3): ((0, 2), (2, 0), +1),
    (2, 0): ((0, 2), (2, 0), -1), (2, 1): ((1, 0), (0, 1), +1), (2, 3): ((1, 2), (2, 1), +1),
    (3, 0): ((1, 0), (0, 1), -1), (3, 1): ((2, 0), (0, 2), +1), (3, 2): ((2, 1), (1, 2), +1),
}
_TSIGN = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], np.float64)   # t_c = 1 + sum sign * m_ii


def matrix_to_quaternion(m):
    """utils/graphics_utils.py:140-201.  Returns (q, chosen candidate index, q_abs of it)."""
    d = np.stack([m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]], -1)
    t = 1.0 + d @ _TSIGN.T                                            # [n,4]
    q_abs = np.sqrt(np.maximum(t, 0.0))
    c = q_abs.argmax(-1)
    n = m.shape[0]
    v = np.zeros((n, 4))
    for i in range(n):
        ci = c[i]
        for k in range(4):
            if k == ci:
                v[i, k] = q_abs[i, ci] ** 2
            else:
                (a, b, s2) = _OFF[(ci, k)]
                v[i, k] = m[i][a] + s2 * m[i][b]
    qa = q_abs[np.arange(n), c]
    return v / (2.0 * np.maximum(qa, 0.1))[:, None], c, qa, v


def forward(xyz, scales, rot, B):
    xyz, scales, rot, B = (np.asarray(a, np.float64) for a in (xyz, scales, rot, B))
    h = xyz @ B[:3, :3].T + B[:3, 3]
    w = xyz @ B[3, :3] + B[3, 3]
    S = np.linalg.norm(B[:3, :3], axis=0)
    Rb = B[:3, :3] / S
    Rm = Rb[None] @ quaternion_to_matrix(rot)
    q, _, _, _ = matrix_to_quaternion(Rm)
    return h / w[:, None], scales * S, q


def backward(xyz, scales, rot, B, g_means, g_scales, g_rot):
    """(dL/dxyz, dL/dscales, dL/drot, dL/dbox2world[4,4])"""
    xyz, scales, rot, B, g_means, g_scales, g_rot = (np.asarray(a, np.float64) for a in
                                                     (xyz, scales, rot, B, g_means, g_scales, g_rot))
    n = xyz.shape[0]
    dB = np.zeros((4, 4))
    # ---- means
    h = xyz @ B[:3, :3].T + B[:3, 3]
    w = xyz @ B[3, :3] + B[3, 3]
    y = h / w[:, None]
    gh = g_means / w[:, None]
    gw = -(g_means * y).sum(1) / w
    d_xyz = gh @ B[:3, :3] + gw[:, None] * B[3, :3][None]
    dB[:3, :3] += gh.T @ xyz
    dB[:3, 3] += gh.sum(0)
    dB[3, :3] += gw @ xyz
    dB[3, 3] += gw.sum()
    # ---- scales
    S = np.linalg.norm(B[:3, :3], axis=0)
    Rb = B[:3, :3] / S
    d_scales = g_scales * S
    dS = (g_scales * scales).sum(0)
    # ---- rotations: q' = v / (2 max(qa, 0.1)), v from Rm = Rb Rq
    Rq = quaternion_to_matrix(rot)
    Rm = Rb[None] @ Rq
    _, c, qa, v = matrix_to_quaternion(Rm)
    dRm = np.zeros((n, 3, 3))
    for i in range(n):
        ci = c[i]
        den = 2.0 * max(qa[i], 0.1)
        gv = g_rot[i] / den                                            # dL/dv
        gqa = -(g_rot[i] * v[i]).sum() / den ** 2 * 2.0 if qa[i] > 0.1 else 0.0      # through the denominator
        # v_c = qa^2 = t_c (t_c > 0 for the winning candidate); qa = sqrt(t_c)
        gt = gv[ci] + (gqa / (2.0 * qa[i]) if qa[i] > 0 else 0.0)
        for a in range(3):
            dRm[i, a, a] += gt * _TSIGN[ci, a]
        for k in range(4):
            if k != ci:
                (a, b, s2) = _OFF[(ci, k)]
                dRm[i][a] += gv[k]
                dRm[i][b] += s2 * gv[k]
    dRb = np.einsum("nij,nkj->ik", dRm, Rq)                            # dL/dRb = sum_n dRm Rq^T
    dRq = np.einsum("ji,njk->nik", Rb, dRm)                            # Rb^T dRm
    # through quaternion_to_matrix (same algebra as oracle/loss_oracle.py)
    r, i_, j, k = (rot[:, a] for a in range(4))
    two_s = 2.0 / (rot * rot).sum(-1)
    M = np.stack([-(j * j + k * k), i_ * j - k * r, i_ * k + j * r, i_ * j + k * r, -(i_ * i_ + k * k), j * k - i_ * r,
                  i_ * k - j * r, j * k + i_ * r, -(i_ * i_ + j * j)], -1).reshape(-1, 3, 3)
    G = dRq
    GM = (G * M).sum((1, 2))
    dr = -k * G[:, 0, 1] + j * G[:, 0, 2] + k * G[:, 1, 0] - i_ * G[:, 1, 2] - j * G[:, 2, 0] + i_ * G[:, 2, 1]
    di = j * (G[:, 0, 1] + G[:, 1, 0]) + k * (G[:, 0, 2] + G[:, 2, 0]) - 2 * i_ * (G[:, 1, 1] + G[:, 2, 2]) + r * (G[:, 2, 1] - G[:, 1, 2])
    dj = -2 * j * (G[:, 0, 0] + G[:, 2, 2]) + i_ * (G[:, 0, 1] + G[:, 1, 0]) + r * (G[:, 0, 2] - G[:, 2, 0]) + k * (G[:, 1, 2] + G[:, 2, 1])
    dk = -2 * k * (G[:, 0, 0] + G[:, 1, 1]) + r * (G[:, 1, 0] - G[:, 0, 1]) + i_ * (G[:, 0, 2] + G[:, 2, 0]) + j * (G[:, 1, 2] + G[:, 2, 1])
    d_rot = two_s[:, None] * np.stack([dr, di, dj, dk], -1) - (two_s ** 2 * GM)[:, None] * rot
    # ---- decompose_T_to_RS backward: Rb[:,j] = B33[:,j] / S_j,  S_j = |B33[:,j]|
    for col in range(3):
        gcol = dRb[:, col]
        dB[:3, col] += gcol / S[col] - (gcol @ Rb[:, col]) * Rb[:, col] / S[col] + dS[col] * Rb[:, col]
    return d_xyz, d_scales, d_rot, dB
