"""Quaternion <-> rotation-matrix conversions of the op-by-op (unfused) composition paths, in this repository's own
formulation.  Conventions are the reference's (utils/graphics_utils.py:140-248 defines the same maps; pinned against its
outputs by tests/test_harness.py): quaternions are (w, x, y, z), need not be unit length going in, and come out of
`matrix_to_quaternion` with the component that defines the chosen branch positive.

The fused paths never call these: csrc/instances.hip (k_inst_fwd / k_box_*) carries the same algebra in registers.
"""
import torch


def _cross_matrix(v):
    """[v]_x with [v]_x u = v x u, batched over leading dimensions."""
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    o = torch.zeros_like(x)
    return torch.stack((o, -z, y, z, o, -x, -y, x, o), -1).reshape(v.shape[:-1] + (3, 3))


def quaternion_to_matrix(q):
    """R(q) = ((w^2 - |v|^2) I + 2 v v^T + 2 w [v]_x) / |q|^2 for q = (w, v): the rotation of the normalised quaternion,
    written homogeneously so that no separate normalisation step is needed (a zero quaternion gives NaN, as any division
    by |q|^2 does -- what an uncovered pixel of the cov_quat image turns into downstream)."""
    w, v = q[..., :1], q[..., 1:]
    vv = (v * v).sum(-1, keepdim=True)
    eye = torch.eye(3, dtype=q.dtype, device=q.device)
    num = (w * w - vv)[..., None] * eye + 2.0 * v[..., :, None] * v[..., None, :] + 2.0 * w[..., None] * _cross_matrix(v)
    return num / (w * w + vv)[..., None]


def matrix_to_quaternion(m, floor=0.1):
    """Shepperd's method on the symmetric matrix S = 4 q q^T, whose entries are linear in a rotation matrix m:
        diag S = 1 + (+ + +, + - -, - + -, - - +) . (m00, m11, m22),   S_0k = antisymmetric part,   S_jk = symmetric part.
    Every row of S is q scaled by 4 q_i; the row with the LARGEST diagonal is the best conditioned one (first maximum on
    ties), and q = S_i / (2 sqrt(S_ii)).  `floor` bounds the divisor from below (0.1: the reference's guard against a
    matrix that is no rotation; a proper rotation always has a diagonal entry >= 1)."""
    d = torch.stack((m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]), -1)
    signs = torch.tensor([[1.0, 1.0, 1.0], [1.0, -1.0, -1.0], [-1.0, 1.0, -1.0], [-1.0, -1.0, 1.0]], dtype=m.dtype, device=m.device)
    diag = 1.0 + d @ signs.t()                                   # [..., 4] = 4 (w^2, x^2, y^2, z^2)
    anti = torch.stack((m[..., 2, 1] - m[..., 1, 2], m[..., 0, 2] - m[..., 2, 0], m[..., 1, 0] - m[..., 0, 1]), -1)   # 4 w (x, y, z)
    sym = torch.stack((m[..., 1, 0] + m[..., 0, 1], m[..., 0, 2] + m[..., 2, 0], m[..., 2, 1] + m[..., 1, 2]), -1)    # 4 (xy, xz, yz)
    i = diag.argmax(-1, keepdim=True)
    root = torch.sqrt(torch.gather(diag, -1, i).clamp_min(0.0))   # 2 |q_i|; only the chosen entry is differentiated
    rows = torch.stack((
        torch.stack((root[..., 0] ** 2, anti[..., 0], anti[..., 1], anti[..., 2]), -1),
        torch.stack((anti[..., 0], root[..., 0] ** 2, sym[..., 0], sym[..., 1]), -1),
        torch.stack((anti[..., 1], sym[..., 0], root[..., 0] ** 2, sym[..., 2]), -1),
        torch.stack((anti[..., 2], sym[..., 1], sym[..., 2], root[..., 0] ** 2), -1)), -2)
    row = torch.gather(rows, -2, i[..., None].expand(i.shape[:-1] + (1, 4))).squeeze(-2)
    return row / (2.0 * root.clamp_min(floor))
