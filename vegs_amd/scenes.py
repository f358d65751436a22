"""Synthetic, seeded scenes and cameras for tests and bench.py.

Host-side only (numpy), device-agnostic.  The camera record restates the matrices the
reference builds in scene/cameras.py:76-88 (world_view_transform, principal-point-offset
projection, full_proj_transform, camera_center) from utils/graphics_utils.py:266-277
(getWorld2View2) and :305-337 (getProjectionMatrixwithPrincipalPointOffset); tests pin it
against golden vectors produced by importing those reference functions.
"""
import math
from dataclasses import dataclass

import numpy as np

SH_C0 = 0.28209479177387814

# KITTI-360 perspective intrinsics at 1408x376
KITTI360_W, KITTI360_H = 1408, 376
KITTI360_FX = KITTI360_FY = 552.554261
KITTI360_CX, KITTI360_CY = 682.049453, 238.769549


def rgb2sh(rgb):
    return (rgb - 0.5) / SH_C0


def world2view(R, t):
    """W2C 4x4 (column-vector form) for camera-to-world rotation R and W2C translation t."""
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = R.T
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return Rt.astype(np.float32)


def projection_ppo(znear, zfar, fovx, fovy, fx, fy, cx, cy, w, h):
    """Principal-point-offset projection, float32 4x4 (column-vector form)."""
    f32 = np.float32
    top_c = math.tan(fovy / 2) * znear
    right_c = math.tan(fovx / 2) * znear
    dx = (cx - w / 2) / fx * znear
    dy = (cy - h / 2) / fy * znear
    top, bottom = top_c + dy, -top_c + dy
    left, right = -right_c + dx, right_c + dx
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = f32(2.0 * znear / (right - left))
    P[1, 1] = f32(2.0 * znear / (top - bottom))
    P[0, 2] = f32((right + left) / (right - left))
    P[1, 2] = f32((top + bottom) / (top - bottom))
    P[3, 2] = 1.0
    P[2, 2] = f32((zfar + znear) / (zfar - znear))
    P[2, 3] = f32(-(zfar * znear) / (zfar - znear))
    return P


@dataclass
class SynthCamera:
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: np.ndarray  # [4,4] float32, row-vector convention (transposed W2C)
    full_proj_transform: np.ndarray   # [4,4] float32
    camera_center: np.ndarray         # [3] float32
    R: np.ndarray                     # camera-to-world rotation [3,3]
    T: np.ndarray                     # W2C translation [3]

    @property
    def tanfovx(self):
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.FoVy * 0.5)


def make_camera(R, t, width, height, fx, fy, cx, cy, znear=0.01, zfar=100.0):
    fovx = 2 * math.atan(width / (2 * fx))
    fovy = 2 * math.atan(height / (2 * fy))
    view = world2view(R, t).T.copy()
    proj = projection_ppo(znear, zfar, fovx, fovy, fx, fy, cx, cy, width, height).T.copy()
    full = (view @ proj).astype(np.float32)
    center = np.linalg.inv(view.astype(np.float64))[3, :3].astype(np.float32)
    return SynthCamera(int(height), int(width), fovx, fovy, view, full, center,
                       np.asarray(R, np.float64), np.asarray(t, np.float64))


def lookat_camera(eye, target, up, width, height, fov_deg):
    """Centred pinhole camera at `eye` looking at `target` (OpenCV axes: x right, y down, z fwd)."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z], axis=1)  # columns = camera axes in world
    t = -R.T @ eye
    f = width / (2 * math.tan(math.radians(fov_deg) / 2))
    return make_camera(R, t, width, height, f, f, width / 2, height / 2)


# world: x forward, y left, z up ; camera: x right, y down, z forward
R_KITTI = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])


def kitti_camera(x_forward=0.0, y_left=0.0, width=1376, height=376):
    """KITTI-360-shaped camera at (x_forward, y_left, 0) looking along +x."""
    s = width / KITTI360_W
    eye = np.array([x_forward, y_left, 0.0])
    t = -R_KITTI.T @ eye
    return make_camera(R_KITTI, t, width, height, KITTI360_FX, KITTI360_FY * (height / KITTI360_H),
                       KITTI360_CX * s, KITTI360_CY * (height / KITTI360_H))


def _normalize(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _mat_to_quat(R):
    """Rotation matrices [n,3,3] -> unit quaternions (w,x,y,z), w >= 0."""
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    q = np.empty((R.shape[0], 4))
    q[:, 0] = np.sqrt(np.maximum(0, 1 + m00 + m11 + m22)) / 2
    q[:, 1] = np.sqrt(np.maximum(0, 1 + m00 - m11 - m22)) / 2
    q[:, 2] = np.sqrt(np.maximum(0, 1 - m00 + m11 - m22)) / 2
    q[:, 3] = np.sqrt(np.maximum(0, 1 - m00 - m11 + m22)) / 2
    q[:, 1] = np.copysign(q[:, 1], R[:, 2, 1] - R[:, 1, 2])
    q[:, 2] = np.copysign(q[:, 2], R[:, 0, 2] - R[:, 2, 0])
    q[:, 3] = np.copysign(q[:, 3], R[:, 1, 0] - R[:, 0, 1])
    return _normalize(q)


def _quat_mul(a, b):
    aw, ax, ay, az = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    bw, bx, by, bz = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], axis=1)


def _frame_from_normal(n, rng):
    """Column-stacked (normal, ortho1, ortho2), cf. utils/graphics_utils.py:346-360."""
    n = _normalize(n)
    helper = _normalize(n + rng.random(3) + 1e-3)
    o1 = _normalize(helper - (n * helper).sum(-1, keepdims=True) * n)
    o2 = _normalize(np.cross(n, o1))
    return np.stack([n, o1, o2], axis=-1)


def scene_random(P=10000, sh_degree=0, seed=0, extent=0.5, scale=0.02):
    """C1: random blob (cf. scene/gaussian_model.py:471 init range)."""
    rng = np.random.default_rng(seed)
    M = 16
    sc = dict(
        means3D=rng.uniform(-extent, extent, (P, 3)),
        scales=np.exp(rng.normal(math.log(scale), 0.3, (P, 3))),
        rotations=_normalize(rng.normal(size=(P, 4))),
        opacities=1 / (1 + np.exp(-rng.normal(size=(P, 1)))),
    )
    shs = np.zeros((P, M, 3))
    shs[:, 0, :] = rgb2sh(rng.uniform(0, 1, (P, 3)))
    K = (sh_degree + 1) ** 2
    if K > 1:
        shs[:, 1:K, :] = rng.normal(0, 0.05, (P, K - 1, 3))
    sc["shs"] = shs
    return {k: v.astype(np.float32) for k, v in sc.items()}, sh_degree


def camera_c1(width=256, height=256):
    return lookat_camera([0.0, -2.0, 0.0], [0, 0, 0], [0, 0, 1.0], width, height, 45.0)


def scene_street(P=500000, length=120.0, sh_degree=3, seed=1, x_min=0.0):
    """C2/C3: KITTI-360-shaped street of VEGS-style discs (utils/norminit_utils.py:217-219)."""
    rng = np.random.default_rng(seed)
    n_road = int(0.4 * P)
    n_fac = int(0.4 * P)
    n_clu = P - n_road - n_fac
    x = rng.uniform(x_min, x_min + length, P)
    y = np.empty(P)
    z = np.empty(P)
    nrm = np.empty((P, 3))
    # road
    y[:n_road] = rng.uniform(-8, 8, n_road)
    z[:n_road] = -1.55 + rng.normal(0, 0.02, n_road)
    nrm[:n_road] = [0, 0, 1]
    # facades
    side = rng.choice([-1.0, 1.0], n_fac)
    y[n_road:n_road + n_fac] = side * rng.uniform(8, 15, n_fac)
    z[n_road:n_road + n_fac] = rng.uniform(-1.5, 10, n_fac)
    nrm[n_road:n_road + n_fac] = np.stack([np.zeros(n_fac), -side, np.zeros(n_fac)], axis=1)
    # clutter
    y[n_road + n_fac:] = rng.uniform(-30, 30, n_clu)
    z[n_road + n_fac:] = rng.uniform(-1.5, 6, n_clu)
    nrm[n_road + n_fac:] = _normalize(rng.normal(size=(n_clu, 3)))
    perm = rng.permutation(P)  # storage order is not spatially sorted
    means = np.stack([x, y, z], axis=1)[perm]
    nrm = nrm[perm]
    q = _mat_to_quat(_frame_from_normal(nrm, rng))
    ang = np.radians(10.0) * rng.normal(size=P)
    axis = _normalize(rng.normal(size=(P, 3)))
    jit = np.concatenate([np.cos(ang / 2)[:, None], axis * np.sin(ang / 2)[:, None]], axis=1)
    q = _normalize(_quat_mul(jit, q))
    scales = np.array([1e-5, 0.1, 0.1]) * np.exp(rng.normal(0, 0.3, (P, 3)))
    M = 16
    shs = np.zeros((P, M, 3))
    shs[:, 0, :] = rgb2sh(rng.uniform(0, 1, (P, 3)))
    K = (sh_degree + 1) ** 2
    if K > 1:
        shs[:, 1:K, :] = rng.normal(0, 0.05, (P, K - 1, 3))
    sc = dict(means3D=means, scales=scales, rotations=q,
              opacities=1 / (1 + np.exp(-rng.normal(0, 1.5, (P, 1)))), shs=shs)
    return {k: v.astype(np.float32) for k, v in sc.items()}, sh_degree
