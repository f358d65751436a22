"""GPU (-m gpu): seeded random sweep of the operator against the oracle -- image sizes that are not multiples
of the tile, every SH degree / storage width, both colour modes, both covariance modes, scale modifiers,
backgrounds, splats from sub-pixel to screen-filling, opacities including exact 0 and 1, cameras inside the
cloud, arbitrary subsets of the five upstream gradients.  Same bars as test_gpu_parity.py: radii / tile lists /
ranges and all five images bit-exact, gradients within the tensor-level tolerance."""
import os

import numpy as np
import pytest
import torch

import test_gpu_parity as tp
from test_gpu_parity import dev  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _cov6(scales, rot, mod):
    q = rot / np.linalg.norm(rot, axis=1, keepdims=True)
    r, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                  2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                  2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    M = R * (scales * mod)[:, None, :]
    S = M @ M.transpose(0, 2, 1)
    return np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)


# VEGS_FUZZ_SEEDS=a:b widens the sweep for a campaign (the default 28 seeds are what every round runs)
_SEEDS = range(*(int(v) for v in os.environ["VEGS_FUZZ_SEEDS"].split(":"))) if os.environ.get("VEGS_FUZZ_SEEDS") else range(28)


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_configuration(seed, dev):  # noqa: F811
    from vegs_amd import scenes
    rng = np.random.default_rng(9000 + seed)
    P = int(rng.choice([1, 2, 63, 64, 65, 300, 1500, 4000]))
    W, H = int(rng.integers(17, 300)), int(rng.integers(9, 200))
    deg = int(rng.integers(0, 4))
    M = int(rng.choice([m for m in (1, 4, 9, 16) if m >= (deg + 1) ** 2]))
    extent = float(rng.choice([0.05, 0.5, 3.0]))
    scale = float(rng.choice([1e-4, 0.01, 0.05, 0.5]))
    sc, _ = scenes.scene_random(P=P, sh_degree=3, seed=seed, extent=extent, scale=scale)
    sc["shs"] = np.ascontiguousarray(sc["shs"][:, :M])
    if seed % 3 == 0:                                        # exact 0 / 1 opacities and un-normalised quaternions
        sc["opacities"] = rng.choice([0.0, 1.0, 0.3, 0.9999], size=(P, 1)).astype(np.float32)
        sc["rotations"] = (sc["rotations"] * rng.uniform(0.2, 3.0, (P, 1))).astype(np.float32)
    eye = rng.normal(size=3)
    eye = eye / np.linalg.norm(eye) * float(rng.choice([0.2, 1.0, 2.5]))      # 0.2: camera inside the cloud
    cam = scenes.lookat_camera(eye, rng.normal(size=3) * 0.1, [0, 0, 1.0], W, H, float(rng.uniform(30, 110)))
    mod = float(rng.choice([1.0, 0.5, 1.7]))
    bg = rng.uniform(0, 1, 3).astype(np.float32)
    pre_col, pre_cov = bool(rng.integers(0, 2)) and seed % 2 == 1, bool(rng.integers(0, 2)) and seed % 4 == 3
    inputs = dict(means3D=sc["means3D"], shs=None if pre_col else sc["shs"],
                  colors_precomp=rng.uniform(0, 1, (P, 3)).astype(np.float32) if pre_col else None,
                  opacities=sc["opacities"], scales=None if pre_cov else sc["scales"],
                  rotations=None if pre_cov else sc["rotations"],
                  cov3D_precomp=_cov6(sc["scales"], sc["rotations"], mod) if pre_cov else None)
    gmask = tuple(int(v) for v in rng.integers(0, 2, 5))
    if not any(gmask):
        gmask = (1, 0, 0, 0, 0)
    # VEGS_FUZZ_HIP_FLAGS (campaigns): e.g. 256 deterministic backward, 512 scan binning, 2048 segment rounds forced on;
    # VEGS_FUZZ_FLAGS: switches that change WHAT is computed, set on both sides (e.g. 32768 = the reference's full tile lists)
    tp._check_against_oracle(inputs, cam, bg, deg, mod, dev, seed=seed, gmask=gmask, M=M,
                             flags=int(os.environ.get("VEGS_FUZZ_FLAGS", "0")),
                             hip_flags=int(os.environ.get("VEGS_FUZZ_HIP_FLAGS", "0")))
