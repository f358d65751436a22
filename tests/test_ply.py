"""CPU: Gaussian-model PLY layout of the reference (scene/gaussian_model.py:182-259)."""
import numpy as np
import pytest

from vegs_amd import ply


def _model(n=37, deg=3, seed=0):
    rng = np.random.default_rng(seed)
    K = (deg + 1) ** 2
    return dict(xyz=rng.normal(size=(n, 3)), features_dc=rng.normal(size=(n, 1, 3)), features_rest=rng.normal(size=(n, K - 1, 3)),
                opacity=rng.normal(size=(n, 1)), scaling=rng.normal(size=(n, 3)), rotation=rng.normal(size=(n, 4)))


def test_layout_is_the_reference_layout(tmp_path):
    m = {k: v.astype(np.float32) for k, v in _model().items()}
    p = str(tmp_path / "point_cloud.ply")
    ply.save_ply(p, **m)
    raw = open(p, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [l.split()[2] for l in lines[3:]]
    assert all(l.startswith("property float ") for l in lines[3:])
    assert names == (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)]
                     + ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])   # :182-194
    rows = np.frombuffer(body, "<f4").reshape(37, 62)
    assert np.array_equal(rows[:, :3], m["xyz"]) and not rows[:, 3:6].any()
    # channel-major SH storage: transpose(1, 2).flatten(start_dim=1)  (:201-202)
    assert np.array_equal(rows[:, 6:9], m["features_dc"][:, 0, :])
    assert np.array_equal(rows[:, 9 + 15 * 1 + 4], m["features_rest"][:, 4, 1])
    assert np.array_equal(rows[:, 54], m["opacity"][:, 0]) and np.array_equal(rows[:, 58:62], m["rotation"])


@pytest.mark.parametrize("deg", [0, 1, 3])
def test_round_trip(tmp_path, deg):
    m = {k: v.astype(np.float32) for k, v in _model(101, deg, 5).items()}
    p = str(tmp_path / "m.ply")
    ply.save_ply(p, **m)
    back = ply.load_ply(p, max_sh_degree=deg)
    for k in m:
        assert back[k].shape == m[k].shape and np.array_equal(back[k], m[k]), k
    with pytest.raises(ValueError):
        ply.load_ply(p, max_sh_degree=2)                    # the reference asserts the f_rest count (:232)


def test_reader_goes_by_property_name_and_type(tmp_path):
    """files from other writers: comments, permuted property order, a double and a uchar column"""
    rng = np.random.default_rng(1)
    n = 5
    names = ply.attribute_names(3, 0)
    order = list(reversed(names)) + ["extra"]
    dt = np.dtype([(k, "<f8" if k == "x" else "<f4") for k in order[:-1]] + [("extra", "u1")])
    data = np.zeros(n, dt)
    for k in names:
        data[k] = rng.normal(size=n)
    p = str(tmp_path / "other.ply")
    with open(p, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\ncomment made elsewhere\nelement vertex 5\n")
        for k in order:
            f.write(f"property {'double' if k == 'x' else ('uchar' if k == 'extra' else 'float')} {k}\n".encode())
        f.write(b"element face 0\nproperty list uchar int vertex_indices\nend_header\n")
        data.tofile(f)
    back = ply.load_ply(p, max_sh_degree=0)
    assert np.allclose(back["xyz"][:, 0], data["x"]) and np.array_equal(back["rotation"][:, 3], data["rot_3"])
    assert back["features_rest"].shape == (5, 0, 3)
