"""CPU: the TIGHT TILE LISTS of the checker (oracle/vr_oracle.c: tile_reachable; vegs_amd/csrc/vr_device.h holds the same
functions) against its own full-rectangle lists (flag VR_FLAG_FULL_TILE_LISTS = the reference's emission rule):
the tight list of every tile is the full list minus pairs that provably cannot contribute -- checked by brute force over the
tile's pixel centres in float64 --, radii are identical, and images and gradients change by rounding only (the sums are
grouped by 256-entry list segments, which now start at other entries: ~5e-7)."""
import numpy as np

from helpers import oracle_cam

FULL = 32768


def _scene():
    from vegs_amd import scenes
    sc, deg = scenes.scene_street(P=30000, length=60.0, sh_degree=1, seed=41)
    sc["opacities"][:300] = 0.002            # below 1/255: such a splat reaches no tile at all
    sc["scales"][300:600] *= 8.0             # rectangles of more than 64 tiles: tested in cells of k x k tiles
    cam = scenes.kitti_camera(0.0, 0.3, 688, 188)
    return sc, deg, cam


def test_tight_lists_drop_only_pairs_that_cannot_contribute():
    from oracle import oracle as orc
    sc, deg, cam = _scene()
    args = (sc["means3D"], sc["shs"], None, sc["opacities"], sc["scales"], sc["rotations"], None)
    o_t, st_t = orc.forward(oracle_cam(cam, [0.1, 0.2, 0.3], deg), *args)
    o_f, st_f = orc.forward(oracle_cam(cam, [0.1, 0.2, 0.3], deg, flags=FULL), *args)
    assert st_t["R"] < st_f["R"] and np.array_equal(o_t["radii"], o_f["radii"])
    for k in ("color", "depth", "cov_quat", "cov_scale", "alpha"):
        # the same fragments in the same order, summed in other 256-entry groups: rounding, three orders below the 1e-4 bar
        assert np.abs(o_t[k] - o_f[k]).max() <= 1e-6 * max(1.0, float(np.abs(o_f[k]).max())), k
    H, W = cam.image_height, cam.image_width
    gx = (W + 15) // 16
    xy, co = st_f["xy"].astype(np.float64), st_f["conic_op"].astype(np.float64)
    dropped = kept = 0
    for t in range(st_f["ranges"].shape[0]):
        full = st_f["point_list"][st_f["ranges"][t, 0]:st_f["ranges"][t, 1]]
        tight = st_t["point_list"][st_t["ranges"][t, 0]:st_t["ranges"][t, 1]]
        keep = np.isin(full, tight)
        assert np.array_equal(full[keep], tight)                     # a sub-sequence: same order (tile, depth, id)
        gone = full[~keep]
        kept += int(keep.sum())
        if gone.size == 0:
            continue
        dropped += gone.size
        ty, tx = divmod(t, gx)
        px = (tx * 16 + np.arange(16))[None, None, :]
        py = (ty * 16 + np.arange(16))[None, :, None]
        dx, dy = xy[gone, 0][:, None, None] - px, xy[gone, 1][:, None, None] - py
        A, B, Cc, op = (co[gone, k][:, None, None] for k in range(4))
        power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
        inside = (px < W) & (py < H)
        alpha = np.where((power <= 0) & inside, np.minimum(0.99, op * np.exp(np.minimum(power, 0))), 0.0)
        assert alpha.max() < 1.0 / 255.0 * (1 - 1e-3), (t, float(alpha.max()))     # nowhere near the blend rule's threshold
    assert dropped > 0.15 * (dropped + kept)                          # a street view: a sizeable share of the pairs
    # rectangles of more than 64 tiles are tested in cells of k x k tiles: they lose tiles too, never gain any, and the
    # count the preprocess step announces is the number of list entries the binning step writes for the Gaussian
    big = (st_f["tiles_touched"] > 64)
    assert big.any() and (st_t["tiles_touched"][big] <= st_f["tiles_touched"][big]).all()
    assert (st_t["tiles_touched"][big] < st_f["tiles_touched"][big]).any()
    assert np.array_equal(np.bincount(st_t["point_list"], minlength=st_t["tiles_touched"].size), st_t["tiles_touched"])
    faint = np.arange(300)
    assert (st_t["tiles_touched"][faint] == 0).all() and (st_f["tiles_touched"][faint[st_f["radii"][faint] > 0]] > 0).all()


def test_tight_lists_leave_gradients_unchanged():
    from oracle import oracle as orc
    sc, deg, cam = _scene()
    args = (sc["means3D"], sc["shs"], None, sc["opacities"], sc["scales"], sc["rotations"], None)
    rng = np.random.default_rng(2)
    H, W = cam.image_height, cam.image_width
    g = [rng.normal(size=s).astype(np.float32) for s in [(3, H, W), (1, H, W), (4, H, W), (3, H, W), (1, H, W)]]
    grads = []
    for flags in (0, FULL):
        oc = oracle_cam(cam, [0, 0, 0], deg, flags=flags)
        _, st = orc.forward(oc, *args)
        grads.append(orc.backward(oc, st, *g))
    from helpers import assert_grad_close
    for k in ("means3D", "means2D", "shs", "opacities", "scales", "rotations"):
        if grads[0][k] is not None:       # (the dropped pairs contribute exact zeros; what moves is the forward's rounding)
            assert_grad_close("tight vs full " + k, grads[0][k], grads[1][k], rtol=1e-4, floor=1e-7)
