"""Several views in flight on separate HIP streams (vegs_amd/views.py): results must not depend on the number of streams.
The consumers this serves are the reference's evaluation / video loops (train.py:338-508, render_video.py:162,202) and
view batches with summed gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _scene(dev, P=300000):
    from vegs_amd import harness, scenes
    sc, deg = scenes.scene_street(P=P, length=60.0, sh_degree=2, seed=41)
    cams = [scenes.kitti_camera(6.0 * s, y, 688, 188) for s in range(4) for y in (0.3, -0.3)]
    T = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in sc.items()}
    cam_ts = [harness.cam_tensors(c, dev) for c in cams]
    return sc, deg, cams, cam_ts, T


@pytest.mark.parametrize("streams", [2, 3])
def test_render_sequence_is_bit_identical_to_one_stream(streams, dev):
    """Forward-only frames through render_sequence with 2 and 3 views in flight == the same frames rendered one after
    the other on the caller's stream, bit for bit, in the order of the cameras; the first frame also equals the oracle."""
    from helpers import OUT_NAMES, oracle_cam
    from oracle import oracle as orc
    from vegs_amd import harness, views
    sc, deg, cams, cam_ts, T = _scene(dev)
    bg = torch.zeros(3, device=dev)
    idx = list(range(len(cams))) * 2       # 16 frames: every stream gets several views, workspaces are recycled
    keys = ("render", "render_depth", "render_cov_quat", "render_cov_scale", "alpha", "radii")

    def frame(i):
        return harness.render(cams[i], T, deg, bg, cam_t=cam_ts[i])

    with torch.no_grad():
        want = [{k: frame(i)[k].cpu().numpy() for k in keys} for i in idx[:len(cams)]]
        got = []
        for pkg in views.render_sequence(idx, frame, dev, streams=streams):
            # consumer work on the caller's stream, as an evaluation loop has it (a metric per frame)
            got.append({k: pkg[k].cpu().numpy() for k in keys})
            _ = float((pkg["render"] ** 2).mean())
    assert len(got) == len(idx)
    for n, (i, g) in enumerate(zip(idx, got)):
        for k in keys:
            assert np.array_equal(g[k], want[i][k]), (n, i, k)
    o, _ = orc.forward(oracle_cam(cams[0], [0, 0, 0], deg), sc["means3D"], sc["shs"], None, sc["opacities"], sc["scales"],
                       sc["rotations"], None)
    for k, n in zip(keys[:5], OUT_NAMES):
        assert np.array_equal(got[0][k], o[n]), n


def test_view_batch_gradients_do_not_depend_on_the_streams(dev):
    """A batch of 6 views, forward + backward each, gradients summed into the shared parameters by autograd:
    VR_FLAG_DETERMINISTIC makes every view's gradient bit-reproducible and autograd adds the views in submission order,
    so one stream and two streams must agree BIT FOR BIT; in the default (atomic) mode they agree like two runs do."""
    from helpers import assert_grad_close
    from vegs_amd import harness, rasterizer, views
    sc, deg, cams, cam_ts, T = _scene(dev, P=120000)
    bg = torch.zeros(3, device=dev)
    rng = np.random.default_rng(3)
    H, W = 188, 688
    gouts = [[torch.tensor(rng.normal(size=s).astype(np.float32), device=dev) for s in [(3, H, W), (4, H, W), (3, H, W)]]
             for _ in range(6)]
    names = ("means3D", "shs", "opacities", "scales", "rotations")

    def one(v):
        pkg = harness.render(cams[v], T, deg, bg, cam_t=cam_ts[v])
        torch.autograd.backward([pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]], gouts[v])
        return int(pkg["radii"].gt(0).sum())

    def batch(streams):
        for k in names:
            T[k].grad = None
        vis = views.view_batch(range(6), one, dev, streams=streams)
        torch.cuda.current_stream().synchronize()
        return vis, [T[k].grad.detach().cpu().numpy().copy() for k in names]

    with rasterizer.flags(rasterizer.FLAG_DETERMINISTIC):
        v1, g1 = batch(1)
        v2, g2 = batch(2)
        v3, g3 = batch(3)
    assert v1 == v2 == v3 and min(v1) > 1000
    for n, a, b, c in zip(names, g1, g2, g3):
        assert np.array_equal(a, b) and np.array_equal(a, c), n
        assert np.isfinite(a).all() and np.abs(a).max() > 0
    _, h1 = batch(1)
    _, h2 = batch(2)
    for n, a, b in zip(names, h1, h2):
        assert_grad_close(f"{n}: two streams vs one", b, a, rtol=1e-3, floor=2e-6)
