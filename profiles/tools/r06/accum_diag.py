"""(round 6) Diagnosis of tests/test_gpu_accumulate.py's two-stream case: deterministic-mode gradient sums of six views, autograd's
accumulation on one stream against two streams (vegs_amd.views.view_batch), each twice; and view by view.
   [VEGS_LIB=...] python profiles/tools/r06/accum_diag.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_accumulate import _scene, _gouts, _batch
from vegs_amd import _capi, harness, rasterizer
_capi.load()
dev = torch.device("cuda", 0)
sc, deg, cams, cam_ts = _scene(dev)
T = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in sc.items()}
bg = torch.zeros(3, device=dev)
gouts = _gouts(dev, 6)


def one(v):
    pkg = harness.render(cams[v], T, deg, bg, cam_t=cam_ts[v])
    torch.autograd.backward([pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]], gouts[v])
    return pkg["viewspace_points"].grad


def cmp(a, b, what):
    bad = []
    for k in a:
        if not torch.equal(a[k], b[k]):
            rows = (a[k] != b[k]).reshape(a[k].shape[0], -1).any(dim=1)
            bad.append((k, int(rows.sum()), float((a[k] - b[k]).abs().max()), float(a[k].abs().max())))
    print(what, "EQUAL" if not bad else bad, flush=True)


w1a, _ = _batch(T, one, 6, dev, False, 1)
w1b, _ = _batch(T, one, 6, dev, False, 1)
cmp(w1a, w1b, "one stream, run 1 vs run 2:")
w2a, _ = _batch(T, one, 6, dev, False, 2)
w2b, _ = _batch(T, one, 6, dev, False, 2)
cmp(w2a, w2b, "two streams, run 1 vs run 2:")
cmp(w1a, w2a, "one stream vs two streams:")
g2, _ = _batch(T, one, 6, dev, True, 2)
cmp(w1a, g2, "one stream (autograd) vs two streams IN PLACE:")
# single views, twice each, in the deterministic mode
for v in range(6):
    r = []
    for rep in range(2):
        for t in T.values():
            t.grad = None
        with rasterizer.flags(rasterizer.get_flags() | 256):
            one(v)
        torch.cuda.synchronize()
        r.append({k: t.grad.clone() for k, t in T.items()})
    cmp(r[0], r[1], f"view {v} alone, run 1 vs run 2:")
