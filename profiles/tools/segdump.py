"""Per-tile list segments n and needed segments of the 16 bench cameras -> gpurun_out/segdump.npz (for offline what-if
studies of forward schedules).  python profiles/tools/segdump.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vegs_amd import _capi, harness, scenes  # noqa: E402

dev = torch.device("cuda:0")
H, W = 376, 1376
T = ((W + 15) // 16) * ((H + 15) // 16)
sc, deg = scenes.scene_street(P=2_000_000, length=250.0, sh_degree=3, seed=2)
T_ = {k: torch.tensor(v, device=dev) for k, v in sc.items()}
out = {}
for s in range(8):
    for j, y in enumerate((0.3, -0.3)):
        cam = scenes.kitti_camera(10.0 * s, y, W, H)
        T2 = {k: v.clone().requires_grad_(True) for k, v in T_.items()}
        pkg = harness.render(cam, T2, deg, torch.zeros(3, device=dev))
        fn = pkg["render"].grad_fn
        saved = _capi.saved_of(fn)
        need_t = torch.zeros(T, dtype=torch.int32, device=dev)
        rg = torch.zeros((T, 2), dtype=torch.int32, device=dev)
        pl = torch.zeros(max(fn.num_rendered, 1), dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        _capi.check(_capi.load().vr_export_needed(C.byref(saved), H, W, need_t.data_ptr(), st))
        _capi.check(_capi.load().vr_debug_export_binning(C.byref(saved), H, W, pl.data_ptr(), rg.data_ptr(), st))
        torch.cuda.synchronize()
        rgn = rg.cpu().numpy().astype(np.int64)
        out[f"n_{2 * s + j}"] = (rgn[:, 1] - rgn[:, 0] + 255) // 256
        out[f"need_{2 * s + j}"] = need_t.cpu().numpy().astype(np.int64)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/segdump.npz", **out)
print("ok")
