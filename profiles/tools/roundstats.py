"""How many list segments would a multi-round forward compute?  Per tile: n = segments of its list, need = segments
any pixel needs (vr_export_needed).  A schedule is a list of cumulative limits L0 < L1 < ...: round r computes the
segments below L_r of every tile still alive; a tile stops after the first round that covers its need (or its list).
Headline scene and the dense one (discs x3).  python profiles/tools/roundstats.py"""
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
for disc in (1.0, 3.0):
    sc, deg = scenes.scene_street(P=2_000_000, length=250.0, sh_degree=3, seed=2)
    sc["scales"] = (sc["scales"] * disc).astype(np.float32)
    T_ = {k: torch.tensor(v, device=dev) for k, v in sc.items()}
    tot = {}
    for x in (0.0, 40.0, 70.0):
        cam = scenes.kitti_camera(x, 0.3, W, H)
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
        n = (rgn[:, 1] - rgn[:, 0] + 255) // 256
        need = need_t.cpu().numpy().astype(np.int64)
        print(f"discs x{disc} view {x}: R {fn.num_rendered} segments {n.sum()} needed {need.sum()} | n percentiles 50/90/99/max "
              f"{np.percentile(n, [50, 90, 99]).astype(int)} {n.max()} | need 50/90/99/max {np.percentile(need, [50, 90, 99]).astype(int)} {need.max()}")
        scheds = {"all at once": [10**9], "8 | rest": [8, 10**9], "4 | 16 | rest": [4, 16, 10**9], "6 | 24 | rest": [6, 24, 10**9],
                  "8 | 32 | rest": [8, 32, 10**9], "4 | 12 | 36 | rest": [4, 12, 36, 10**9], "6 | 18 | 54 | 162 | rest": [6, 18, 54, 162, 10**9],
                  "doubling from 4": [4 * 2 ** k for k in range(9)] + [10**9]}
        for name, sched in scheds.items():
            comp, alive = 0, []
            done = np.zeros(T, bool)
            prev = 0
            for L in sched:
                c = np.minimum(n, L)
                comp += int((c - np.minimum(n, prev))[~done].sum())
                alive.append(int((~done).sum()))
                done |= (c >= need) | (c == n)
                prev = L
                if done.all():
                    break
            tot.setdefault(name, []).append((comp, len(alive), alive))
        # tile-dependent first limit: max(6, n / d), then + 3 x 6, then the rest
        for d in (2, 3, 4, 6, 8):
            L0 = np.maximum(6, n // d)
            c0 = np.minimum(n, L0)
            done0 = (c0 >= need) | (c0 == n)
            c1 = np.minimum(n, c0 + np.maximum(8 + c0 // 2, 18))
            done1 = done0 | (c1 >= need) | (c1 == n)
            comp = int(c0.sum() + (c1 - c0)[~done0].sum() + (n - c1)[~done1].sum())
            tot.setdefault(f"max(6, n/{d}) | +max(8+lo/2,18) | rest", []).append((comp, 3, [T, int((~done0).sum()), int((~done1).sum())]))
    for name, v in tot.items():
        print(f"  {name:28s} computed {np.mean([c for c, _, _ in v]):9.0f}  rounds {max(r for _, r, _ in v)}  tiles alive per round {v[1][2]}")
