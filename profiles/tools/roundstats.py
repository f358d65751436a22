import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from vegs_amd import _capi, harness, scenes
dev = torch.device('cuda:0')
sc, deg = scenes.scene_street(P=2_000_000, length=250.0, sh_degree=3, seed=2)
T_ = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in sc.items()}
al = lambda v: (v + 255) // 256 * 256
for x in (0.0, 40.0, 70.0):
    cam = scenes.kitti_camera(x, 0.3, 1376, 376)
    pkg = harness.render(cam, T_, deg, torch.zeros(3, device=dev))
    fn = pkg['render'].grad_fn
    b = fn.buffers[1].cpu().numpy()
    T = 86 * 24
    cap = fn.binning_capacity
    S = cap // 256 + T
    o_seg = al(T * 8); o_need = o_seg + al((((T + 1 + 63) // 64 * 64) + 4 * S) * 4)
    seg_off = b[o_seg:o_seg + (T + 1) * 4].view(np.uint32).astype(np.int64)
    need = b[o_need:o_need + T * 4].view(np.uint32).astype(np.int64)
    nseg = np.diff(seg_off)
    print('view', x, 'tiles', T, 'segs', nseg.sum(), 'needed', need.sum(), 'max nseg', nseg.max(), 'max need', need.max())
    print('  need percentiles', np.percentile(need, [50, 75, 90, 99]), 'nseg percentiles', np.percentile(nseg, [50, 75, 90, 99]))
    for sched in ([2, 6, 14, 30, 62, 126, 10**9], [4, 16, 10**9], [4, 12, 36, 10**9], [3, 8, 20, 50, 10**9], [8, 10**9]):
        comp = 0
        for t in range(T):
            for L in sched:
                c = min(nseg[t], L)
                if c >= need[t] or c == nseg[t]:
                    comp += c
                    break
        print('  schedule', sched[:-1], 'computed', comp, 'rounds', len(sched))
