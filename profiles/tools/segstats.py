"""Segment / strip statistics of one forward of the headline scene (run on the GPU box):
needed vs total segments, relevant entries per (segment, strip), chunk fill of k_seg_bwd."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vegs_amd import _capi, harness, scenes  # noqa: E402

dev = torch.device('cuda:0')
sc, deg = scenes.scene_street(P=2_000_000, length=250.0, sh_degree=3, seed=2)
T_ = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in sc.items()}
al = lambda v: (v + 255) // 256 * 256
for x in (0.0, 40.0):
    cam = scenes.kitti_camera(x, 0.3, 1376, 376)
    pkg = harness.render(cam, T_, deg, torch.zeros(3, device=dev))
    fn = pkg['render'].grad_fn
    b = fn.buffers[1].cpu().numpy()
    T = 86 * 24
    cap = fn.binning_capacity
    S = cap // 256 + T
    o_seg = al(T * 8); o_need = o_seg + al((((T + 1 + 63) // 64 * 64) + 4 * S) * 4); o_pl = o_need + al(T * 4)
    o_tb = o_pl + al(max(cap, 1) * 4); o_part = o_tb + al(S * 256 * 4); o_mask = o_part + al(S * 13 * 256 * 4)
    seg_off = b[o_seg:o_seg + (T + 1) * 4].view(np.uint32).astype(np.int64)
    need = b[o_need:o_need + T * 4].view(np.uint32).astype(np.int64)
    nseg = np.diff(seg_off)
    masks = b[o_mask:o_mask + S * 16 * 8].view(np.uint64).reshape(S, 4, 4)
    pop = np.zeros((S, 4), np.int64)
    for part in range(4):
        m = masks[:, :, part]
        pop += np.array([bin(int(v)).count('1') for v in m.reshape(-1)]).reshape(S, 4)
    needed_ids = np.concatenate([np.arange(seg_off[t], seg_off[t] + need[t]) for t in range(T)])
    pn = pop[needed_ids]
    chunks = (pn + 63) // 64
    print('view x', x, 'R', fn.num_rendered, 'segments', nseg.sum(), 'needed', need.sum(),
          'F', _capi.count_fragments(fn, 376, 1376, dev))
    print('  relevant entries per (needed segment, strip): mean %.1f  median %d  p90 %d ; all segs mean %.1f' %
          (pn.mean(), np.median(pn), np.percentile(pn, 90), pop[:nseg.sum()].mean()))
    print('  k_seg_bwd chunks per (segment, strip): mean %.2f ; lane fill %.3f ; chunks total %d' %
          (chunks.mean(), pn.sum() / max(chunks.sum() * 64, 1), chunks.sum()))
