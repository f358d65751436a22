import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vegs_amd import harness, scenes, _capi
dev = torch.device('cuda:0')
sc, deg = scenes.scene_street(P=2_000_000, length=250.0, sh_degree=3, seed=2)
T_ = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in sc.items()}
for x in (0.0, 40.0):
    cam = scenes.kitti_camera(x, 0.3, 1376, 376)
    pkg = harness.render(cam, T_, deg, torch.zeros(3, device=dev))
    fn = pkg['render'].grad_fn
    b = fn.buffers[1].cpu().numpy()
    T = 86*24
    al = lambda v: (v + 255)//256*256
    o_seg = al(T*8); o_need = o_seg + al((T+1)*4)
    ranges = b[:T*8].view(np.int32).reshape(T,2)
    seg_off = b[o_seg:o_seg+(T+1)*4].view(np.uint32)
    need = b[o_need:o_need+T*4].view(np.uint32)
    nseg = np.diff(seg_off.astype(np.int64))
    print('view x', x, 'R', fn.num_rendered, 'total segs', nseg.sum(), 'needed', need.sum(), 'max nseg', nseg.max(), 'max needed', need.max(),
          'tiles with needed<nseg', (need < nseg).sum(), 'F', _capi.count_fragments(fn, 376, 1376, dev))
    print(' needed histogram', np.percentile(need, [50, 90, 99, 100]), ' nseg', np.percentile(nseg, [50,90,99,100]))
