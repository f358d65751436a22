import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vegs_amd import harness, scenes
dev = torch.device('cuda:0')
sc, deg = scenes.scene_random(P=5000, sh_degree=1, seed=1, scale=0.05)
cam = scenes.camera_c1(160, 96)
for name, mod in [("nan_scale", lambda t: t["scales"].__setitem__((slice(0, 50),), float('nan'))),
                  ("inf_scale", lambda t: t["scales"].__setitem__((slice(0, 50),), float('inf'))),
                  ("huge_scale", lambda t: t["scales"].__setitem__((slice(0, 50),), 1e20)),
                  ("nan_mean", lambda t: t["means3D"].__setitem__((slice(0, 50),), float('nan'))),
                  ("inf_mean", lambda t: t["means3D"].__setitem__((slice(0, 50), 0), float('inf'))),
                  ("nan_rot", lambda t: t["rotations"].__setitem__((slice(0, 50),), float('nan'))),
                  ("zero_rot", lambda t: t["rotations"].__setitem__((slice(0, 50),), 0.0)),
                  ("nan_opac", lambda t: t["opacities"].__setitem__((slice(0, 50),), float('nan'))),
                  ("neg_opac", lambda t: t["opacities"].__setitem__((slice(0, 50),), -1.0)),
                  ("big_opac", lambda t: t["opacities"].__setitem__((slice(0, 50),), 50.0)),
                  ("nan_sh", lambda t: t["shs"].__setitem__((slice(0, 50),), float('nan')))]:
    t = {k: torch.tensor(v.copy(), device=dev) for k, v in sc.items()}
    mod(t)
    t = {k: v.requires_grad_(True) for k, v in t.items()}
    pkg = harness.render(cam, t, deg, torch.zeros(3, device=dev))
    (pkg["render"].nan_to_num().sum() + pkg["render_cov_scale"].nan_to_num().sum()).backward()
    torch.cuda.synchronize()
    fn = pkg["render"].grad_fn
    print(name, "R", fn.num_rendered, "V", fn.num_visible, "finite img", bool(torch.isfinite(pkg["render"]).all()),
          "radii max", int(pkg["radii"].max()), flush=True)
print("done")
