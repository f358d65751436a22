"""Soak of the multi-rank training step on ONE GPU: `world` ranks (gloo handles, direct or factored exchange), `iters`
iterations with a densify every 10 and an opacity reset at 30, the digest of EVERY rank's whole state (parameters, Adam
moments, step counters, statistics) compared after every iteration.
    python profiles/tools/dist_soak.py [world=4] [iters=60] [exchange=direct]"""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r"""
import hashlib, os, sys, numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from vegs_amd import dist as vdist, iteration, rasterizer, scenes, harness
rank, world, local = vdist.init_from_env()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
sc, deg = scenes.scene_street(P=120000, length=80.0, sh_degree=3, seed=31)
H, W = 188, 688
cams = [scenes.kitti_camera(3.0 * (i // 2), 0.3 if i %% 2 == 0 else -0.3, W, H) for i in range(16)]
rng = np.random.default_rng(17)
gts = [torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32), device=dev) for _ in range(4)]
normals = [torch.tensor(rng.normal(size=(3, H, W)).astype(np.float32), device=dev) for _ in range(4)]
sch = iteration.Schedule(extent=20.0, densify_from_iter=5, densification_interval=10, opacity_reset_interval=30, densify_grad_threshold=2e-5)
rasterizer.set_flags(rasterizer.get_flags() | rasterizer.FLAG_DETERMINISTIC)
tr = iteration.Trainer(sc, dev, n_boxes=3, fused=True, box_points=2000, factored_sh=True, lrs=iteration.REFERENCE_LRS,
                       optimise_boxes=True, world=world, rank=rank, exchange=%(exchange)r, schedule=sch, seed=5)
bg = torch.zeros(3, device=dev)
out = []
for it in range(%(iters)d):
    v = vdist.view_for_rank(it, rank, world, len(cams))
    loss, _, _ = tr.step(cams[v], harness.cam_tensors(cams[v], dev), deg, bg, gts[it %% 4], normals[it %% 4])
    h = hashlib.sha256()
    for k, t in sorted(tr.state_tensors().items()):
        h.update(k.encode()); h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    out.append((h.hexdigest()[:16], int(tr.p["xyz"].shape[0]), float(loss)))
torch.save(out, os.path.join(%(out)r, "soak%%d.pt" %% rank))
torch.distributed.barrier(); torch.distributed.destroy_process_group()
print("RANK_OK", rank)
"""
world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
exchange = sys.argv[3] if len(sys.argv) > 3 else "direct"
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
script = WORKER % dict(root=ROOT, iters=iters, exchange=exchange, out=out)
procs = []
for r in range(world):
    env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VEGS_DIST_BACKEND="gloo")
    procs.append(subprocess.Popen([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
outs = [p.communicate(timeout=1500)[0] for p in procs]
if not all("RANK_OK" in o for o in outs):
    print("\n".join(o[-2000:] for o in outs)); sys.exit(1)
import torch
R = [torch.load(os.path.join(out, f"soak{r}.pt")) for r in range(world)]
bad = [it for it in range(iters) if len({R[r][it][0] for r in range(world)}) != 1]
sizes = [R[0][it][1] for it in range(iters)]
print(f"{world} ranks, {iters} iterations, exchange {exchange}: iterations with differing rank states: {bad if bad else 'none'}; "
      f"static model {sizes[0]} -> {sizes[-1]} Gaussians ({sum(1 for a, b in zip(sizes, sizes[1:]) if a != b)} densifications); "
      f"loss {R[0][0][2]:.4f} -> {R[0][-1][2]:.4f}")
sys.exit(1 if bad else 0)
