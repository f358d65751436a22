"""Debug driver for the direct exchange: `python xgmi_debug.py <world> <reserve 0|1> <sizes...>` spawns the ranks on one GPU."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from vegs_amd import dist as vdist, xgmi
rank, world, local = vdist.init_from_env()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ex = xgmi.DirectExchange(rank, world, dev)
sizes = %(sizes)r
if %(reserve)d:
    ex.reserve(11 * max(sizes) + 64, 3 * (max(sizes) + 17) + 16)
def data(it, r, P):
    rng = np.random.default_rng(1000 * it + r)
    g = [rng.normal(size=s).astype(np.float32) for s in ((P, 3), (P, 1), (P, 3), (P, 4))]
    return g, rng.normal(size=(1, P + 17, 3)).astype(np.float32), rng.normal(size=(1, 3)).astype(np.float32)
for it, P in enumerate(sizes):
    g, f, c = data(it, rank, P)
    ps = [torch.nn.Parameter(torch.zeros(x.shape, device=dev)) for x in g]
    for p, x in zip(ps, g):
        p.grad = torch.tensor(x, device=dev)
    try:
        F, C = ex.exchange(ps, torch.tensor(f, device=dev), torch.tensor(c, device=dev), 1)
        ex.check()
    except Exception as e:
        fl = ex.win[:4096].view(torch.int64) if ex.win is not None else None
        print("rank", rank, "it", it, "ERR", e, "flags a", fl[0:8*world:8].tolist(), "b", fl[128:128+8*world:8].tolist(), "g0", fl[256:256+8*world:8].tolist(), "g1", fl[384:384+8*world:8].tolist(), flush=True)
        raise
    allr = [data(it, r, P) for r in range(world)]
    for k, p in enumerate(ps):
        want = allr[0][0][k].copy()
        for r in range(1, world):
            want = want + allr[r][0][k]
        want = want * np.float32(1.0 / world)
        got = p.grad.cpu().numpy()
        if not np.array_equal(got, want):
            bad = np.nonzero((got != want).reshape(-1))[0]
            print("rank", rank, "it", it, "seg", k, "mismatch", bad.size, "of", got.size, "first", bad[:4], "last", bad[-4:], flush=True)
    for r in range(world):
        if not np.array_equal(F[r].cpu().numpy(), allr[r][1][0]):
            print("rank", rank, "it", it, "gather mismatch from", r, flush=True)
    print("rank", rank, "it", it, "P", P, "done", flush=True)
torch.cuda.synchronize()
torch.distributed.barrier()
ex.close()
print("RANK_OK", rank)
"""
world, reserve = int(sys.argv[1]), int(sys.argv[2])
sizes = [int(x) for x in sys.argv[3:]]
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
script = WORKER % dict(root=ROOT, sizes=sizes, reserve=reserve)
procs = []
for r in range(world):
    env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VEGS_DIST_BACKEND="gloo")
    procs.append(subprocess.Popen([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
for r, p in enumerate(procs):
    out = p.communicate(timeout=300)[0]
    print("\n".join(l for l in out.splitlines() if l.startswith("rank") or "RANK_OK" in l or "Error" in l))
