"""GPU (-m gpu): the gradient exchange on the RCCL backend ("nccl" on ROCm).  A 1-GPU box cannot host two ranks,
so this runs ONE rank through the real RCCL code path (communicator creation, the ReduceOp.AVG probe, the in-place
per-tensor collectives on the GPU); the multi-rank arithmetic is covered by tests/test_dist_gloo.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from vegs_amd import dist as vdist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)      # init_from_env only initialises for WORLD_SIZE > 1
assert dist.get_backend() == "nccl"
dev = torch.device("cuda", 0)
ps = [torch.nn.Parameter(torch.randn(5000, k, device=dev)) for k in (3, 48, 1, 3, 4)] + [torch.nn.Parameter(torch.randn(7, device=dev))]
want = []
for p in ps:
    p.grad = torch.randn_like(p)
    want.append(p.grad.clone())
vdist.allreduce_grads(ps, world=2, flat_bucket_bytes=4096)      # pretend two views: exercises the collectives
avg = vdist._avg_supported(None, dev)
for p, w in zip(ps, want):
    ref = w if avg else w * 0.5                                   # AVG over the single rank / SUM then 1/world
    assert torch.allclose(p.grad, ref), (avg, (p.grad - ref).abs().max())
g2d = torch.randn(5000, 3, device=dev); vis = torch.rand(5000, device=dev) > 0.5; radii = torch.randint(0, 9, (5000,), device=dev, dtype=torch.int32)
s, d, r = vdist.allreduce_densification_stats(g2d, vis, radii)
assert torch.equal(r, radii) and torch.allclose(d[:, 0], vis.float())
dist.destroy_process_group()
print("RCCL_OK avg=%%s" %% avg)
"""


def test_gradient_exchange_on_rccl_single_rank():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29653")
    script = SCRIPT % ROOT
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout + r.stderr
