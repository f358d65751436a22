"""Do two side streams really run side by side?  Eight fresh ViewStreams(2) in a row, an 8-view training batch each
(headline scene), for a given stream priority:  python profiles/tools/stream_queues.py [-1|0]
HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4).  A normal-priority side stream that
lands on the default stream's queue sits behind the barrier packets of autograd's gradient accumulation there (which
wait for the OTHER side stream): that batch serialises (1.61 instead of 1.36-1.41 ms per view, one instance in eight --
also with GPU_MAX_HW_QUEUES=8).  High-priority streams have queues of their own: all eight instances overlap."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, argparse
from vegs_amd import _capi, rasterizer, views
args = argparse.Namespace(workload="c3", gaussians=0, width=1376, height=376)
device = torch.device("cuda", 0); torch.cuda.set_device(device); _capi.load()
sc, deg, cams, P = bench.build_workload(args)
rasterizer.needed_hints(False)
wl = bench.prepare(sc, deg, cams, device, np.random.default_rng(77), count=False)
prio = int(sys.argv[1]) if len(sys.argv) > 1 else -1
orig = views.ViewStreams.__init__
def init(self, device, n=2, priority=prio):
    orig(self, device, n, priority)
views.ViewStreams.__init__ = init
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "priority", prio)
for rep in range(8):
    step = bench.make_step(wl, 0, 1, 8, False, mode="train", streams=2)
    for i in range(2): step(i)
    dt, done, runs = bench.timed_median(step, 4, 1, 3)
    print("instance", rep, round(dt / len(done) * 1e3, 4), flush=True)
