"""Two (or more) views in flight on separate HIP streams: views/s of the headline scene for forward-only rendering and
for view batches, against the one-stream rate.  Run on the GPU box:  python profiles/tools/streams_probe.py"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from vegs_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3"); ap.add_argument("--gaussians", type=int, default=0)
ap.add_argument("--width", type=int, default=1376); ap.add_argument("--height", type=int, default=376)
args = ap.parse_args()
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
_capi.load()
sc, deg, cams, P = bench.build_workload(args)
wl = bench.prepare(sc, deg, cams, device, np.random.default_rng(1234), count=False)
for mode, vps in (("forward", 16), ("train", 8), ("noglue", 8)):
    for streams in (1, 2, 3, 4):
        step = bench.make_step(wl, 0, 1, vps, mode=mode, streams=streams)
        for i in range(2):
            step(i)
        dt, views, runs = bench.timed_median(step, 4, 1, 3, first=2)
        print(f"{mode:8s} views/step {vps:2d} streams {streams}: {dt / len(views) * 1e3:7.4f} ms per view  {len(views) / dt:8.1f} views/s", flush=True)
