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


# ---- two ranks through the real operator.  RCCL refuses two ranks on one device, so the ranks of this test talk over
# gloo (VEGS_DIST_BACKEND=gloo, vegs_amd/dist.py) while each of them renders on the GPU: the HIP path, the per-rank view
# assignment and the exchange arithmetic are the production ones; only the transport differs from the 8-GPU job.
WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from vegs_amd import dist as vdist, harness, scenes
rank, world, local = vdist.init_from_env()
assert world == 2 and torch.distributed.get_backend() == "gloo"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
sc, deg = scenes.scene_street(P=60000, length=80.0, sh_degree=3, seed=12)
cams = [scenes.kitti_camera(5.0, 0.3, 688, 188), scenes.kitti_camera(5.0, -0.3, 688, 188)]     # a stereo pair
T = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in sc.items()}
params = [T[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
g = np.load(%(gpath)r)
v = vdist.view_for_rank(0, rank, world, len(cams))
assert v == rank
pkg = harness.render(cams[v], T, deg, torch.zeros(3, device=dev))
torch.autograd.backward([pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]],
                        [torch.tensor(g["gc%%d" %% v], device=dev), torch.tensor(g["gq%%d" %% v], device=dev), torch.tensor(g["gs%%d" %% v], device=dev)])
vdist.allreduce_grads(params, world)
gsum, den, mr = vdist.allreduce_densification_stats(pkg["viewspace_points"].grad, pkg["visibility_filter"], pkg["radii"])
dense = {"grad_" + k: T[k].grad.cpu().numpy() for k in T}
# ---- the same iteration with the FACTORED SH exchange: 3 floats per Gaussian and view all-gathered instead of 48 all-reduced
from vegs_amd import optim
from vegs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
for t in T.values():
    t.grad = None
ct = harness.cam_tensors(cams[v], dev)
rs = GaussianRasterizationSettings(188, 688, cams[v].tanfovx, cams[v].tanfovy, torch.zeros(3, device=dev), 1.0,
                                   ct["viewmatrix"], ct["projmatrix"], deg, ct["campos"], False, False)
sink = torch.zeros(60000, 3, device=dev, requires_grad=True)
out = GaussianRasterizer(rs)(means3D=T["means3D"], means2D=torch.zeros(60000, 3, device=dev, requires_grad=True),
                             opacities=T["opacities"], shs=T["shs"], scales=T["scales"], rotations=T["rotations"],
                             sh_color_grad=sink)
torch.autograd.backward([out[0], out[2], out[3]],
                        [torch.tensor(g["gc%%d" %% v], device=dev), torch.tensor(g["gq%%d" %% v], device=dev), torch.tensor(g["gs%%d" %% v], device=dev)])
assert T["shs"].grad is None
F, C = vdist.exchange_factored([T[k] for k in ("means3D", "opacities", "scales", "rotations")], sink.grad, ct["campos"], world)
assert F.shape == (2, 60000, 3) and torch.equal(F[rank], sink.grad)
fact_shs = optim.sh_grad_from_factors(T["means3D"].detach(), C, F, deg, 16, 1.0 / world)
fact_means3D = T["means3D"].grad.cpu().numpy()
# ---- and once more with the OVERLAPPED exchange: the backward in two ABI calls (vr_backward_render ->
# hook: the factors start travelling -> vr_backward_preprocess), the other gradients row-sparse
for t in T.values():
    t.grad = None
sink2 = torch.zeros(60000, 3, device=dev, requires_grad=True)
out = GaussianRasterizer(rs)(means3D=T["means3D"], means2D=torch.zeros(60000, 3, device=dev, requires_grad=True),
                             opacities=T["opacities"], shs=T["shs"], scales=T["scales"], rotations=T["rotations"],
                             sh_color_grad=sink2)
ex = vdist.FactorExchange(world, sparse_rows=True)
ex.begin(ct["campos"])
torch.autograd.backward([out[0], out[2], out[3]],
                        [torch.tensor(g["gc%%d" %% v], device=dev), torch.tensor(g["gq%%d" %% v], device=dev), torch.tensor(g["gs%%d" %% v], device=dev)])
assert ex._f is not None and ex._f.data_ptr() == sink2.grad.data_ptr()      # the hook ran between the two halves
F2, C2 = ex.finish([T[k] for k in ("means3D", "opacities", "scales", "rotations")], out[5])
from vegs_amd import rasterizer
assert rasterizer._split_hook is None
assert torch.equal(F2[rank], sink2.grad) and torch.equal(C2, C)
ovl_shs = optim.sh_grad_from_factors(T["means3D"].detach(), C2, F2, deg, 16, 1.0 / world)
np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), gsum=gsum.cpu().numpy(), den=den.cpu().numpy(), mr=mr.cpu().numpy(),
         fact_shs=fact_shs.cpu().numpy(), fact_means3D=fact_means3D, ovl_shs=ovl_shs.cpu().numpy(),
         ovl_means3D=T["means3D"].grad.cpu().numpy(), ovl_rows=np.int64(ex.rows_exchanged), **dense)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print("RANK_OK", rank)
"""


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_equal_mean_of_two_views(tmp_path):
    """2 ranks x 1 view each, all-reduced  ==  1 process averaging the two views' gradients (loss = mean over the
    views), through the real HIP operator; plus the densification statistics of the view batch."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import assert_grad_close
    from vegs_amd import harness, scenes
    H, W = 188, 688
    rng = np.random.default_rng(4)
    gd = {}
    for v in range(2):
        gd[f"gc{v}"] = rng.normal(size=(3, H, W)).astype(np.float32)
        gd[f"gq{v}"] = rng.normal(size=(4, H, W)).astype(np.float32)
        gd[f"gs{v}"] = rng.normal(size=(3, H, W)).astype(np.float32)
    gpath = str(tmp_path / "gouts.npz")
    np.savez(gpath, **gd)
    script = WORKER % dict(root=ROOT, gpath=gpath, out=str(tmp_path))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VEGS_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs) and all("RANK_OK" in o for o in outs), "\n".join(outs)

    # ---- one process, both views, mean of the gradients
    dev = torch.device("cuda:0")
    sc, deg = scenes.scene_street(P=60000, length=80.0, sh_degree=3, seed=12)
    cams = [scenes.kitti_camera(5.0, 0.3, W, H), scenes.kitti_camera(5.0, -0.3, W, H)]
    T = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in sc.items()}
    gsum = torch.zeros(60000, 1, device=dev)
    den = torch.zeros(60000, 1, device=dev)
    mr = torch.zeros(60000, dtype=torch.int32, device=dev)
    for v, cam in enumerate(cams):
        pkg = harness.render(cam, T, deg, torch.zeros(3, device=dev))
        torch.autograd.backward([pkg["render"], pkg["render_cov_quat"], pkg["render_cov_scale"]],
                                [torch.tensor(gd[f"gc{v}"], device=dev), torch.tensor(gd[f"gq{v}"], device=dev),
                                 torch.tensor(gd[f"gs{v}"], device=dev)])
        vis = pkg["visibility_filter"]
        gsum += torch.norm(pkg["viewspace_points"].grad[:, :2], dim=-1, keepdim=True) * vis[:, None]
        den += vis[:, None].float()
        mr = torch.maximum(mr, pkg["radii"])
    want = {k: (T[k].grad / 2).cpu().numpy() for k in T}
    R = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    for k in T:
        assert np.array_equal(R[0]["grad_" + k], R[1]["grad_" + k]), k          # identical on every rank
        assert_grad_close("2-rank " + k, R[0]["grad_" + k], want[k], rtol=1e-3, floor=2e-6)
    # factored SH exchange: rebuilt from the all-gathered 3-float factors == the dense all-reduced gradient
    assert np.array_equal(R[0]["fact_shs"], R[1]["fact_shs"])
    assert_grad_close("2-rank factored shs", R[0]["fact_shs"], want["shs"], rtol=1e-3, floor=2e-6)
    assert_grad_close("2-rank factored vs dense exchange", R[0]["fact_shs"], R[0]["grad_shs"], rtol=1e-3, floor=2e-6)
    assert_grad_close("2-rank factored means3D", R[0]["fact_means3D"], want["means3D"], rtol=1e-3, floor=2e-6)
    # overlapped exchange (split backward + row-sparse all-reduce): the same tensors once more
    assert np.array_equal(R[0]["ovl_shs"], R[1]["ovl_shs"]) and np.array_equal(R[0]["ovl_means3D"], R[1]["ovl_means3D"])
    assert_grad_close("2-rank overlapped shs", R[0]["ovl_shs"], want["shs"], rtol=1e-3, floor=2e-6)
    assert_grad_close("2-rank overlapped means3D", R[0]["ovl_means3D"], want["means3D"], rtol=1e-3, floor=2e-6)
    assert 0 < int(R[0]["ovl_rows"]) <= 60000
    assert np.array_equal(R[0]["den"], den.cpu().numpy()) and np.array_equal(R[0]["mr"], mr.cpu().numpy())
    assert_grad_close("2-rank grad-norm sum", R[0]["gsum"], gsum.cpu().numpy(), rtol=1e-3, floor=2e-6)
    assert (R[0]["den"] == 2).sum() > 1000                                         # the stereo views overlap


MULTI_VIEW_WORKER = r"""
import os, sys, types, numpy as np, torch
sys.path.insert(0, %(root)r)
import bench
from vegs_amd import dist as vdist
rank, world, local = vdist.init_from_env()
assert world == 2 and torch.distributed.get_backend() == "gloo"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
args = types.SimpleNamespace(workload="c2", gaussians=120000, width=688, height=188, disc_scale=1.0)
sc, deg, cams, P = bench.build_workload(args)
wl = bench.prepare(sc, deg, cams, dev, np.random.default_rng(1234))
# deterministic backward: the comparison below is about the EXCHANGE; with fp32 atomics two runs of the same view already
# differ by more (edge-on discs: 1e-3 of a row) than the two schemes may
from vegs_amd import rasterizer
rasterizer.set_flags(rasterizer.get_flags() | rasterizer.FLAG_DETERMINISTIC)
out = {}
for scheme in ("dense", "factored"):
    for p in wl["params"]:
        p.grad = None
    step = bench.make_step(wl, rank, world, 2, exchange=scheme, keep_grads=True)        # TWO views per rank and step
    views = step(3)
    assert len(views) == 2
    torch.cuda.synchronize()
    for k, p in zip(("means3D", "shs", "opacities", "scales", "rotations"), wl["params"]):
        out[scheme + "_" + k] = p.grad.cpu().numpy()
np.savez(os.path.join(%(out)r, "mv%%d.npz" %% rank), **out)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print("RANK_OK", rank)
"""


def test_bench_multi_view_factored_exchange_equals_the_dense_one(tmp_path):
    """bench.py --gpus 2 --views-per-step 2 (round-5 verdict item 7a): with the factored exchange every view's SH gradient
    stays its 3-float factor, the other 11 floats accumulate IN PLACE over the rank's two views, and the step ends with one
    all-gather of 2 x 2 factors and one all-reduce of 11 floats -- the step's final gradients must be the dense scheme's
    (all-reduce of 59 floats of autograd's sums), on both ranks, per row."""
    import numpy as np
    from helpers import assert_grad_close
    env = dict(os.environ, VEGS_DIST_BACKEND="gloo")
    script = MULTI_VIEW_WORKER % dict(root=ROOT, out=str(tmp_path))
    sp = tmp_path / "worker_mv.py"
    sp.write_text(script)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(sp)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.count("RANK_OK") == 2, r.stdout[-3000:] + r.stderr[-3000:]
    R = [np.load(tmp_path / f"mv{q}.npz") for q in range(2)]
    for k in ("means3D", "shs", "opacities", "scales", "rotations"):
        for q in range(2):
            assert np.abs(R[q]["dense_" + k]).max() > 0
            assert_grad_close(f"rank {q} {k}", R[q]["factored_" + k], R[q]["dense_" + k], rtol=2e-5, floor=2e-7,
                              outliers=0.0, near=0.0)
        # every rank ends the step with the same tensors (fp32 atomics inside a view differ from run to run: not bit-equal)
        assert_grad_close(f"ranks agree {k}", R[0]["factored_" + k], R[1]["factored_" + k], rtol=2e-5, floor=2e-7, outliers=0.0, near=0.0)


def test_bench_two_ranks_gloo_transport(tmp_path):
    """bench.py --gpus 2 under torch.distributed.run (the driver's launch line), both ranks on the one GPU of this
    box with the gloo transport: the N > 1 code path of the benchmark runs end to end and reports n_gpus 2."""
    import json
    env = dict(os.environ, VEGS_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--gaussians", "200000", "--no-cpu-baseline", "--exchange", "factored"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["config"]["gaussians"] == 200000 and "roofline" in res
    ex = res["exchange"]
    assert ex["scheme"].startswith("factored") and ex["exchange_bytes_per_rank"] == 12 * 200000 + 44 * 200000
    assert ex["ms_per_step_without_exchange"] > 0 and "exchange_exposed_ms" in ex


@pytest.mark.parametrize("exchange", ["factored", "direct"])
def test_bench_eight_ranks_full_size_dry_run(exchange):
    """BASELINE C4 at FULL size -- 2 M Gaussians, 8 views per iteration, one per rank -- as the driver launches it
    (torch.distributed.run, bench.py --gpus 8), with the eight ranks sharing the one GPU of this box (8 x ~2.2 GB): the
    production view assignment, split backward, overlapped exchange and local SH rebuild at the headline size.  Transport:
    gloo staged through the host ("factored": what stands in for RCCL here, which refuses several ranks per device) and
    the direct hipIpc exchange (the ranks map each other's windows exactly as eight GPUs would).  What a single-GPU box
    cannot show is RCCL itself at N = 8 and the xGMI links."""
    import json
    env = dict(os.environ, VEGS_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2",
           "--warmup", "1", "--repeats", "1", "--no-cpu-baseline", "--no-variants", "--exchange", exchange]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 8 and res["config"]["gaussians"] == 2_000_000 and res["value"] > 0
    assert res["exchange"]["scheme"].startswith(exchange) and res["exchange"]["exchange_bytes_per_rank"] == 7 * 12 * 2_000_000 + int(14 / 8 * 44 * 2_000_000)
    assert res["mfragments_per_s"] > 0 and res["roofline"]["launches_timed"] > 0


def _bench_auto(extra_env=None, extra_args=()):
    import json
    env = dict(os.environ, VEGS_DIST_BACKEND="gloo", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--repeats", "1", "--gaussians", "200000", "--no-cpu-baseline", *extra_args]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line, whatever the children did
    return json.loads(lines[0])


def test_bench_exchange_auto_times_rccl_first_and_the_direct_exchange_in_children():
    """bench.py --gpus 2 with the DEFAULT exchange (auto): the headline is timed by the parents with the RCCL-shaped
    factored exchange (gloo here: two ranks share this box's one GPU) and KEPT; the direct hipIpc exchange is set up,
    verified and timed by sacrificial child processes afterwards, and the line reports both."""
    res = _bench_auto()
    ex = res["exchange"]
    assert ex["scheme"].startswith("factored") and res["n_gpus"] == 2 and res["value"] > 0
    auto = ex["auto"]
    assert auto["factored_ms_per_step"] == res["ms_per_step"]
    d = auto["direct"]
    assert d["ok"] is True and d["child_rc"] == 0 and d["line"]["value"] > 0, d
    assert d["line"]["exchange"]["scheme"].startswith("direct") and auto["direct_ms_per_step"] == d["line"]["ms_per_step"]
    assert auto["faster"] in ("direct", "factored") and "exchange_exposed_ms" in ex
    com = ex["communicator"]
    assert com["ranks_seen"] == 2 and [r["rank"] for r in com["ranks"]] == [0, 1] and all(r["device"] == 0 for r in com["ranks"])


@pytest.mark.parametrize("fault", ["segv", "hang"])
def test_bench_line_survives_a_direct_exchange_that_faults_or_hangs(fault):
    """The first contact with peer-to-peer writes across devices happens on the driver's scaling run.  Whatever it does
    there must not cost the record: a child that SEGFAULTS after mapping the peers' windows, or HANGS there (killed by its
    parent at the wall-clock limit, together with the sibling that waits for it), leaves a valid bench line with the RCCL
    number as the headline and `direct.ok` false."""
    args = ("--probe-timeout", "45") if fault == "hang" else ()
    res = _bench_auto({"VEGS_XGMI_PROBE_FAULT": fault}, args)
    ex = res["exchange"]
    assert res["value"] > 0 and ex["scheme"].startswith("factored") and ex["auto"]["factored_ms_per_step"] == res["ms_per_step"]
    d = ex["auto"]["direct"]
    assert d["ok"] is False and "line" not in d and "direct_ms_per_step" not in ex["auto"]
    if fault == "hang":
        assert d.get("timed_out") is True or d["child_rc"] != 0
        assert d["elapsed_s"] < 120


def test_bench_line_contract_single_gpu():
    """bench.py's one JSON line carries everything the driver and the judge read (N = 1, small workload)."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--gaussians", "150000",
           "--no-variants"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "mfragments_per_s",
              "blended_mfragments_per_s"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "views/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["hints"] == "off" and d["repeats"] == 5 and len(d["ms_per_step_regions"]) == 5
    assert min(d["ms_per_step_regions"]) <= d["ms_per_step"] <= max(d["ms_per_step_regions"])
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and 0 < rf["frac"] < 1 and "traffic" in rf
    # provenance of the counter figures, and the secondary ceilings (VALU issue, L2 atomics of the backward)
    assert "traffic_collected" in rf and (rf["traffic"] is None) == (isinstance(rf["traffic_collected"], str))
    sec = rf["secondary"]
    assert abs(sec["l2_atomics"]["atomics_per_view"] - 17 * sec["l2_atomics"]["flushes_per_view"]) <= 17 and sec["l2_atomics"]["flushes_per_view"] > 0
    # fragments, list entries and the roofline fraction are priced on the lists the kernels WALK (the build's own, tighter
    # lists); the same views on the REFERENCE's full tile rectangles (BASELINE.md's definition) are reported beside them
    mc = d["config"]["mean_counters"]
    assert mc["R_lists"] < mc["R"] and mc["F_lists"] < mc["F"] and d["mfragments_per_s"] < d["mfragments_per_s_reference_lists"]
    N = d["config"]["width"] * d["config"]["height"]
    assert abs(rf["alg_bytes_per_launch"] - (72 * mc["R_lists"] + 56 * N + 68 * mc["V"])) <= 72    # (the counters are rounded to 0.1)
    assert rf["frac"] < rf["frac_on_reference_lists"] < 1 and rf["whole_view_frac"] < rf["whole_view_frac_on_reference_lists"] < 1
    # every stage's own fraction of the HBM peak (SURVEY 8a bytes on the build's lists / its HIP-event time)
    assert set(("preprocess", "binning", "render_fwd", "render_bwd", "preprocess_bwd")) <= set(rf["stages"])
    for st in rf["stages"].values():
        assert st["ms"] > 0 and abs(st["frac"] - st["alg_bytes"] / (st["ms"] * 1e-3) / 8e12) < 2e-4
    if rf["traffic"] is not None:
        assert 0 < sec["valu"]["frac"] < 1 and sec["valu"]["peak_ginst_per_s"] == 1228.8
    assert set(("preprocess", "render_fwd", "render_bwd", "preprocess_bwd", "k_seg_bwd")) <= set(rf["stage_ms"])
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb and cb["unit"] == "views/s"
    assert d["blended_mfragments_per_s"] < d["mfragments_per_s"]


def test_bench_c5_workload_line():
    """bench.py --workload c5 (BASELINE C5's full training step: static model + box instances with optimised instance
    models and BoxModels) prints its one JSON line (small sizes; N = 1)."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5", "--gaussians", "120000", "--boxes", "2",
           "--steps", "3", "--warmup", "2", "--repeats", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "views/s" and d["value"] > 0
    assert d["config"]["workload"].startswith("c5") and d["config"]["boxes"] == 2
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and rf["launches_timed"] > 0
