"""CPU: the view-sharded exchange step (vegs_amd/dist.py) on the gloo backend, world_size 2 and 4."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from vegs_amd import dist as vdist
    r, w, _ = vdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    P = 1000
    g = torch.Generator().manual_seed(100 + rank)
    shapes = [(P, 3), (P, 16, 3), (P, 1), (P, 3), (P, 4)]     # the 59 floats per Gaussian
    params = [torch.zeros(s, requires_grad=True) for s in shapes]
    local = [torch.randn(s, generator=g) for s in shapes]
    for p, l in zip(params, local):
        p.grad = l.clone()
    # small bucket threshold so both the in-place and the flat-bucket paths are exercised
    vdist.allreduce_grads(params, world, flat_bucket_bytes=20000)
    vg = torch.randn(P, 3, generator=g)
    vis = torch.rand(P, generator=g) > 0.5
    radii = torch.randint(0, 50, (P,), generator=g, dtype=torch.int32)
    gsum, den, mr = vdist.allreduce_densification_stats(vg, vis, radii)
    # factored SH exchange: 11 floats all-reduced, the 3-float factor and the camera centre all-gathered
    others = [torch.zeros(s, requires_grad=True) for s in [(P, 3), (P, 1), (P, 3), (P, 4)]]
    olocal = [torch.randn(p.shape, generator=g) for p in others]
    for p, l in zip(others, olocal):
        p.grad = l.clone()
    factor = torch.randn(P, 3, generator=g)
    campos = torch.randn(3, generator=g)
    F, Cc = vdist.exchange_factored(others, factor, campos, world)
    torch.save(dict(grads=[p.grad for p in params], local=local, gsum=gsum, den=den, mr=mr, vg=vg, vis=vis,
                    radii=radii, view=[vdist.view_for_rank(s, rank, world, 16) for s in range(4)],
                    others=[p.grad for p in others], olocal=olocal, factor=factor, campos=campos, F=F, C=Cc),
               os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_view_sharded_gradient_exchange_gloo(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    R = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    for k in range(5):
        want = sum(R[q]["local"][k] for q in range(world)) / world          # loss = mean over the views
        for r in range(world):
            assert torch.allclose(R[r]["grads"][k], want, atol=1e-6)
    want_g = sum(torch.norm(R[r]["vg"][:, :2], dim=-1, keepdim=True) * R[r]["vis"][:, None].float() for r in range(world))
    want_d = sum(R[r]["vis"][:, None].float() for r in range(world))
    want_m = R[0]["radii"]
    for q in range(1, world):
        want_m = torch.maximum(want_m, R[q]["radii"])
    for r in range(world):
        assert torch.allclose(R[r]["gsum"], want_g, atol=1e-6)
        assert torch.equal(R[r]["den"], want_d)
        assert torch.equal(R[r]["mr"], want_m)
    # factored exchange: every rank ends up with all factors / camera centres in rank order, the rest averaged
    for r in range(world):
        assert R[r]["F"].shape == (world, 1000, 3) and R[r]["C"].shape == (world, 3)
        for q in range(world):
            assert torch.equal(R[r]["F"][q], R[q]["factor"]) and torch.equal(R[r]["C"][q], R[q]["campos"])
        for k in range(4):
            assert torch.allclose(R[r]["others"][k], sum(R[q]["olocal"][k] for q in range(world)) / world, atol=1e-6)
    # consecutive views go to consecutive ranks, every view visited once per cycle
    for r in range(world):
        assert R[r]["view"] == [(s * world + r) % 16 for s in range(4)]


def test_single_process_is_a_no_op():
    from vegs_amd import dist as vdist
    p = torch.zeros(4, 3, requires_grad=True)
    p.grad = torch.ones(4, 3)
    vdist.allreduce_grads([p], world=1)
    assert torch.equal(p.grad, torch.ones(4, 3))
    g, d, m = vdist.allreduce_densification_stats(torch.ones(4, 3), torch.tensor([True, False, True, True]),
                                                  torch.tensor([1, 2, 3, 4], dtype=torch.int32))
    assert torch.allclose(g[:, 0], torch.tensor([2 ** 0.5, 0, 2 ** 0.5, 2 ** 0.5]))
    assert torch.equal(d[:, 0], torch.tensor([1.0, 0, 1, 1]))


def _worker_overlapped(rank, world, port, outdir):
    """FactorExchange (the overlapped scheme's arithmetic: on gloo nothing overlaps, the calls and their order are the
    production ones) with the row-sparse all-reduce: every rank sees only a window of the Gaussians."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from vegs_amd import dist as vdist
    vdist.init_from_env(backend="gloo")
    P = 4000
    g = torch.Generator().manual_seed(500 + rank)
    # camera `rank` sees Gaussians [300 rank, 300 rank + 700): the union over 8 ranks is 70 % of the rows for world 8
    radii = torch.zeros(P, dtype=torch.int32)
    radii[300 * rank:300 * rank + 700] = torch.randint(1, 40, (700,), generator=g, dtype=torch.int32)
    vis = (radii > 0)[:, None]
    others = [torch.zeros(s, requires_grad=True) for s in [(P, 3), (P, 1), (P, 3), (P, 4)]]
    olocal = [torch.randn(p.shape, generator=g) * vis.reshape((P,) + (1,) * (p.dim() - 1)).float() for p in others]
    for p, l in zip(others, olocal):
        p.grad = l.clone()
    factor = torch.randn(P, 3, generator=g) * vis.float()
    campos = torch.randn(3, generator=g)
    ex = vdist.FactorExchange(world, sparse_rows=True)
    ex.begin(campos)
    ex._on_factors(factor)                    # what the op's backward does between its two halves
    F, Cc = ex.finish(others, radii)
    torch.save(dict(others=[p.grad for p in others], olocal=olocal, factor=factor, campos=campos, F=F, C=Cc,
                    rows=ex.rows_exchanged, bytes=vdist.exchange_bytes_per_rank(P, world, "factored", ex.rows_exchanged)),
               os.path.join(outdir, f"o{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_overlapped_factor_exchange_with_sparse_rows_gloo(tmp_path, world):
    """World 8 = BASELINE config C4's rank count: the factors of all 8 views reach every rank, the other gradients are
    averaged, and only the rows visible on some rank travel (2800 of 4000 at world 8, 1000 at world 2)."""
    mp.spawn(_worker_overlapped, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    R = [torch.load(tmp_path / f"o{r}.pt") for r in range(world)]
    union = 300 * (world - 1) + 700
    for r in range(world):
        assert R[r]["rows"] == union                      # sparse path taken, exactly the union of the visible rows
        for k in range(4):
            want = sum(R[q]["olocal"][k] for q in range(world)) / world
            assert torch.allclose(R[r]["others"][k], want, atol=1e-6), (r, k)
        assert R[r]["F"].shape == (world, 4000, 3) and R[r]["C"].shape == (world, 3)
        for q in range(world):
            assert torch.equal(R[r]["F"][q], R[q]["factor"]) and torch.equal(R[r]["C"][q], R[q]["campos"])
        assert R[r]["bytes"] == int((world - 1) * 12 * 4000 + 2.0 * (world - 1) / world * 44 * union)


def _worker_views(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from vegs_amd import dist as vdist
    vdist.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(7 + rank)
    # two local views per rank: [n_local, rows, 3] factors and [n_local, 3] camera centres, gathered rank-major
    f, c = torch.randn(2, 301, 3, generator=g), torch.randn(2, 3, generator=g)
    F, Cc = vdist.all_gather_views(f, c, world)
    # everything small in ONE flat bucket (instance models + BoxModel deltas of the trainer), copied back by one foreach
    ps = [torch.zeros(s, requires_grad=True) for s in [(40, 3), (40, 1, 3), (4,), (3,), (3,), (0, 3), (17, 4)]]
    local = [torch.randn(p.shape, generator=g) for p in ps]
    for p, l in zip(ps, local):
        p.grad = l.clone()
    ps[5].grad = None                                   # a parameter without a gradient is skipped (same on every rank)
    vdist.allreduce_grads(ps, world, flat_bucket_bytes=1 << 40)
    # the overlapped exchange's hook is taken out again when the backward raises
    from vegs_amd import rasterizer
    ex = vdist.FactorExchange(world)
    try:
        with ex.armed(torch.zeros(3)):
            assert rasterizer._split_hook is not None
            with pytest.raises(RuntimeError):
                ex.begin(torch.zeros(3))                # nested begin refused
            raise KeyError("backward failed")
    except KeyError:
        pass
    assert rasterizer._split_hook is None and not ex._armed
    torch.save(dict(f=f, c=c, F=F, C=Cc, grads=[p.grad for p in ps], local=local), os.path.join(outdir, f"v{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_views_and_flat_bucket_gloo(tmp_path):
    """The pieces the multi-rank trainer adds to the exchange (vegs_amd/iteration.py: Trainer.step_views): rank-major
    all-gather of several local views, one flat bucket for many small gradients, the hook hygiene of FactorExchange."""
    world = 2
    mp.spawn(_worker_views, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    R = [torch.load(tmp_path / f"v{r}.pt") for r in range(world)]
    for r in range(world):
        assert R[r]["F"].shape == (4, 301, 3) and R[r]["C"].shape == (4, 3)
        for q in range(world):
            assert torch.equal(R[r]["F"][2 * q:2 * q + 2], R[q]["f"]) and torch.equal(R[r]["C"][2 * q:2 * q + 2], R[q]["c"])
    for k in range(7):
        if k == 5:
            assert R[0]["grads"][k] is None
            continue
        want = (R[0]["local"][k] + R[1]["local"][k]) * 0.5
        assert torch.equal(R[0]["grads"][k], R[1]["grads"][k]) and torch.allclose(R[0]["grads"][k], want, rtol=0, atol=1e-7)


def _worker_multiview(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from oracle import torch_ref
    from vegs_amd import dist as vdist
    vdist.init_from_env(backend="gloo")
    P, k, deg = 500, 2, 3
    g = torch.Generator().manual_seed(1)                       # the model: the same on every rank
    means = torch.randn(P, 3, generator=g, dtype=torch.float64)
    sh = torch.randn(P, 16, 3, generator=g, dtype=torch.float64, requires_grad=True)
    g = torch.Generator().manual_seed(50 + rank)               # this rank's k views
    campos = (torch.randn(k, 3, generator=g) * 3).double()      # (fp32 values: the gather carries camera centres as fp32)
    factors = torch.randn(k, P, 3, generator=g, dtype=torch.float64)
    others = [torch.zeros(s, dtype=torch.float64, requires_grad=True) for s in [(P, 3), (P, 1), (P, 3), (P, 4)]]
    per_view = [[torch.randn(p.shape, generator=g, dtype=torch.float64) for p in others] for _ in range(k)]

    def dense_sh(c, f):      # dL/dsh of one view = basis(dir(c, mean)) x factor (oracle/torch_ref.sh_to_rgb is linear in sh)
        d = means - c[None]
        d = d / d.norm(dim=1, keepdim=True)
        return torch.autograd.grad(torch_ref.sh_to_rgb(deg, sh, d), sh, grad_outputs=f)[0]
    # dense scheme: autograd's sum over the local views, then the mean over the ranks of all 59 floats
    sh_dense = sum(dense_sh(campos[v], factors[v]) for v in range(k))
    dparams = [torch.zeros_like(sh_dense, requires_grad=True)] + [torch.zeros_like(p, requires_grad=True) for p in others]
    dparams[0].grad = sh_dense.clone()
    for j, p in enumerate(dparams[1:]):
        p.grad = sum(per_view[v][j] for v in range(k))
    vdist.allreduce_grads(dparams, world, flat_bucket_bytes=1 << 40)
    # factored scheme with several views per rank (bench.py make_step: fact_multi; iteration.Trainer.step_views): the factors
    # stay factors, the 11 other floats are accumulated locally (in place) and reduced once
    for j, p in enumerate(others):
        p.grad = per_view[0][j].clone()
        for v in range(1, k):
            p.grad.add_(per_view[v][j])
    F, Cc = vdist.all_gather_views(factors, campos, world)
    vdist.allreduce_grads(others, world, flat_bucket_bytes=1 << 40)
    sh_fact = sum(dense_sh(Cc[v].double(), F[v]) for v in range(world * k)) / world
    torch.save(dict(sh_dense=dparams[0].grad, sh_fact=sh_fact, others_dense=[p.grad for p in dparams[1:]],
                    others_fact=[p.grad for p in others], F=F, C=Cc, model=vdist.exchange_model(2_000_000, world, "factored", k)),
               os.path.join(outdir, f"m{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_multi_view_factored_exchange_equals_dense_gloo(tmp_path):
    """Several views per rank and step at N > 1 (bench.py --views-per-step k, round 6): all-gather of the k factors per rank +
    all-reduce of the locally accumulated 11 floats  ==  all-reduce of the 59 dense floats of autograd's per-rank sums.  The
    SH rebuild is done here with the float64 restatement's basis (the HIP kernel's is pinned against it on the GPU)."""
    world = 2
    mp.spawn(_worker_multiview, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    R = [torch.load(tmp_path / f"m{r}.pt") for r in range(world)]
    for r in range(world):
        assert R[r]["F"].shape == (4, 500, 3) and R[r]["C"].shape == (4, 3)
        assert torch.allclose(R[r]["sh_fact"], R[r]["sh_dense"], rtol=0, atol=1e-12)
        for a, b in zip(R[r]["others_fact"], R[r]["others_dense"]):
            assert torch.allclose(a, b, rtol=0, atol=1e-12)
    assert torch.equal(R[0]["sh_fact"], R[1]["sh_fact"])


def test_exchange_model_matches_design_section_8():
    """vegs_amd.dist.exchange_model: the PREDICTION bench.py prints beside the measured exchange time."""
    from vegs_amd import dist as vdist
    assert vdist.exchange_model(2_000_000, 1, "factored") is None
    m = vdist.exchange_model(2_000_000, 8, "factored")
    assert m["links_per_gpu"] == 7 and m["all_reduce"]["bytes"] == 88_000_000 and m["all_gather"]["block_bytes"] == 24_000_000
    assert m["all_reduce"]["bytes_per_link_direct"] == 22_000_000                 # 2 x 11 MB per link (DESIGN: "2 x 72 us")
    assert abs(m["all_reduce"]["ms_direct"] - 0.1438) < 1e-3 and abs(m["all_gather"]["ms_direct"] - 0.1569) < 1e-3
    assert abs(m["all_reduce"]["ms_ring"] - 2 * 7 / 8 * 88e6 / 300e9 * 1e3) < 1e-3
    d = vdist.exchange_model(2_000_000, 8, "dense")
    assert d["all_gather"] is None and d["all_reduce"]["bytes"] == 472_000_000 and abs(d["all_reduce"]["ms_ring"] - 2.7533) < 1e-3
    m2 = vdist.exchange_model(2_000_000, 8, "factored", views_per_rank=2)
    assert m2["all_gather"]["block_bytes"] == 48_000_000 and m2["all_reduce"] == m["all_reduce"]
