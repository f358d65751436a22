"""Random sweeps of the neighbouring rows (losses, Adam, 3-NN distances, instance transform, activations) against their
restatements, at sizes the committed tests do not pin -- odd image sizes, tensor sizes off every vector width, tiny and
degenerate inputs.  Prints one line per family; exits non-zero on the first mismatch.
    PYTHONPATH=. python profiles/tools/sweep_neighbours.py [n_cases]"""
import sys
import types

import numpy as np
import torch

from oracle import instance_oracle, loss_oracle
from simple_knn._C import distCUDA2
from vegs_amd import instances, losses, optim, scenes

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


def sweep_losses():
    for case in range(N):
        rng = np.random.default_rng(100 + case)
        H, W = int(rng.integers(1, 90)), int(rng.integers(1, 130))
        x = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
        y = np.clip(x + rng.normal(0, 0.2, x.shape), 0, 1).astype(np.float32)
        q = rng.normal(size=(4, H, W)).astype(np.float32)
        s = rng.uniform(1e-4, 0.4, (3, H, W)).astype(np.float32)
        n = rng.normal(size=(3, H, W)).astype(np.float32)
        lam, lam_n = float(rng.uniform(0, 1)), float(rng.uniform(0, 0.1))
        xt, qt, st = (torch.tensor(a, device=dev, requires_grad=True) for a in (x, q, s))
        cam = types.SimpleNamespace(original_normal=torch.tensor(n, device=dev), R=scenes.R_KITTI)
        loss, aux = losses.training_loss(xt, torch.tensor(y, device=dev), cam, qt, st, lam, lam_n)
        loss.backward()
        l1, ss, gx = loss_oracle.photometric(x, y, 1.0 - lam, -lam)
        ng, dq, ds = loss_oracle.normal_guidance(q, s, n, scenes.R_KITTI, lam_n)
        want = (1 - lam) * l1 + lam * (1 - ss) + lam_n * loss_oracle.normal_guidance(q, s, n, scenes.R_KITTI)[0]
        assert abs(loss.item() - want) < 3e-6 * max(1, abs(want)), (case, H, W, loss.item(), want)
        assert rel(xt.grad.cpu().numpy(), gx) < 1e-4, (case, H, W, "dimage", rel(xt.grad.cpu().numpy(), gx))
        assert rel(qt.grad.cpu().numpy(), dq) < 3e-4 and rel(st.grad.cpu().numpy(), ds) < 1e-4, (case, H, W, "ng grads")
    print(f"losses: {N} random frames (1x1 ... 89x129) ok")


def sweep_adam():
    for case in range(N):
        rng = np.random.default_rng(200 + case)
        shapes = [tuple(int(v) for v in rng.integers(0, 40, int(rng.integers(1, 4)))) for _ in range(int(rng.integers(1, 12)))]
        pc = [torch.nn.Parameter(torch.tensor(rng.normal(size=s).astype(np.float32))) for s in shapes]
        pg = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in pc]
        lrs = [float(10 ** rng.uniform(-5, -1)) for _ in shapes]
        oc = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(pc, lrs)], lr=0.0, eps=1e-15)
        og = optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(pg, lrs)], lr=0.0, eps=1e-15)
        for it in range(4):
            for a, b in zip(pc, pg):
                if rng.random() < 0.15:
                    a.grad = b.grad = None
                    continue
                g = rng.normal(0, 10 ** rng.uniform(-6, 1), tuple(a.shape)).astype(np.float32)
                a.grad, b.grad = torch.tensor(g), torch.tensor(g, device=dev)
            oc.step(); og.step()
        for a, b in zip(pc, pg):
            if a.numel():
                assert rel(b.detach().cpu().numpy(), a.detach().numpy()) < 3e-6, (case, tuple(a.shape))
    print(f"adam: {N} random parameter sets (empty tensors, skipped gradients, 1 ... 11 tensors) ok")


def sweep_knn():
    for case in range(N):
        rng = np.random.default_rng(300 + case)
        n = int(rng.choice([4, 5, 7, 63, 64, 65, 200, 1000, 3000]))
        kind = case % 3
        pts = rng.normal(size=(n, 3)) * (10 ** rng.uniform(-3, 2))
        if kind == 1:
            pts[:, 2] = 0.0                                   # planar
        if kind == 2:
            pts = np.round(pts / pts.std() * 3) * pts.std() / 3    # lattice: many exact ties and duplicates
        pts = pts.astype(np.float32)
        d2 = ((pts[:, None, :].astype(np.float64) - pts[None].astype(np.float64)) ** 2).sum(-1)
        d2[np.arange(n), np.arange(n)] = np.inf
        want = np.sort(d2, 1)[:, :3].mean(1)
        got = distCUDA2(torch.tensor(pts, device=dev)).cpu().numpy()
        assert np.allclose(got, want, rtol=2e-4, atol=1e-12 * float(d2[np.isfinite(d2)].max())), (case, n, kind, np.abs(got - want).max())
    print(f"knn: {N} random clouds (4 ... 3000 points, planar, lattices with ties) ok")


def sweep_instances():
    for case in range(N):
        rng = np.random.default_rng(400 + case)
        k = int(rng.integers(1, 6))
        boxes, Bs = [], []
        for _ in range(k):
            n = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1000]))
            m = rng.normal(size=(n, 3)).astype(np.float32)
            s = np.exp(rng.normal(-3, 1, (n, 3))).astype(np.float32)
            q = rng.normal(size=(n, 4)).astype(np.float32)
            A = rng.normal(size=(3, 3)); Q, _ = np.linalg.qr(A)
            if np.linalg.det(Q) < 0:
                Q[:, 0] = -Q[:, 0]
            B = np.eye(4, dtype=np.float32)
            B[:3, :3] = (Q * np.exp(rng.normal(0, 0.5, 3))[None, :]).astype(np.float32)
            B[:3, 3] = rng.normal(0, 5, 3)
            boxes.append({"means3D": m, "scales": s, "rotations": q, "shs": np.zeros((n, 16, 3), np.float32), "opacities": np.ones((n, 1), np.float32)})
            Bs.append(B)
        tb = [{kk: torch.tensor(v, device=dev, requires_grad=kk in ("means3D", "scales", "rotations")) for kk, v in b.items()} for b in boxes]
        tB = [torch.tensor(B, device=dev, requires_grad=True) for B in Bs]
        kw = instances.prepare_and_merge(None, tb, tB)
        gm, gs, gr = (rng.normal(size=tuple(kw[kk].shape)).astype(np.float32) for kk in ("means3D", "scales", "rotations"))
        torch.autograd.backward([kw["means3D"], kw["scales"], kw["rotations"]], [torch.tensor(g, device=dev) for g in (gm, gs, gr)])
        row = 0
        for b, B, t, tBi in zip(boxes, Bs, tb, tB):
            n = b["means3D"].shape[0]
            m_, s_, q_ = instance_oracle.forward(b["means3D"], b["scales"], b["rotations"], B)
            sl = slice(row, row + n)
            assert rel(kw["means3D"][sl].detach().cpu().numpy(), m_) < 1e-5 and rel(kw["scales"][sl].detach().cpu().numpy(), s_) < 1e-5
            # q and -q are the same rotation; the candidate choice is part of the restated function, so signs must agree
            assert np.abs(kw["rotations"][sl].detach().cpu().numpy() - q_).max() < 2e-5, (case, "rotations")
            dm, dsc, dr, dB = instance_oracle.backward(b["means3D"], b["scales"], b["rotations"], B, gm[sl], gs[sl], gr[sl])
            assert rel(t["means3D"].grad.cpu().numpy(), dm) < 2e-4 and rel(t["scales"].grad.cpu().numpy(), dsc) < 2e-4, (case, "dm ds")
            assert rel(t["rotations"].grad.cpu().numpy(), dr) < 5e-4, (case, "dr", rel(t["rotations"].grad.cpu().numpy(), dr))
            assert rel(tBi.grad.cpu().numpy(), dB) < 5e-4, (case, "dB", rel(tBi.grad.cpu().numpy(), dB))
            row += n
    print(f"instances: {N} random frames (1 ... 5 instances of 1 ... 1000 Gaussians, anisotropic box2world) ok")


if __name__ == "__main__":
    sweep_losses()
    sweep_adam()
    sweep_knn()
    sweep_instances()
