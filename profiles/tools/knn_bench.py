import torch, time, numpy as np
from simple_knn._C import distCUDA2
for n in [100_000, 1_000_000, 2_000_000]:
    rng = np.random.default_rng(1)
    pts = np.stack([rng.uniform(0, 400, n), rng.normal(0, 6, n), rng.normal(0, 2, n)], 1).astype(np.float32)
    t = torch.tensor(pts, device="cuda:0")
    distCUDA2(t); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): distCUDA2(t)
    torch.cuda.synchronize()
    print("knn", n, (time.time() - t0) / 3 * 1e3, "ms", flush=True)
