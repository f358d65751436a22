import sys, time, torch
sys.path.insert(0, '.')
from vegs_amd import optim
dev = torch.device('cuda', 0)
P = 2_000_000
g = torch.Generator(device=dev).manual_seed(1)
means = torch.randn(P, 3, device=dev, generator=g) * 10
campos = torch.zeros(1, 3, device=dev)
for name, fac in (("random", torch.randn(1, P, 3, device=dev, generator=g) * 1e-3), ("zero", torch.zeros(1, P, 3, device=dev))):
    dc = torch.nn.Parameter(torch.randn(P, 1, 3, device=dev)); rest = torch.nn.Parameter(torch.randn(P, 15, 3, device=dev))
    opt = optim.Adam([{"params": [dc], "lr": 1e-3, "name": "f_dc"}, {"params": [rest], "lr": 1e-4, "name": "f_rest"}], lr=0.0, eps=1e-15)
    for _ in range(3):
        optim.adam_step_sh_factored(opt, dc, rest, means, campos, fac, 3, 1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        optim.adam_step_sh_factored(opt, dc, rest, means, campos, fac, 3, 1.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(name, round(ms, 4), "ms", round(P * 1152 / ms / 1e9, 2), "TB/s")
    del dc, rest, opt
