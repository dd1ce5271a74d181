"""diagnostic: forward + backward of one training layer (5x5 128 -> 128 with batch norm, no activation: g_cB1 of the 4x
generator at tile 8) repeated with the allocator shuffled in between; every output is compared bit for bit with the first
repetition, and a deviation is described (which tensor, how many elements, where)"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
import mpgan_amd  # noqa: F401
from mpgan_amd.train import ConvLayerFn
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n, h, cin, cout, k = 4, 32, 128, 128, 5
g = torch.Generator(device=dev).manual_seed(3)
x0 = torch.randn((n, h, h, cin), device=dev, generator=g).relu_()
w0 = torch.randn((k, k, cin, cout), device=dev, generator=g)
gamma0 = 1 + 0.1 * torch.randn((cout,), device=dev, generator=g)
beta0 = 0.1 * torch.randn((cout,), device=dev, generator=g)
dy0 = torch.randn((n, h, h, cout), device=dev, generator=g) * 1e-4
rng = np.random.default_rng(0)
ref = None
bad = 0
keepalive = []
for it in range(reps):
    # shuffle the allocator: free / allocate blocks of random sizes so that the step's tensors land elsewhere
    if it % 3 == 0:
        keepalive = [torch.empty((int(rng.integers(1, 1 << 20)),), device=dev) for _ in range(int(rng.integers(0, 6)))]
    x = x0.clone().requires_grad_(True)
    w = w0.clone().requires_grad_(True)
    gamma = gamma0.clone().requires_grad_(True)
    beta = beta0.clone().requires_grad_(True)
    cfg = {"stride": (1, 1), "wscale": 0.02, "act": None, "leak": 0.2, "prec": 3, "fc": False, "eps": 1e-3}
    y = ConvLayerFn.apply(x, w, None, gamma, beta, cfg)
    dx, dw, dg, db = torch.autograd.grad(y, (x, w, gamma, beta), dy0)
    out = {"y": y.detach(), "dx": dx, "dw": dw, "dgamma": dg, "dbeta": db}
    if ref is None:
        ref = {kk: v.clone() for kk, v in out.items()}
        continue
    for kk, v in out.items():
        if kk == "dw":      # row ranges are combined with atomics: not bit-stable, compare loosely
            e = float((v - ref[kk]).norm() / ref[kk].norm())
            if e > 1e-5:
                bad += 1
                print("rep %d dw: relative deviation %.3e" % (it, e), flush=True)
        elif not torch.equal(v, ref[kk]):
            d = (v - ref[kk]).abs()
            idx = (d > 0).nonzero()
            e = float((v - ref[kk]).norm() / ref[kk].norm())
            if e > 1e-6:
                bad += 1
                desc = ""
                if v.dim() == 4:
                    desc = "n %d..%d y %d..%d x %d..%d c %d..%d" % tuple(int(f(idx[:, a])) for a in range(4) for f in (torch.min, torch.max))
                print("rep %d %s: %d of %d elements differ, relative %.3e  %s" % (it, kk, idx.shape[0], v.numel(), e, desc), flush=True)
print("repetitions %d, deviating outputs %d" % (reps, bad))
