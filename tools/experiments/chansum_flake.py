"""diagnostic: mpg_bn_train_bwd on fixed data repeated; how many distinct dbeta / dgamma / dx results, how far apart"""
import sys
sys.path.insert(0, "/root/repo")
import torch
import mpgan_amd  # noqa: F401
from mpgan_amd import train_ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(5)
n, c = 4 * 32 * 32, 128
x = torch.randn((4, 32, 32, c), device=dev, generator=g)
dy = torch.randn((4, 32, 32, c), device=dev, generator=g) * 1e-4
mean = x.mean(dim=(0, 1, 2)).contiguous()
var = x.var(dim=(0, 1, 2), unbiased=False).contiguous()
gamma = (1 + 0.1 * torch.randn((c,), device=dev, generator=g)).contiguous()
ref = None
seen = {}
worst = [0.0, 0.0, 0.0]
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5000):
    dx, dgam, dbet = train_ops.bn_train_bwd(dy, x, mean, var, gamma)
    if ref is None:
        ref = (dx.clone(), dgam.clone(), dbet.clone())
        exact = (dy.double().sum(dim=(0, 1, 2)), (dy.double() * ((x.double() - mean.double()) / torch.sqrt(var.double() + 1e-3))).sum(dim=(0, 1, 2)))
        print("dbeta vs float64: %.2e   dgamma vs float64: %.2e" % (float((dbet.double() - exact[0]).norm() / exact[0].norm()),
                                                                       float((dgam.double() - exact[1]).norm() / exact[1].norm())))
        continue
    key = (dbet.cpu().numpy().tobytes(), dgam.cpu().numpy().tobytes())
    seen[key] = seen.get(key, 0) + 1
    for i, (a, b) in enumerate(zip((dx, dgam, dbet), ref)):
        worst[i] = max(worst[i], float((a - b).norm() / b.norm()))
print("distinct (dbeta, dgamma) results: %d; counts %s" % (len(seen), sorted(seen.values(), reverse=True)[:8]))
print("largest relative deviation from the first call: dx %.2e dgamma %.2e dbeta %.2e" % tuple(worst))
