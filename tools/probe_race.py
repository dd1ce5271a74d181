"""diagnostic: do single launches give the same result when other launches run concurrently on other streams?"""
import sys
import torch
sys.path.insert(0, ".")
import mpgan_amd
from mpgan_amd import ops
dev = "cuda:0"
N, H = 8, 256
def mk(cin, cout, k, extra, seed, prec=2):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn((N, H, H, cin), device=dev, generator=g).relu_()
    w = torch.randn((k, k, cin, cout), device=dev, generator=g)
    segs = [ops.Segment(x, ops.pack_conv_weights(w, wscale=0.05, prec=prec))]
    if extra:
        x2 = torch.randn((N, H, H, extra), device=dev, generator=g).relu_()
        w2 = torch.randn((1, 1, extra, cout), device=dev, generator=g)
        segs.append(ops.Segment(x2, ops.pack_conv_weights(w2, wscale=0.05, prec=prec)))
    return segs
def run(segs):
    return ops.conv2d_fused(segs, (H, H), act="relu", want_f32=True)
which = sys.argv[1] if len(sys.argv) > 1 else "small"
CO = {"b2A": (128, 32, 5, None), "b2B": (32, 8, 5, 128), "b3A": (8, 2, 5, None), "b1A": (8, 128, 5, None), "b1B": (128, 128, 5, 8),
      "b0B": (2, 8, 5, 1), "b3B": (2, 1, 5, 8)}
if which.startswith("pair:"):
    co = CO[which.split(":")[1]]
    LAYX = [(2, 1, 5, 8), co, co, co]
LAY = LAYX if which.startswith("pair:") else {"small": [(8, 2, 5, None), (2, 1, 5, 8), (1, 2, 5, None), (2, 8, 5, 1)],
       "mixed": [(8, 2, 5, None), (128, 128, 5, 8), (2, 8, 5, 1), (128, 32, 5, None)],
       "big": [(128, 128, 5, 8), (128, 32, 5, None), (8, 128, 5, None), (32, 8, 5, 128)],
       "tail": [(2, 1, 5, 8), (128, 32, 5, None), (32, 8, 5, 128), (8, 2, 5, None), (8, 128, 5, None), (128, 128, 5, 8)]}[which]
jobs = [mk(*l, seed=i) for i, l in enumerate(LAY)]
ref = []
for j in jobs:
    ref.append(run(j).clone()); torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in jobs]
bad = 0
for rep in range(20):
    outs = []
    for st, j in zip(streams, jobs):
        with torch.cuda.stream(st):
            outs.append([run(j) for _ in range(3)])
    torch.cuda.synchronize()
    for i, os_ in enumerate(outs):
        for o in os_:
            if not torch.equal(o, ref[i]):
                bad += 1
                d = (o - ref[i]).abs()
                print("rep %d job %d %s: %d elements differ, max %.3e" % (rep, i, LAY[i], int((d > 0).sum()), float(d.max())))
print(which, "mismatching launches:", bad)
