"""diagnostic: one generator call (8 slices of 256^2) repeated on k streams concurrently vs its serial result"""
import sys
import torch
sys.path.insert(0, ".")
import mpgan_amd
from mpgan_amd import multipass as MP
dev = "cuda:0"
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
side = 256 if mode == 1 else 64
cfg = dict(tile_low=64, up_res=4, channels=1, upsampling_mode=mode, batch_norm=True)
g = MP.Generator("gen_resnet", cfg, None, 2, device=dev, seed=778)
gens = [g] + [g.clone() for _ in range(k - 1)]
xs = [torch.rand((8, side, side, 1), device=dev, generator=torch.Generator(device=dev).manual_seed(i)) for i in range(k)]
ref = []
for gg, x in zip(gens, xs):
    ref.append(gg(x).clone()); torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(k)]
bad = 0
for rep in range(10):
    outs = []
    for it in range(4):
        for st, gg, x in zip(streams, gens, xs):
            with torch.cuda.stream(st):
                outs.append(gg(x))
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        r = ref[i % k]
        if not torch.equal(o, r):
            bad += 1
            d = (o - r).abs()
            idx = (d > 0).nonzero()
            print("rep %d call %d lane %d: %d px differ max %.3e; n %d..%d y %d..%d x %d..%d" % (
                rep, i, i % k, idx.shape[0], float(d.max()), int(idx[:, 0].min()), int(idx[:, 0].max()),
                int(idx[:, 1].min()), int(idx[:, 1].max()), int(idx[:, 2].min()), int(idx[:, 2].max())))
print("lanes", k, "mode", mode, "mismatching calls:", bad, "of", 40 * k)
import ctypes, os
if os.environ.get("MPGAN_LIB_OVERRIDE", "").endswith("diag.so"):
    L = ctypes.CDLL(os.environ["MPGAN_LIB_OVERRIDE"])
    buf = (ctypes.c_uint * 2)()
    print("rc", L.mpg_debug_small_diag(buf), "small-kernel LDS check: bad weights %d, bad tile words %d" % (buf[0], buf[1]))
