"""development probe: time single fused-conv launches of the 4x generator layers with parts disabled"""
import sys, ctypes
import numpy as np, torch
sys.path.insert(0, ".")
import mpgan_amd
from mpgan_amd import ops, _lib

dev = "cuda:0"
N, H = 8, 256
def run(cin, cout, k, segs_extra=None, dbg=0, prec=3, iters=20):
    g8kw = dict(want_g8=True)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, H, H, cin), device=dev, generator=g).relu_()
    w = torch.randn((k, k, cin, cout), device=dev, generator=g)
    pk = ops.pack_conv_weights(w, wscale=0.05, prec=prec)
    segs = [ops.Segment(x, pk)]
    if segs_extra:
        c2 = segs_extra
        x2 = torch.randn((N, H, H, c2), device=dev, generator=g).relu_()
        w2 = torch.randn((1, 1, c2, cout), device=dev, generator=g)
        segs.append(ops.Segment(x2, ops.pack_conv_weights(w2, wscale=0.05, prec=prec)))
    out = torch.empty((N, H, H, cout), device=dev)
    f = lambda: ops.conv2d_fused(segs, (H, H), act="relu", want_f32=False, reserved=dbg, **g8kw)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

layers = [("b0.A 1->2", 1, 2, 5, None), ("b0.B 2->8+s1", 2, 8, 5, 1), ("b1.A 8->128", 8, 128, 5, None),
          ("b1.B 128->128+s8", 128, 128, 5, 8), ("b2.A 128->32", 128, 32, 5, None), ("b2.B 32->8+s128", 32, 8, 5, 128),
          ("b3.A 8->2", 8, 2, 5, None), ("b3.B 2->1+s8", 2, 1, 5, 8)]
print("%-20s %9s %9s %9s" % ("layer", "F16X3", "F16F6", "F16X1"))
for name, cin, cout, k, ex in layers:
    t = [run(cin, cout, k, ex, 0, p) for p in (3, 2, 1)]
    print("%-20s %9.1f %9.1f %9.1f" % ((name,) + tuple(t)), flush=True)
