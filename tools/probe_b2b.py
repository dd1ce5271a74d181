#!/usr/bin/env python3
"""timing probe: resBlock 2 convB (5x5 32->8) and its 1x1 128->8 shortcut, alone and fused, 8 slices of 256^2"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib
import torch
sys.modules.setdefault("mpgan_amd", importlib.import_module("multi-pass-gan_amd"))
from mpgan_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n, h = 8, 256
a = torch.randn((n, h, h, 32), device=dev, generator=g).relu_()
x = torch.randn((n, h, h, 128), device=dev, generator=g).relu_()
wb = torch.randn((5, 5, 32, 8), device=dev, generator=g) * 0.03
ws = torch.randn((1, 1, 128, 8), device=dev, generator=g) * 0.08
ga, gx = ops.to_g8(a), ops.to_g8(x)
pb, ps = ops.pack_conv_weights(wb, prec=2), ops.pack_conv_weights(ws, prec=2)


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


print("5x5 32->8 alone      %.1f us" % timeit(lambda: ops.conv2d_fused([ops.Segment(ga, pb)], (h, h), act="relu", want_f32=False, want_g8=True)))
print("1x1 128->8 alone     %.1f us" % timeit(lambda: ops.conv2d_fused([ops.Segment(gx, ps)], (h, h), act="relu", want_f32=False, want_g8=True)))
print("fused                %.1f us" % timeit(lambda: ops.conv2d_fused([ops.Segment(ga, pb), ops.Segment(gx, ps)], (h, h), act="relu", want_f32=False, want_g8=True)))
print("fused, skip K loop   %.1f us" % timeit(lambda: ops.conv2d_fused([ops.Segment(ga, pb), ops.Segment(gx, ps)], (h, h), act="relu", want_f32=False, want_g8=True, reserved=1)))
