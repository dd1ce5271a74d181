#!/usr/bin/env python3
"""development probe (needs a library built with the clock instrumentation): shader clock during the K loop of
resBlock 1 convB (5x5 128->128, 8 slices of 256^2, F16F6)"""
import os, sys, importlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
sys.modules.setdefault("mpgan_amd", importlib.import_module("multi-pass-gan_amd"))
from mpgan_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n, h = 8, 256
x = torch.randn((n, h, h, 128), device=dev, generator=g).relu_()
w = torch.randn((5, 5, 128, 128), device=dev, generator=g) * 0.02
gx = ops.to_g8(x)
pk = ops.pack_conv_weights(w, prec=2)
buf = torch.zeros((n, h, h, 128), device=dev)
for _ in range(20):
    ops.conv2d_fused([ops.Segment(gx, pk)], (h, h), act="relu", want_f32=False, want_g8=True)
torch.cuda.synchronize()
for rep in range(3):
    for _ in range(30):
        ops.conv2d_fused([ops.Segment(gx, pk)], (h, h), act="relu", want_f32=False, want_g8=True)
    ops.conv2d_fused([ops.Segment(gx, pk)], (h, h), act="relu", want_f32=False, want_g8=True, post_add=buf, reserved=4)
    torch.cuda.synchronize()
    v = buf.flatten()[:32].cpu().numpy().reshape(16, 2)
    cyc, rt = v[:, 0], v[:, 1]
    print("K loop of one tile: %.0f shader cycles, %.1f us (100 MHz counter) -> %.0f MHz" % (cyc.mean(), rt.mean() / 100.0, (cyc / rt).mean() * 100.0))
