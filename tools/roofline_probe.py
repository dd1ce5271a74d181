#!/usr/bin/env python3
"""Launches the dominant kernel of the C2 bench (resBlock 1 convB, 5x5 128->128 + 1x1 8->128 shortcut,
8 slices of 256^2) a few times; run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
(separate passes) to get its HBM traffic.  Same launch as bench.py's dominant_kernel_roofline()."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import mpgan_amd  # noqa: F401
from mpgan_amd import ops

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = "cuda:0"
n, h, w = 8, 256, 256
g = torch.Generator(device=dev).manual_seed(1)
a = torch.randn((n, h, w, 128), device=dev, generator=g).relu_()
x = torch.randn((n, h, w, 8), device=dev, generator=g).relu_()
wb = torch.randn((5, 5, 128, 128), device=dev, generator=g)
ws = torch.randn((1, 1, 8, 128), device=dev, generator=g)
bias = torch.randn(128, device=dev, generator=g)
pkb = ops.pack_conv_weights(wb, wscale=float(np.sqrt(2.0 / 3200)), prec=prec)
pks = ops.pack_conv_weights(ws, wscale=float(np.sqrt(2.0 / 8)), prec=prec)
segs = [ops.Segment(a, pkb), ops.Segment(x, pks)]
for _ in range(iters):
    ops.conv2d_fused(segs, (h, w), bias=bias, act="relu", want_f32=False, want_g8=True)
torch.cuda.synchronize()
print("done")
