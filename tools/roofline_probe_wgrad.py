#!/usr/bin/env python3
"""Launches the dominant kernel of the training benches (matrix-core weight gradient of a 5x5 128->128 conv on 16
tiles of 256^2: the scaled G8 conversion of dy that the data gradient shares + wgrad_ring_kernel<5,5,1,3>) a few times; run under
`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) to get the HBM traffic per kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mpgan_amd  # noqa: F401
from mpgan_amd import train_ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((16, 256, 256, 128), device=dev, generator=g).relu_()
dy = torch.randn((16, 256, 256, 128), device=dev, generator=g) * 1e-4
# as the training step calls it (train.ConvLayerFn.backward): max |dy| comes from the kernel that produced dy, x is a forward
# activation (split unscaled); `--standalone` reduces both inside the call, as round 1 measured it
standalone = "--standalone" in sys.argv
from mpgan_amd import ops
dy_amax = None if standalone else ops.absmax(dy)
x_amax = None if standalone else train_ops.unit_amax(x.device)
xg = None if standalone else ops.to_g8(x)          # the forward launch's operand, kept for the backward pass
torch.cuda.synchronize()
for _ in range(iters):
    if standalone:
        train_ops.conv2d_wgrad_mfma(x, dy, 5, 5, 0.025, 3, dy_amax, x_amax)
    else:
        dg = ops.to_g8(dy, amax=dy_amax)           # shared with the data-gradient convolution
        train_ops.conv2d_wgrad_g8(xg, dg, 5, 5, 0.025, 3, None, dy_amax)
torch.cuda.synchronize()
print("done")
