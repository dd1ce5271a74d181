"""development probe: mpg_conv2d_wgrad_g8 on the convolution shapes of the training steps (16 tiles), us per call; run once
per library (MPGAN_LIB_OVERRIDE) to compare the one-filter-row kernel with the ring form"""
import sys
import torch
sys.path.insert(0, ".")
import mpgan_amd  # noqa: F401
from mpgan_amd import ops, train_ops
dev = "cuda:0"
SHAPES = [  # k, cin, cout, tile
    (5, 8, 128, 256), (5, 128, 128, 256), (5, 128, 32, 256), (5, 32, 8, 256), (5, 128, 128, 64), (5, 8, 128, 64), (5, 128, 32, 64),
    (4, 32, 64, 128), (4, 64, 128, 64), (4, 128, 128, 64), (3, 64, 64, 128), (3, 128, 128, 64), (3, 32, 32, 256), (3, 256, 256, 32),
]
out = []
for k, cin, cout, t in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((16, t, t, cin), device=dev, generator=g).relu_()
    dy = torch.randn((16, t, t, cout), device=dev, generator=g) * 1e-4
    am = ops.absmax(dy)
    xg, dg = ops.to_g8(x), ops.to_g8(dy, amax=am)
    f = lambda: train_ops.conv2d_wgrad_g8(xg, dg, k, k, 0.025, 3, None, am)
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(8):
        f()
    e1.record(); torch.cuda.synchronize()
    out.append("%dx%d %d->%d @%d: %.0f" % (k, k, cin, cout, t, e0.elapsed_time(e1) / 8 * 1e3))
print(" | ".join(out))
