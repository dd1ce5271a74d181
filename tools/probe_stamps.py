"""diagnostic: per-phase cycle sums of the F16F8 K loop (library built with -DMPG_STAMPS=1, MPGAN_LIB_OVERRIDE), b1.B and
b2.A launches; prints the mean over waves / blocks of the four sums divided by the number of stages."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import mpgan_amd
from mpgan_amd import ops
dev = "cuda:0"
N, H = 8, 256
def one(cin, cout, k, extra, waves):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, H, H, cin), device=dev, generator=g).relu_()
    w = torch.randn((k, k, cin, cout), device=dev, generator=g)
    segs = [ops.Segment(x, ops.pack_conv_weights(w, wscale=0.05, prec=2))]
    if extra:
        x2 = torch.randn((N, H, H, extra), device=dev, generator=g).relu_()
        w2 = torch.randn((1, 1, extra, cout), device=dev, generator=g)
        segs.append(ops.Segment(x2, ops.pack_conv_weights(w2, wscale=0.05, prec=2)))
    out = torch.zeros((N, H, H, cout), device=dev)
    for _ in range(3):
        ops.conv2d_fused(segs, (H, H), act="relu", out=out, want_g8c=True, reserved=8 | 2)
    torch.cuda.synchronize()
    nblk = N * (H // 16) * (H // 32)
    st = out.view(torch.int32).reshape(-1)[: nblk * waves * 4].reshape(nblk, waves, 4).cpu().numpy().astype(np.float64)
    stages = (cin // 8) * 25 / 8.0
    print("cin %d cout %d: %d blocks x %d waves, %.1f stages" % (cin, cout, nblk, waves, stages))
    for name, sel in (("all blocks", slice(None)), ("block 0", slice(0, 1)), ("block 700", slice(700, 701))):
        m = st[sel].mean(axis=(0, 1)) / stages
        print("  %-10s per stage: dma-issue %7.0f  fp16 %7.0f  fp8 %7.0f  barrier+vmcnt %7.0f  | sum %7.0f (100 MHz ticks? see clock)" % ((name,) + tuple(m) + (m.sum(),)))
    w0 = st[:, :, :].mean(axis=0) / stages
    print("  per wave (dma, f16, f8, bar):", np.round(w0).astype(int).tolist())
one(128, 128, 5, 8, 8)
one(128, 32, 5, None, 4)
