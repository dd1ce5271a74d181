"""diagnostic: per-phase cycle sums of the F16F6 K loop (library built with -DMPG_STAMPS=1, MPGAN_LIB_OVERRIDE), b1.B and
b2.A, one launch each: per wave and stage, s_memtime cycles from the barrier release to the first satisfied fragment wait,
through the fp16 groups, through the correction steps, and waiting at the next barrier"""
import os
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import mpgan_amd
from mpgan_amd import ops
dev = "cuda:0"
N, H = 8, 256
for name, cin, cout, waves in (("b1.B 128->128", 128, 128, 8), ("b2.A 128->32", 128, 32, 4), ("96->96 3x3", 96, 96, 8)):
    k = 3 if "3x3" in name else 5
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, H, H, cin), device=dev, generator=g).relu_()
    w = torch.randn((k, k, cin, cout), device=dev, generator=g)
    segs = [ops.Segment(x, ops.pack_conv_weights(w, wscale=0.05, prec=2))]
    out = torch.zeros((N, H, H, cout), device=dev)
    for _ in range(3):
        ops.conv2d_fused(segs, (H, H), act="relu", out=out, want_g8=True, reserved=8 | 2)
    torch.cuda.synchronize()
    nblk = N * (H // 16) * (H // 32)
    st = out.view(torch.int32).flatten()[:nblk * waves * 4].cpu().numpy().astype(np.int64).reshape(nblk, waves, 4)
    nstage = ((cin // 8) * (k * k if k * k >= 16 else 12) + 7) // 8
    per = st.mean(axis=0) / max(nstage - 2, 1)
    print(name, "stages", nstage, " cycles per stage (head, fp16, bf6, barrier) per wave:")
    for wv in range(waves):
        print("   wave %d: %s  sum %d" % (wv, np.round(per[wv]).astype(int).tolist(), int(per[wv].sum())))
