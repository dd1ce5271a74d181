"""development probe: the four small-channel layers of the 4x generator (conv_small_kernel), us per launch of 8 slices
of 256^2; run once per library (MPGAN_LIB_OVERRIDE) for an A/B"""
import sys
import torch
sys.path.insert(0, ".")
import mpgan_amd
from mpgan_amd import ops
dev = "cuda:0"
N, H = 8, 256
def run(cin, cout, k, extra, prec=2, iters=50):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, H, H, cin), device=dev, generator=g).relu_()
    w = torch.randn((k, k, cin, cout), device=dev, generator=g)
    segs = [ops.Segment(x, ops.pack_conv_weights(w, wscale=0.05, prec=prec))]
    if extra:
        x2 = torch.randn((N, H, H, extra), device=dev, generator=g).relu_()
        w2 = torch.randn((1, 1, extra, cout), device=dev, generator=g)
        segs.append(ops.Segment(x2, ops.pack_conv_weights(w2, wscale=0.05, prec=prec)))
    f = lambda: ops.conv2d_fused(segs, (H, H), act="relu", want_f32=False, want_g8=True)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tot = 0.0
for name, cin, cout, ex in (("b0.A 1->2", 1, 2, None), ("b0.B 2->8+s1", 2, 8, 1), ("b3.A 8->2", 8, 2, None), ("b3.B 2->1+s8", 2, 1, 8)):
    t = run(cin, cout, 5, ex)
    tot += t
    print("%-16s %7.1f us" % (name, t), flush=True)
print("%-16s %7.1f us" % ("sum", tot))


def run_pair(cin, cmid, cout, up, out_f32, iters=50):
    """the same two blocks as ONE launch (mpg_conv2d_small_pair): resBlock 0 reads the low-res slice with the x4 upsample
    fused (pass 1) and emits G8, resBlock 3 emits the fp32 slice"""
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, H >> up, H >> up, cin), device=dev, generator=g).relu_()
    pk = lambda k, ci, co: ops.pack_conv_weights(torch.randn((k, k, ci, co), device=dev, generator=g), wscale=0.05, prec=2)
    pa, pb, ps = pk(5, cin, cmid), pk(5, cmid, cout), pk(1, cin, cout)
    xg = ops.to_g8(x)
    f = lambda: ops.conv2d_small_pair(xg, 0, up, pa, pb, ps, (H, H), act_a="relu", act_b="relu", want_f32=out_f32, want_g8=not out_f32)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("%-16s %7.1f us   (one launch; up x4 fused)" % ("b0 1->2->8 pair", run_pair(1, 2, 8, 2, False)))
print("%-16s %7.1f us   (one launch; no upsample)" % ("b0 1->2->8 pair", run_pair(1, 2, 8, 0, False)))
print("%-16s %7.1f us   (one launch)" % ("b3 8->2->1 pair", run_pair(8, 2, 1, 0, True)))
