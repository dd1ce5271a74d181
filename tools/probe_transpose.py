"""HBM-bound companions: mean time and GB/s of mpg_volume_transpose per permutation (256^3 and 512^3, 2 x 4 x N^3
algorithmic bytes), of the cutoff kernel and of the axis zoom; 50 launches each, events on the launch stream."""
import sys
sys.path.insert(0, ".")
import torch
import mpgan_amd
from mpgan_amd import ops
dev = "cuda:0"
def timeit(f, iters=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print("| op | size | mean us | GB/s | frac of 8 TB/s |")
print("|---|---|---|---|---|")
for n in (256, 512):
    v = torch.rand((n, n, n), device=dev)
    nbytes = 2 * 4 * n ** 3
    for perm in ((2, 0, 1), (1, 2, 0), (2, 1, 0), (1, 0, 2), (0, 2, 1)):
        for cut in (0.0, 5e-4):
            us = timeit(lambda: ops.volume_transpose(v, perm, cutoff=cut))
            print("| transpose %s cutoff %g | %d^3 | %.1f | %.0f | %.2f |" % (perm, cut, n, us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000))
    us = timeit(lambda: ops.cutoff(v))
    print("| cutoff | %d^3 | %.1f | %.0f | %.2f |" % (n, us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000))
low = torch.rand((64, 64, 64, 4), device=dev)
for ax in range(3):
    us = timeit(lambda: ops.axis_zoom_linear(low, ax, 8))
    b = 4 * 4 * (64 ** 3 + 8 * 64 ** 3)
    print("| axis zoom x8 axis %d | 64^3x4 | %.1f | %.0f | %.2f |" % (ax, us, b / us / 1e3, b / us / 1e3 / 8000))
