"""diagnostic: full-size volumes through the lane pipeline vs one lane; where do they differ?"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import mpgan_amd
from mpgan_amd import multipass as MP
from mpgan_amd.synthetic import synthetic_volume
dev = "cuda:0"
SIM, UP = 64, 4
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg1 = dict(tile_low=SIM, up_res=UP, channels=1, upsampling_mode=2, batch_norm=True)
cfg2 = dict(tile_low=SIM, up_res=UP, channels=1, upsampling_mode=1, batch_norm=True)
g1 = MP.Generator("gen_resnet", cfg1, None, 2, device=dev, seed=777)
g2 = MP.Generator("gen_resnet", cfg2, None, 2, device=dev, seed=778)
lows = [torch.as_tensor(synthetic_volume(SIM, 1, i)).to(dev) for i in range(8)]
ref = [o.cpu().numpy() for o in MP.two_pass_4x_batch(g1, g2, lows, UP, batch=8)]
ref2 = [o.cpu().numpy() for o in MP.two_pass_4x_batch(g1, g2, lows, UP, batch=8)]
print("one lane repeatable:", all(np.array_equal(a, b) for a, b in zip(ref, ref2)))
lanes = [(g1.clone(), g2.clone()) for _ in range(nl - 1)]
for rep in range(3):
    got = [o.cpu().numpy() for o in MP.two_pass_4x_batch(g1, g2, lows, UP, batch=8, lanes=lanes)]
    for i, (a, b) in enumerate(zip(got, ref)):
        if not np.array_equal(a, b):
            d = np.abs(a - b)
            idx = np.argwhere(d > 0)
            print("rep %d volume %d: %d voxels differ, max %.3e; z range %d..%d y %d..%d x %d..%d" % (
                rep, i, len(idx), d.max(), idx[:, 0].min(), idx[:, 0].max(), idx[:, 1].min(), idx[:, 1].max(),
                idx[:, 2].min(), idx[:, 2].max()))
        else:
            print("rep %d volume %d: identical" % (rep, i))
import ctypes, os
if os.environ.get("MPGAN_LIB_OVERRIDE", "").endswith("diag.so"):
    from mpgan_amd import _lib
    L = ctypes.CDLL(os.environ["MPGAN_LIB_OVERRIDE"])
    buf = (ctypes.c_uint * 2)()
    torch.cuda.synchronize()
    print("rc", L.mpg_debug_small_diag(buf), "small-kernel LDS check: bad weights %d, bad tile words %d" % (buf[0], buf[1]))
