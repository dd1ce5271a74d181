"""diagnostic (library built with -DMPG_WG_DIAG=1): where the waves of wgrad_mfma_kernel<5,1,3> spend their cycles"""
import ctypes, os, sys
sys.path.insert(0, ".")
import torch
import mpgan_amd
from mpgan_amd import train_ops, ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((16, 256, 256, 128), device=dev, generator=g).relu_()
dy = torch.randn((16, 256, 256, 128), device=dev, generator=g) * 1e-4
da, xa = ops.absmax(dy), train_ops.unit_amax(x.device)
L = ctypes.CDLL(os.environ["MPGAN_LIB_OVERRIDE"])
buf = (ctypes.c_ulonglong * 8)()
xg, dg = ops.to_g8(x), ops.to_g8(dy, amax=da)
train_ops.conv2d_wgrad_g8(xg, dg, 5, 5, 0.025, 3, None, da); torch.cuda.synchronize()
L.mpg_debug_wg_diag(buf, 1)
train_ops.conv2d_wgrad_g8(xg, dg, 5, 5, 0.025, 3, None, da); torch.cuda.synchronize()
L.mpg_debug_wg_diag(buf, 1)
n = buf[6]
print("waves sampled %d, chunks per wave %.1f" % (n, buf[7] / n))
for name, v in zip(("fetch issue", "k-steps", "stash", "barrier", "loop total", "wait vmcnt"), buf[:6]):
    print("%-12s %10.0f cycles per wave  (%.0f per chunk)" % (name, v / n, v / max(buf[7], 1)))
