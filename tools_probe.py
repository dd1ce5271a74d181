"""scratch probe (GPU box): end-to-end error and timing of the 4x generator per precision map"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import mpgan_amd
from mpgan_amd import multipass as MP
from mpgan_amd.synthetic import synthetic_volume

def rel(a, b): return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))

sim, up = 64, 4
low = synthetic_volume(sim, 1, 0)
lowd = torch.as_tensor(low).cuda()
configs = {
  "all3": (3, None),
  "all1": (1, None),
  "1B@1": (3, [("g_cB1", 1)]),
  "1B,2A@1": (3, [("g_cB1", 1), ("g_cA2", 1)]),
  "1A,1B,2A@1": (3, [("g_cB1", 1), ("g_cA2", 1), ("g_cA1", 1)]),
}
outs = {}
for name, (prec, pm) in configs.items():
    g1 = MP.Generator("gen_resnet", dict(tile_low=sim, up_res=up, channels=1, upsampling_mode=2), None, prec, seed=31, prec_map=pm)
    g2 = MP.Generator("gen_resnet", dict(tile_low=sim, up_res=up, channels=1, upsampling_mode=1), None, prec, seed=32, prec_map=pm)
    for it in range(3):
        torch.cuda.synchronize(); t = time.time()
        out, v1 = MP.two_pass_4x(g1, g2, lowd, up, batch=8)
        torch.cuda.synchronize(); dt = time.time() - t
    outs[name] = (out.cpu().numpy().astype(np.float64), v1.cpu().numpy().astype(np.float64))
    print("%-12s %.4fs/volume %.1f TFLOP/s-equiv  final err vs all3 %.2e  pass1 %.2e" % (
        name, dt, 36.71 / dt, rel(outs[name][0], outs["all3"][0]), rel(outs[name][1], outs["all3"][1])), flush=True)
