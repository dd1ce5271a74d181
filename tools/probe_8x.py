"""development probe: time the 8x two/three-pass pipeline on one 64^3 -> 512^3 volume (BASELINE config C4, one GPU)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import mpgan_amd
from mpgan_amd import multipass as MP
from mpgan_amd.synthetic import synthetic_volume

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nets = int(sys.argv[2]) if len(sys.argv) > 2 else 3
CFG = [dict(first_gen=True, filter_size=3, start_fms=256, max_fms=256, add_adj=True, first_nn_arch=True, use_res_net=True),
       dict(first_gen=False, filter_size=5, start_fms=192, max_fms=192, use_res_net=True),
       dict(first_gen=False, filter_size=5, start_fms=192, max_fms=96, use_res_net=False)]
GF = [136.63, 405.48, 397.21]
low = torch.as_tensor(synthetic_volume(64, 4, 0)).cuda()
gens = [MP.Generator("growing_gen", dict(tile_low=64, up_res=8, channels=4, **c), None, prec, seed=100 + i) for i, c in enumerate(CFG[:nets])]
for it in range(2):
    torch.cuda.synchronize(); t = time.time()
    out = MP.multipass_8x(gens, low, 8)
    torch.cuda.synchronize(); dt = time.time() - t
    print("prec %d, %d nets: %.3f s/volume, %.1f TFLOP/s algorithmic" % (prec, nets, dt, sum(GF[:nets]) * 512 / 1e3 / dt), flush=True)
# per-network time
xs = torch.randn(8, 64, 64, 6, device="cuda")
for i, g in enumerate(gens):
    nb = 8 if i == 0 else 2
    x = torch.randn(nb, 64, 64, 6 if i == 0 else 4, device="cuda")
    y = None if i == 0 else torch.rand(nb, 512, 512, device="cuda")
    for _ in range(2): g(x, y)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5): g(x, y)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5 / nb
    print("net%d: %.3f ms/slice, %.1f TFLOP/s" % (i + 1, dt * 1e3, GF[i] / 1e3 / dt), flush=True)
