"""development probe: end-to-end parity (vs the oracle, 12 slices per pass, as bench.py measures it) and time of the C2
pipeline with single layers of the generator moved to another precision (Session prec_map)"""
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import torch
import bench
from mpgan_amd import multipass as MP
from mpgan_amd.synthetic import synthetic_volume

dev = "cuda:0"
SIM, UP = bench.SIM, bench.UP
maps = {"default": None, "b2.A x1": [("g_cA2", 1)], "b2.A+b2.B x1": [("g_cA2", 1), ("g_cB2", 1)], "b1.A f6": [("g_cA1", 2)]}
lows_np = [synthetic_volume(SIM, 1, i) for i in range(4)]
lows = [torch.as_tensor(v).to(dev) for v in lows_np]
runs = {}
gens = {}
for name, pm in maps.items():
    cfg1 = dict(tile_low=SIM, up_res=UP, channels=1, upsampling_mode=2, batch_norm=True)
    cfg2 = dict(tile_low=SIM, up_res=UP, channels=1, upsampling_mode=1, batch_norm=True)
    g1 = MP.Generator("gen_resnet", cfg1, None, 2, device=dev, seed=777, prec_map=pm)
    g2 = MP.Generator("gen_resnet", cfg2, None, 2, device=dev, seed=778, prec_map=pm)
    gens[name] = (g1, g2)
    final, v1 = MP.two_pass_4x(g1, g2, lows[0], UP, batch=8)
    runs[name] = (final.cpu().numpy(), v1.cpu().numpy())
    lanes = [(g1.clone(), g2.clone())]
    for _ in range(2):
        MP.two_pass_4x_batch(g1, g2, lows, UP, batch=8, lanes=lanes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        MP.two_pass_4x_batch(g1, g2, lows, UP, batch=8, lanes=lanes)
    torch.cuda.synchronize()
    print("%-14s %.2f volumes/s" % (name, 12 / (time.perf_counter() - t0)), flush=True)
g1, g2 = gens["default"]
base, parity, idx = bench.cpu_baseline_and_parity(g1.params(), g2.params(), lows_np[0], runs)
for k, v in parity.items():
    print("%-14s pass1 %.2e pass2 %.2e" % (k, v["pass1"], v["pass2_given_pass1"]))
