"""development probe: one generator network alone (for rocprofv3 --kernel-trace); prints its launch plan"""
import sys
import torch
sys.path.insert(0, ".")
import mpgan_amd
from mpgan_amd import multipass as MP
which, prec = int(sys.argv[1]), int(sys.argv[2])
CFG = [dict(first_gen=True, filter_size=3, start_fms=256, max_fms=256, add_adj=True, first_nn_arch=True, use_res_net=True),
       dict(first_gen=False, filter_size=5, start_fms=192, max_fms=192, use_res_net=True),
       dict(first_gen=False, filter_size=5, start_fms=192, max_fms=96, use_res_net=False)]
g = MP.Generator("growing_gen", dict(tile_low=64, up_res=8, channels=4, **CFG[which]), None, prec, seed=100)
nb = 8 if which == 0 else 2
x = torch.randn(nb, 64, 64, 6 if which == 0 else 4, device="cuda")
y = None if which == 0 else torch.rand(nb, 512, 512, device="cuda")
for _ in range(4):
    g(x, y)
torch.cuda.synchronize()
for e in g.sess.plan_summary(g.sampler):
    if e["kind"] == "conv2d_fused":
        print("PLAN cout=%d prec=%d segs=%s" % (e["cout"], e["prec"], [(s["cin"], s["kernel"], s["up_log2"]) for s in e["segments"]]))
