"""diagnostic: the full-size 8x three-network pipeline with 1, 2 and 3 pass lanes must give the same bits"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import mpgan_amd
from mpgan_amd import multipass as MP
from mpgan_amd.synthetic import synthetic_volume
CFG = [dict(first_gen=True, filter_size=3, start_fms=256, max_fms=256, add_adj=True, first_nn_arch=True, use_res_net=True),
       dict(first_gen=False, filter_size=5, start_fms=192, max_fms=192, use_res_net=True),
       dict(first_gen=False, filter_size=5, start_fms=192, max_fms=96, use_res_net=False)]
low = torch.as_tensor(synthetic_volume(64, 4, 0)).cuda()
gens = [MP.Generator("growing_gen", dict(tile_low=64, up_res=8, channels=4, **c), None, 2, seed=100 + i) for i, c in enumerate(CFG)]
res = {}
for lanes in (1, 2, 3, 2):
    MP.set_pass_lanes(lanes)
    out = MP.multipass_8x(gens, low, 8)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    if lanes in res:
        print("lanes %d repeat identical: %s" % (lanes, np.array_equal(o, res[lanes])))
    res[lanes] = o
for lanes in (2, 3):
    d = np.abs(res[lanes] - res[1])
    print("lanes %d vs 1: identical %s, differing voxels %d, max %.3e" % (lanes, np.array_equal(res[lanes], res[1]), int((d > 0).sum()), float(d.max())))
