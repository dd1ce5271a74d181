"""Times one stage-3 iteration of the 8x progressive-growing training (C5 per GPU: tileSize 16 -> 128^2,
batch 16, firstNNArch, startFms 256, WGAN-GP).  usage: python tools/time_train8x.py [tile] [batch] [steps]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from mpgan_amd.arch import Cfg8x  # noqa: E402
from mpgan_amd.train import Trainer8x  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 16
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = Cfg8x(tileSizeLow=tile, upRes=8, n_inputChannels=6, start_fms=256, max_fms=256)
tr = Trainer8x(cfg)
rng = np.random.default_rng(0)
xs = torch.as_tensor(rng.random((batch, tile * tile * 6)).astype(np.float32), device="cuda:0")
ys = torch.as_tensor(rng.random((batch, (tile * 8) ** 2)).astype(np.float32), device="cuda:0")
for _ in range(2):
    tr.train_step(xs, ys, 3.0)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    d, g = tr.train_step(xs, ys, 3.0)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print("8x stage 3, tile %d -> %d^2, batch %d: %.1f ms / iteration (%.2f it/s, %.1f tiles/s) disc_loss %.4f gen_loss %.4f"
      % (tile, tile * 8, batch, dt * 1e3, 1 / dt, batch / dt, float(d), float(g)))
