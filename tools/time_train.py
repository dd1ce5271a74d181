"""Times the 4x GAN training iteration (C3 of BASELINE.json) and prints a per-kernel breakdown hint.
usage: python tools/time_train.py [tileSizeLow] [batch] [steps]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from mpgan_amd.train import Trainer4x  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 16
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
tr = Trainer4x(tileSizeLow=tile, upRes=4, n_inputChannels=4, batch_norm=True)
rng = np.random.default_rng(0)
xs = torch.as_tensor(rng.random((batch, tile * tile * 4)).astype(np.float32), device="cuda:0")
ys = torch.as_tensor(rng.random((batch, (tile * 4) ** 2)).astype(np.float32), device="cuda:0")
for _ in range(3):
    tr.train_step(xs, ys)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    d, g = tr.train_step(xs, ys)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print("tile %d batch %d: %.2f ms / iteration (%.1f it/s, %.0f tiles/s) disc_loss %.4f gen_loss %.4f"
      % (tile, batch, dt * 1e3, 1 / dt, batch / dt, float(d.detach()), float(g.detach())))
for _ in range(2):
    tr.train_step_graphed(xs, ys)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    d, g = tr.train_step_graphed(xs, ys)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print("  hipGraph replay: %.2f ms / iteration (%.1f it/s, %.0f tiles/s) disc_loss %.4f gen_loss %.4f"
      % (dt * 1e3, 1 / dt, batch / dt, float(d), float(g)))
