"""host TileCreator vs tiles_device.DeviceTileCreator: tiles per second of selectRandomTiles / selectRandomTempoTiles at the
C3 training shape (tileSize 16 -> 64^2, 4 low-res channels, batch 16), plain and augmented, and the 4x training iteration
fed by each.  Synthetic frames."""
import contextlib, io, random, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import mpgan_amd
from mpgan_amd import tilecreator_t as tc
from mpgan_amd.tiles_device import DeviceTileCreator
from mpgan_amd.train import Trainer4x

tile, sim, up, C, batch, frames = 16, 64, 4, 4, 16, 40
rng = np.random.default_rng(0)
low = rng.random((frames, 1, sim, sim, C * 3)).astype(np.float32)
high = rng.random((frames, 1, sim * up, sim * up, 3)).astype(np.float32)
kw = dict(tileSizeLow=tile, simSizeLow=sim, upres=up, dim=2, dim_t=3, densityMinimum=0.0, channelLayout_low="d,vx,vy,vz",
          channelLayout_high="d")
with contextlib.redirect_stdout(io.StringIO()):
    host, dev = tc.TileCreator(**kw), DeviceTileCreator(**kw)
    for t in (host, dev):
        t.initDataAugmentation(rot=2, minScale=0.85, maxScale=1.15, flip=True)
        t.addData(low.copy(), high.copy())


def rate(fn, n_tiles, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return n_tiles * reps / (time.time() - t0)


print("tiles/s                         host      device")
for name, aug in (("plain", False), ("augmented (rot 2, scale, flip)", True)):
    random.seed(1); np.random.seed(1)
    h = rate(lambda: host.selectRandomTiles(batch, augment=aug), batch)
    random.seed(1); np.random.seed(1)
    d = rate(lambda: dev.selectRandomTilesDevice(batch, augment=aug), batch)
    print("%-30s %9.0f %9.0f" % (name, h, d))
    h = rate(lambda: host.selectRandomTempoTiles(batch, True, aug, n_t=3, dt=0.5), batch)
    d = rate(lambda: dev.selectRandomTempoTilesDevice(batch, True, aug, n_t=3, dt=0.5), batch)
    print("%-30s %9.0f %9.0f   (coherent triples + positions)" % (name, h, d))

tr = Trainer4x(tileSizeLow=tile, upRes=up, n_inputChannels=C, batch_norm=True, device="cuda:0")
n_in, n_out = tile * tile * C, (tile * up) ** 2
for name, aug in (("plain", False), ("augmented", True)):
    for who, get in (("host tiles", lambda: host.selectRandomTiles(batch, augment=aug)),
                     ("device tiles", lambda: dev.selectRandomTilesDevice(batch, augment=aug))):
        def it():
            bx, by = get()
            tr.train_step(bx[..., :C].reshape(-1, n_in) if bx.shape[-1] != C else bx.reshape(-1, n_in), by[..., :1].reshape(-1, n_out))
        for _ in range(3):
            it()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20):
            it()
        torch.cuda.synchronize()
        print("training iteration (eager) with %-12s %-10s: %.2f ms" % (who, name, (time.time() - t0) / 20 * 1e3))
