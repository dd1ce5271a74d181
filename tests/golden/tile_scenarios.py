"""TileCreator call sequences shared by the fixture generator (run on the reference module) and
tests/test_tilecreator.py (run on ours).  Everything random is seeded inside run_scenario."""
import numpy as np

SCENARIOS = {
    # 2D, 4x, single frames, sparse density (exercises the minimum-density retry loop)
    "tc2d": dict(dim=2, tile=8, sim=16, upres=4, dim_t=1, frames=6, low="d,vx,vy,vz", high="d",
                 dens_min=0.25, sparse=True, calls=("tiles", "batch", "batch_test")),
    # 2D coherent triples for the temporal discriminator
    "tc2d_t3": dict(dim=2, tile=8, sim=16, upres=2, dim_t=3, frames=5, low="d,vx,vy,vz", high="d",
                    dens_min=0.02, sparse=False, calls=("batch_t", "tempo")),
    # 2D augmentation: scale + free rotation + flip
    "tc2d_aug": dict(dim=2, tile=8, sim=32, upres=2, dim_t=1, frames=4, low="d,vx,vy,vz", high="d",
                     dens_min=0.02, sparse=False, aug=dict(rot=2, minScale=0.8, maxScale=1.2, flip=True),
                     calls=("aug",)),
    # 2D augmentation of coherent frames, flip only
    "tc2d_flip_t2": dict(dim=2, tile=8, sim=16, upres=2, dim_t=2, frames=4, low="d,vx,vy,vz", high="d,vx,vy,vz",
                         dens_min=0.02, sparse=False, aug=dict(rot=0, minScale=1, maxScale=1, flip=True),
                         calls=("aug_t",)),
    # 3D plain + augmentation (quaternion rotation, scaling, flip)
    "tc3d": dict(dim=3, tile=4, sim=12, upres=2, dim_t=1, frames=3, low="d,vx,vy,vz", high="d",
                 dens_min=0.02, sparse=False, aug=dict(rot=2, minScale=0.9, maxScale=1.1, flip=True),
                 calls=("tiles", "batch", "aug")),
    # per-axis upres as the multi-pass training uses it (hard-coded 8x along x in getRandomTile)
    "tc3d_axis": dict(dim=3, tile=4, sim=8, upres=[1, 1, 8], dim_t=1, frames=3, low="d,vx,vy,vz", high="d",
                      dens_min=0.02, sparse=False, calls=("batch",)),
}


def make_frames(sc):
    rng = np.random.default_rng(abs(hash_name(sc)) % (2 ** 31))
    s, up = sc["sim"], sc["upres"]
    cl = len(sc["low"].split(",")) * sc["dim_t"]
    ch = len(sc["high"].split(",")) * sc["dim_t"]
    z = 1 if sc["dim"] == 2 else s
    upv = [up, up, up] if np.isscalar(up) else list(up)
    zh = 1 if sc["dim"] == 2 else s * upv[0]
    low = rng.random((sc["frames"], z, s, s, cl)).astype(np.float32)
    high = rng.random((sc["frames"], zh, s * upv[1], s * upv[2], ch)).astype(np.float32)
    if sc["sparse"]:
        low[:, :, :, : s // 2, 0::4] *= 0.05
    return low, high


def hash_name(sc):
    # stable across processes (no str hash)
    key = "%d-%s-%s-%d-%d" % (sc["dim"], sc["tile"], sc["sim"], sc["dim_t"], sc["frames"])
    return sum((i + 1) * ord(c) for i, c in enumerate(key))


def run_scenario(mod, sc, low, high, py_random, np_mod):
    """mod: a tilecreator_t module (reference or ours). Returns a dict of arrays."""
    tc = mod.TileCreator(tileSizeLow=sc["tile"], simSizeLow=sc["sim"], upres=sc["upres"], dim=sc["dim"],
                         dim_t=sc["dim_t"], densityMinimum=sc["dens_min"], channelLayout_low=sc["low"],
                         channelLayout_high=sc["high"], partTrain=0.7, partTest=0.3)
    if "aug" in sc:
        tc.initDataAugmentation(**sc["aug"])
    tc.addData(low.copy(), high.copy())
    out = {"borders": np.asarray(tc.setBorders)}
    for i, call in enumerate(sc["calls"]):
        py_random.seed(100 + i)
        np_mod.random.seed(200 + i)
        if call == "tiles":
            dl, dh = tc.getDatum(1)
            out["datum_low"], out["datum_high"] = dl, dh
            tl, th = tc.getFrameTiles(2)
            out["frame_tiles_low"], out["frame_tiles_high"] = tl, th
            strided = tc.createTiles(dl, tc.tile_shape_low, strides=sc["tile"] // 2)
            out["strided_tiles"] = strided
            n = [(dl.shape[k] - tc.tile_shape_low[k]) // tc.tile_shape_low[k] + 1 for k in range(3)]
            out["concat"] = tc.concatTiles(tl, n)
            if sc["dim"] == 2:
                out["concat_border"] = tc.concatTiles(tl, n, [0, 1, 2, 0])
        elif call == "batch":
            out["batch_low"], out["batch_high"] = tc.selectRandomTiles(5, isTraining=True)
        elif call == "batch_test":
            out["batch_test_low"], out["batch_test_high"] = tc.selectRandomTiles(3, isTraining=False)
        elif call == "batch_t":
            out["batch_t_low"], out["batch_t_high"] = tc.selectRandomTiles(4, isTraining=True, tile_t=2)
        elif call == "tempo":
            a, b, c = tc.selectRandomTempoTiles(6, isTraining=True, augment=False, n_t=3, dt=0.5)
            out["tempo_low"], out["tempo_high"], out["tempo_pos"] = a, b, c
        elif call == "aug":
            out["aug_low"], out["aug_high"] = tc.selectRandomTiles(4, isTraining=True, augment=True)
        elif call == "aug_t":
            out["aug_t_low"], out["aug_t_high"] = tc.selectRandomTiles(4, isTraining=True, augment=True, tile_t=2)
    return out
