"""f2: the device tile pipeline (multi-pass-gan_amd/tiles_device.py + csrc/mpgan_tiles.hip) against the host TileCreator
under identical seeds -- which itself reproduces the reference's batches bit for bit (tests/test_tilecreator.py).
Plain batches must be bit-equal; augmented ones agree to float32 interpolation rounding."""
import contextlib
import io
import os
import random
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from tile_scenarios import SCENARIOS, make_frames  # noqa: E402

pytestmark = pytest.mark.gpu


def _pair(sc, low, high):
    import mpgan_amd  # noqa: F401
    from mpgan_amd import tilecreator_t as TC
    from mpgan_amd.tiles_device import DeviceTileCreator
    kw = dict(tileSizeLow=sc["tile"], simSizeLow=sc["sim"], upres=sc["upres"], dim=sc["dim"], dim_t=sc["dim_t"],
              densityMinimum=sc["dens_min"], channelLayout_low=sc["low"], channelLayout_high=sc["high"], partTrain=0.7, partTest=0.3)
    with contextlib.redirect_stdout(io.StringIO()):
        host, dev = TC.TileCreator(**kw), DeviceTileCreator(**kw)
        for t in (host, dev):
            if "aug" in sc:
                t.initDataAugmentation(**sc["aug"])
            t.addData(low.copy(), high.copy())
    return host, dev


def _seed(i):
    random.seed(100 + i)
    np.random.seed(200 + i)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_device_tiles_match_host(name):
    sc = SCENARIOS[name]
    low, high = make_frames(sc)
    host, dev = _pair(sc, low, high)
    calls = {"batch": dict(n=5, tr=True, aug=False, t=1), "batch_test": dict(n=3, tr=False, aug=False, t=1),
             "batch_t": dict(n=4, tr=True, aug=False, t=2), "aug": dict(n=4, tr=True, aug=True, t=1),
             "aug_t": dict(n=4, tr=True, aug=True, t=2)}
    for i, call in enumerate(sc["calls"]):
        if call in calls:
            c = calls[call]
            _seed(i)
            hl, hh = host.selectRandomTiles(c["n"], isTraining=c["tr"], augment=c["aug"], tile_t=c["t"])
            state = (random.getstate(), np.random.get_state())
            _seed(i)
            dl, dh = dev.selectRandomTilesDevice(c["n"], isTraining=c["tr"], augment=c["aug"], tile_t=c["t"])
            # the random streams were consumed identically
            assert random.getstate() == state[0] and all(np.array_equal(a, b) for a, b in zip(np.random.get_state(), state[1]) if isinstance(a, np.ndarray))
            dl, dh = dl.cpu().numpy(), dh.cpu().numpy()
            assert dl.shape == hl.shape and dh.shape == hh.shape, (call, dl.shape, hl.shape)
            if c["aug"] and (sc["aug"]["rot"] == 2 or sc["aug"]["minScale"] != 1):
                np.testing.assert_allclose(dl, hl, rtol=2e-5, atol=2e-6, err_msg=call)
                np.testing.assert_allclose(dh, hh, rtol=2e-5, atol=2e-6, err_msg=call)
            else:
                assert np.array_equal(dl, hl.astype(np.float32)) and np.array_equal(dh, hh.astype(np.float32)), call
        elif call == "tempo":
            _seed(i)
            a, b, p = host.selectRandomTempoTiles(6, isTraining=True, augment=False, n_t=3, dt=0.5)
            _seed(i)
            da, db, dp = dev.selectRandomTempoTilesDevice(6, isTraining=True, augment=False, n_t=3, dt=0.5)
            assert np.array_equal(da.cpu().numpy(), a) and np.array_equal(db.cpu().numpy(), b)
            np.testing.assert_allclose(dp.cpu().numpy(), p, rtol=1e-5, atol=1e-5)


def test_rot90_and_flip_orientation_bookkeeping():
    """quarter turns + flips composed into one signed axis permutation: every cube rotation of tilecreator_t, with the
    vector components following (packed frames) -- against np.rot90 / np.flip on the host"""
    import mpgan_amd  # noqa: F401
    from mpgan_amd import tilecreator_t as TC
    from mpgan_amd.tiles_device import _Orientation, tile_orient
    rng = np.random.default_rng(0)
    for dim in (2, 3):
        src = rng.standard_normal((1 if dim == 2 else 5, 5, 5, 8)).astype(np.float32)        # two packed (d,vx,vy,vz) frames
        maps = {0: TC.ChannelMap("d,vx,vy,vz", dim)}
        aug = TC._Augment(maps, [])
        for seq in TC.CUBE_ROTATIONS[dim]:
            for flip_axis in (None, 0, 1, 2):
                if dim == 2 and flip_axis == 0:
                    continue
                pair = {0: src.copy()}
                o = _Orientation()
                for plane in seq:
                    pair = aug.quarter_turn(pair, plane)
                    o.quarter_turn(plane, True)
                if flip_axis is not None:
                    pair = aug.flip(pair, flip_axis)
                    o.flip_axis(flip_axis)
                cmap, csign = list(range(8)), [1.0] * 8
                for f in range(2):
                    for comp in range(3):
                        cmap[f * 4 + 1 + comp] = f * 4 + 1 + o.comp_src[comp]
                        csign[f * 4 + 1 + comp] = o.comp_sign[comp]
                want = np.ascontiguousarray(pair[0])
                out = torch.empty(want.shape, dtype=torch.float32, device="cuda:0")
                tile_orient(torch.as_tensor(src).cuda(), [0, 0, 0], src.shape[:3], o.perm, o.flip, cmap, csign, out)
                assert np.array_equal(out.cpu().numpy(), want), (dim, seq, flip_axis)


def test_resample_kernel_vs_scipy():
    """mpg_resample_affine against scipy.ndimage.zoom / affine_transform (order 1, mode 'constant'), including the
    rows scipy's zoom zeroes when its last coordinate rounds past the end"""
    import scipy.ndimage
    import mpgan_amd  # noqa: F401
    from mpgan_amd.tiles_device import resample_affine
    rng = np.random.default_rng(4)
    for trial in range(40):
        z = 1 if trial % 2 == 0 else int(rng.integers(2, 6))
        y, x = int(rng.integers(3, 30)), int(rng.integers(3, 30))
        a = rng.standard_normal((z, y, x, 3)).astype(np.float32)
        f = rng.uniform(0.6, 1.8)
        zoom = [1 if z == 1 else f, f, f, 1]
        ref = scipy.ndimage.zoom(a, zoom, order=1, mode="constant", cval=0.0)
        out_shape = ref.shape[:3]
        ratio = [(a.shape[k] - 1) / (out_shape[k] - 1) if out_shape[k] > 1 else 1.0 for k in range(3)]
        got = resample_affine(torch.as_tensor(a).cuda(), out_shape, np.diag(ratio), [0, 0, 0]).cpu().numpy()
        assert np.abs(got - ref).max() < 2e-6, (trial, a.shape, out_shape)
        th = rng.uniform(0, 2 * np.pi)
        m = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])
        c = np.array(a.shape[:3]) / 2 - 0.5
        off = c - m.dot(c)
        ref2 = np.stack([scipy.ndimage.affine_transform(a[..., k], m, off, order=1, mode="constant", cval=0.0) for k in range(3)], -1)
        got2 = resample_affine(torch.as_tensor(a).cuda(), a.shape[:3], m, off).cpu().numpy()
        assert np.abs(got2 - ref2).max() < 2e-6, trial
