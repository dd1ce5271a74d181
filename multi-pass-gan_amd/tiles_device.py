"""``TileCreator`` with the frames resident in HBM: the same random decisions as ``tilecreator_t.TileCreator`` (and
therefore as the reference, tools_wscale/tilecreator_t.py), the array work in HIP kernels (csrc/mpgan_tiles.hip).

The reference cuts and augments every training tile with numpy / scipy on one host thread and hands the batch to
``sess.run`` through a feed (``selectRandomTiles`` :457-489, ``generateTile`` :491-546); once the training step
itself takes milliseconds that is the bottleneck.  Here

* ``addData`` also uploads the frames once ([frames, z, y, x, channels * dim_t] float32);
* ``selectRandomTilesDevice`` / ``selectRandomTempoTilesDevice`` return CUDA tensors.  Plain batches cost two
  launches (one gather per resolution) for the whole batch.  Augmented tiles run per sample: gather of the oversized
  crop -> resample (zoom) -> resample (rotation) -> final crop + quarter turns + flip in one launch;
* every draw from Python's ``random`` and ``numpy.random`` happens in the host order of the parent class.  The
  minimum-density test of a candidate crop (``getRandomTile``'s retry loop, :576-642) reads the host copy of the
  frame for the first crop -- identical decisions -- and, for the crop after a resampling, the density channel of
  the device result copied back (a few hundred floats); that one can differ from the host path only when a tile's
  density sits within float rounding of the threshold.

Plain batches are bit-equal to the host path; augmented ones agree to ~1e-6 (float64 coordinates as in
scipy.ndimage, float32 interpolation arithmetic).  Reference behaviours kept: see ``tilecreator_t`` (single-frame
tiles keep their vector components unrotated, :648-668).
"""
import ctypes
import random

import numpy as np
import torch

from . import _lib
from . import tilecreator_t as tc
from .ops import _ptr, _stream
from .tilecreator_t import DATA_KEY_HIGH, DATA_KEY_LOW, CUBE_ROTATIONS, TilecreatorError


def _ints(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def _doubles(vals):
    return (ctypes.c_double * len(vals))(*[float(v) for v in vals])


def _floats(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


# ------------------------------------------------------------------------------------------------
# thin wrappers of the C ABI
# ------------------------------------------------------------------------------------------------
def tile_gather(frames, table, tile_zyx, channels):
    """frames [F,Z,Y,X,Cf] (device), table [B,5] int32 rows (frame, first channel, z0, y0, x0) -> [B,tz,ty,tx,channels]"""
    lib = _lib.load()
    f, z, y, x, cf = frames.shape
    tbl = torch.as_tensor(np.ascontiguousarray(table, dtype=np.int32)).to(frames.device)
    b = tbl.shape[0]
    out = torch.empty((b,) + tuple(int(v) for v in tile_zyx) + (int(channels),), dtype=torch.float32, device=frames.device)
    _lib.check(lib.mpg_tile_gather(_stream(), _ptr(frames), f, z, y, x, cf, _ptr(tbl), b, int(tile_zyx[0]), int(tile_zyx[1]),
                                   int(tile_zyx[2]), int(channels), _ptr(out)), "mpg_tile_gather")
    return out


def resample_affine(src, out_zyx, matrix, offset, channel_mix=None):
    """scipy.ndimage.affine_transform(order=1, mode='constant') of every channel of src [z,y,x,c] (device)"""
    lib = _lib.load()
    zs, ys, xs, c = src.shape
    dst = torch.empty(tuple(int(v) for v in out_zyx) + (c,), dtype=torch.float32, device=src.device)
    mix = None if channel_mix is None else _floats(np.asarray(channel_mix, dtype=np.float32).reshape(-1))
    _lib.check(lib.mpg_resample_affine(_stream(), _ptr(src), zs, ys, xs, c, _ptr(dst), int(out_zyx[0]), int(out_zyx[1]),
                                       int(out_zyx[2]), _doubles(np.asarray(matrix, dtype=np.float64).reshape(-1)),
                                       _doubles(offset), mix), "mpg_resample_affine")
    return dst


def tile_orient(src, crop_off, crop_size, perm, flip, chan_map, chan_sign, out):
    lib = _lib.load()
    zs, ys, xs, c = src.shape
    _lib.check(lib.mpg_tile_orient(_stream(), _ptr(src), zs, ys, xs, c, _ints(crop_off), _ints(crop_size), _ints(perm), _ints(flip),
                                   _ints(chan_map), _floats(chan_sign), _ptr(out)), "mpg_tile_orient")
    return out


def semilagr_positions(vel, dt, n_out):
    """getSemiLagrPosBatch (:1345-1378), 2D: vel [B,h,w,3] device, dt [B] -> [B,n_out,n_out,2]"""
    lib = _lib.load()
    b, h, w, _ = vel.shape
    pos = torch.empty((b, n_out, n_out, 2), dtype=torch.float32, device=vel.device)
    _lib.check(lib.mpg_semilagr_positions(_stream(), _ptr(vel), _ptr(dt), b, h, w, int(n_out), _ptr(pos)), "mpg_semilagr_positions")
    return pos


# ------------------------------------------------------------------------------------------------
# orientation bookkeeping: a sequence of np.rot90 / np.flip as one signed axis permutation
# ------------------------------------------------------------------------------------------------
class _Orientation(object):
    """output axis k reads source axis perm[k], reversed if flip[k]; vector component (x,y,z) <-> grid axis (2,1,0)"""

    def __init__(self):
        self.perm, self.flip = [0, 1, 2], [0, 0, 0]
        # component of OUTPUT vector c (0 x, 1 y, 2 z) = sign * source component src
        self.comp_src, self.comp_sign = [0, 1, 2], [1.0, 1.0, 1.0]

    def flip_axis(self, axis, vectors=True):
        self.flip[axis] ^= 1
        if vectors:
            self.comp_sign[2 - axis] = -self.comp_sign[2 - axis]

    def quarter_turn(self, plane, vectors=True):
        """np.rot90(a, axes=(p, q)): out[.., i_p, .., i_q, ..] = a[.., i_q, .., n_q... ]: new axis p is old axis q reversed, new q is old p"""
        p, q = plane
        old_perm, old_flip = list(self.perm), list(self.flip)
        # rot90 with k=1 equals flip along q after swapping the two axes: out = swapaxes(flip(a, q), p, q)
        self.perm[p], self.flip[p] = old_perm[q], old_flip[q] ^ 1
        self.perm[q], self.flip[q] = old_perm[p], old_flip[p]
        if vectors:
            ca, cb = 2 - p, 2 - q           # tilecreator_t: ch[a], ch[b] = -ch[b], ch[a]
            sa, sb = self.comp_src[ca], self.comp_src[cb]
            ga, gb = self.comp_sign[ca], self.comp_sign[cb]
            self.comp_src[ca], self.comp_sign[ca] = sb, -gb
            self.comp_src[cb], self.comp_sign[cb] = sa, ga


class DeviceTileCreator(tc.TileCreator):
    def __init__(self, *args, device="cuda:0", **kw):
        super().__init__(*args, **kw)
        self.device = torch.device(device)
        self.dev = {DATA_KEY_LOW: None, DATA_KEY_HIGH: None}

    # ------------------------------------------------------------------ data
    def addData(self, low, high, flip_vel_z=True):
        super().addData(low, high, flip_vel_z)
        for key in (DATA_KEY_LOW, DATA_KEY_HIGH):
            host = np.ascontiguousarray(np.stack(self.data[key]), dtype=np.float32)
            self.dev[key] = torch.as_tensor(host).to(self.device)

    def clearData(self):
        super().clearData()
        self.dev = {DATA_KEY_LOW: None, DATA_KEY_HIGH: None}

    # ------------------------------------------------------------------ decisions (host, reference order)
    def _draw_frame(self, isTraining, tile_t):
        lo, hi = (0, self.setBorders[0]) if isTraining else (self.setBorders[0], self.setBorders[1])
        frame = random.randrange(lo, hi)
        first = 0
        if tile_t < self.dim_t:
            first = random.randrange(0, self.dim_t - tile_t)
        else:
            tile_t = self.dim_t
        return frame, first, tile_t

    def _draw_offset(self, density, low_shape, tileShapeLow, bounds):
        """getRandomTile's retry loop on a host density array [z,y,x] (channel 0 of the first packed frame)"""
        start, stop, size_low, size_high, mult = self._tile_geometry(low_shape, tileShapeLow, bounds)
        off = None
        for _ in range(19):
            off = np.asarray([random.randrange(int(start[a]), int(stop[a])) for a in range(3)])
            tile = density[off[0]:off[0] + size_low[0], off[1]:off[1] + size_low[1], off[2]:off[2] + size_low[2]]
            if tile.sum(dtype=np.float64) >= self.densityMinimum * size_low[0] * size_low[1] * size_low[2]:
                break
        return off, size_low, size_high, mult

    def _density(self, arr, first_channel, frames):
        """what hasMinDensity sums (:905-925): channel 0 of the tile when the layout has several channels, else the
        whole tile, i.e. every packed frame's single channel"""
        cl = int(self.tile_shape_low[-1])
        if cl > 1:
            return arr[..., first_channel]
        return arr[..., first_channel:first_channel + frames].astype(np.float64).sum(axis=-1)

    def _checks(self, isTraining, tile_t):
        have = self.setBorders[0] if isTraining else self.setBorders[1] - self.setBorders[0]
        if have < 1:
            self.TCError('no training data.' if isTraining else 'no test data.')
        if tile_t > self.dim_t:
            self.TCError('not enough coherent frames. Requested {}, given {}'.format(tile_t, self.dim_t))
        if self.dev[DATA_KEY_LOW] is None:
            self.TCError('no data on the device: call addData first')
        if self.premadeTiles or self.data_flags[DATA_KEY_HIGH]['isLabel']:
            self.TCError('premade tiles / label data are served by the host TileCreator')

    # ------------------------------------------------------------------ batches
    def selectRandomTilesDevice(self, selectionSize, isTraining=True, augment=False, tile_t=1):
        """selectRandomTiles (:457-489) returning CUDA tensors [n, z, y, x, channels * tile_t]"""
        self._checks(isTraining, tile_t)
        cl, ch = int(self.tile_shape_low[-1]), int(self.tile_shape_high[-1])
        if augment and self.useDataAug:
            lows = torch.empty((selectionSize,) + tuple(int(v) for v in self.tile_shape_low[:3]) + (cl * tile_t,),
                               dtype=torch.float32, device=self.device)
            highs = torch.empty((selectionSize,) + tuple(int(v) for v in self.tile_shape_high[:3]) + (ch * tile_t,),
                                dtype=torch.float32, device=self.device)
            for b in range(selectionSize):
                self._generate_tile_device(isTraining, tile_t, lows[b], highs[b])
            return lows, highs
        tl, th = np.zeros((selectionSize, 5), np.int32), np.zeros((selectionSize, 5), np.int32)
        for b in range(selectionSize):
            frame, first, tt = self._draw_frame(isTraining, tile_t)
            host = self.data[DATA_KEY_LOW][frame]
            off, size_low, size_high, mult = self._draw_offset(self._density(host, first * cl, tt), host.shape, None, [0, 0, 0, 0])
            tl[b] = (frame, first * cl, off[0], off[1], off[2])
            th[b] = (frame, first * ch, off[0] * mult[0], off[1] * mult[1], off[2] * mult[2])
        tt = min(tile_t, self.dim_t)
        low = tile_gather(self.dev[DATA_KEY_LOW], tl, self.tile_shape_low[:3], cl * tt)
        high = tile_gather(self.dev[DATA_KEY_HIGH], th, size_high[:3], ch * tt)
        return low, high

    def _generate_tile_device(self, isTraining, tile_t, out_low, out_high):
        """generateTile (:491-546) for one sample, written into out_low / out_high.  Draw order as in the parent."""
        cl, ch = int(self.tile_shape_low[-1]), int(self.tile_shape_high[-1])
        frame, first, tt = self._draw_frame(isTraining, tile_t)
        host_low = self.data[DATA_KEY_LOW][frame]
        dens = self._density(host_low, first * cl, tt)
        cur = {}                      # key -> device array [z,y,x,c]
        factor = None
        if self.do_scaling or self.do_rotation:
            grow = 1.5 if self.do_rotation else 1
            if self.do_scaling:
                factor = np.random.uniform(self.scaleFactor[0], self.scaleFactor[1])
                grow /= factor
            big = np.ceil(self.tile_shape_low * grow)
            if self.dim == 2:
                big[0] = 1
            off, size_low, size_high, mult = self._draw_offset(dens, host_low.shape, big.astype(int), [0, 0, 0, 0])
            cur[DATA_KEY_LOW] = tile_gather(self.dev[DATA_KEY_LOW], [(frame, first * cl, off[0], off[1], off[2])], size_low[:3], cl * tt)[0]
            cur[DATA_KEY_HIGH] = tile_gather(self.dev[DATA_KEY_HIGH], [(frame, first * ch, off[0] * mult[0], off[1] * mult[1], off[2] * mult[2])],
                                             size_high[:3], ch * tt)[0]
            host_density = None       # the density of `cur` now lives on the device
        else:
            host_density = dens
        if factor is not None:
            ref = np.array(cur[DATA_KEY_LOW].shape)
            want = [1 if self.dim == 2 else factor, factor, factor, 1]
            ratio = np.round(ref * want) / ref                     # scale(): both arrays use the LOW array's ratio (:818-826)
            for key, nch in ((DATA_KEY_LOW, cl), (DATA_KEY_HIGH, ch)):
                a = cur[key]
                out_shape = [int(round(a.shape[k] * ratio[k])) for k in range(3)]
                zoom = [(a.shape[k] - 1) / (out_shape[k] - 1) if out_shape[k] > 1 else 1.0 for k in range(3)]
                cur[key] = resample_affine(a, out_shape, np.diag(zoom), [0.0, 0.0, 0.0], self._vector_mix(key, nch, tt, None, factor))
        margin = np.zeros(4)
        if self.do_rotation:
            margin = np.array(tuple(cur[DATA_KEY_LOW].shape)) * 0.16
            rot = tc.draw_rotation(self.dim)
            for key, nch in ((DATA_KEY_LOW, cl), (DATA_KEY_HIGH, ch)):
                a = cur[key]
                centre = np.array(a.shape[:3]) / 2 - 0.5
                m = rot.T[:3, :3]
                offset = centre - m.dot(centre)
                cur[key] = resample_affine(a, a.shape[:3], m, offset, self._vector_mix(key, nch, tt, rot[:3, :3], None))
        if cur:
            host_density = self._density(cur[DATA_KEY_LOW].cpu().numpy(), 0, tt)
            low_shape = tuple(cur[DATA_KEY_LOW].shape)
        else:
            low_shape = host_low.shape
        off, size_low, size_high, mult = self._draw_offset(host_density, low_shape, None, margin)
        orient = _Orientation()
        turn_vectors = tt > 1         # single-frame tiles keep their components under quarter turns (tilecreator_t docstring)
        if self.do_rot90:
            for plane in CUBE_ROTATIONS[self.dim][np.random.choice(len(CUBE_ROTATIONS[self.dim]))]:
                orient.quarter_turn(plane, turn_vectors)
        if self.do_flip:
            axis = np.random.choice(4)
            if axis < 3:
                orient.flip_axis(int(axis))
        for key, nch, size, o, out in ((DATA_KEY_LOW, cl, size_low, off, out_low), (DATA_KEY_HIGH, ch, size_high, off * mult, out_high)):
            if cur:
                src = cur[key]
            else:       # no resampling happened: crop straight out of the frame store
                width = nch * tt
                src = self.dev[key][frame][..., (first * nch):(first * nch) + width].contiguous()
            cmap, csign = self._component_maps(key, nch, tt, orient)
            want = tuple(int(size[orient.perm[k]]) for k in range(3)) + (nch * tt,)
            if tuple(out.shape) != want:
                self.TCError('Wrong tile shape after data augmentation. is: {}. goal: {}.'.format(want, tuple(out.shape)))
            tile_orient(src, o[:3], size[:3], orient.perm, orient.flip, cmap, csign, out)

    def _vector_mix(self, key, nch, frames, rot3, factor):
        """channel mixing matrix of a resampling step: vector components scale with the grid (always) and rotate with it
        (packed coherent frames only), everything else passes through.  None when it would be the identity."""
        vectors = self.maps[key].vectors()
        if not vectors or (factor is None and (rot3 is None or frames <= 1)):
            return None
        c = nch * frames
        mix = np.eye(c, dtype=np.float64)
        for f in range(frames):
            for ix, iy, iz in vectors:
                idx = [f * nch + iz, f * nch + iy, f * nch + ix]             # (z, y, x) order of the rotation matrix
                block = np.eye(3)
                if rot3 is not None and frames > 1:
                    block = np.asarray(rot3, dtype=np.float64)
                if factor is not None:
                    block = block * factor
                for r in range(3):
                    for s in range(3):
                        mix[idx[r], idx[s]] = block[r, s]
        return mix

    def _component_maps(self, key, nch, frames, orient):
        c = nch * frames
        cmap, csign = list(range(c)), [1.0] * c
        for f in range(frames):
            for triple in self.maps[key].vectors():             # triple = [ix, iy, iz]
                for comp in range(3):
                    dst = f * nch + triple[comp]
                    cmap[dst] = f * nch + triple[orient.comp_src[comp]]
                    csign[dst] = orient.comp_sign[comp]
        return cmap, csign

    def selectRandomTempoTilesDevice(self, selectionSize, isTraining=True, augment=False, n_t=3, dt=0.25):
        """selectRandomTempoTiles (:1382-1412) on the device: flattened low tiles, high tiles and look-up positions"""
        samples = int(max(1, selectionSize // n_t))
        low, high = self.selectRandomTilesDevice(samples, isTraining, augment, tile_t=n_t)
        rows = samples * n_t
        tl, th = self.tileSizeLow, self.tileSizeHigh
        if self.dim != 2:
            raise NotImplementedError('3D look-up positions are undefined in the reference (tilecreator_t.py:1360)')

        def unpack(batch, t):
            b = batch.reshape(samples, int(t[0]), int(t[1]), int(t[2]), n_t, -1)
            return b.permute(0, 4, 1, 2, 3, 5).reshape(rows, int(t[0]), int(t[1]), int(t[2]), -1).contiguous()

        low = unpack(low, tl)
        vi = self.c_lists[DATA_KEY_LOW][tc.C_KEY_VELOCITY][0]
        vel = low[..., vi].reshape(rows, int(tl[1]), int(tl[2]), 3).contiguous()
        steps = torch.tensor([i * dt for i in range(n_t // 2, -n_t // 2, -1)] * samples, dtype=torch.float32, device=self.device)
        pos = semilagr_positions(vel, steps, int(th[1])).reshape(rows, -1)
        return low.reshape(rows, -1), unpack(high, th).reshape(rows, -1), pos
