"""Training-tile supply with the public interface of the reference's ``tools_wscale/tilecreator_t.py``
(class TileCreator :60, selectRandomTiles :457, generateTile :491, getRandomTile :576, augmentation
:648-879, getSemiLagrPosBatch :1345, selectRandomTempoTiles :1382).

What is kept is the CONTRACT: constructor arguments, method names, array shapes / dtypes (including
the places where the reference's arithmetic promotes to float64), the error class, and the order in
which Python's ``random`` (frame and offset choices; seeded 42 at import, :49) and ``numpy.random``
(augmentation parameters) are consumed -- so seeded runs give the reference's batches bit for bit
(tests/golden/tile_golden.npz was produced by the reference module itself).

The implementation is organised differently: a ``ChannelMap`` resolves a layout string into index
arrays once; every augmentation is a (parameters drawn) -> (applied with index arithmetic on whole
arrays) pair in ``_Augment``, so the sampling decisions and the array work are separate and the latter
can run on either numpy arrays (here) or device tensors (``tiles_device.py``).

Deliberate differences: ``rot=1`` draws the cube rotation by index (the reference's
``np.random.choice`` on a ragged list, :527, fails on numpy >= 1.24); PNG output uses Pillow
(``scipy.misc`` is gone); a 2D layout without a z velocity component is reported as a
TilecreatorError instead of the reference's bare ValueError.
"""
import itertools
import random
import re

import numpy as np
import scipy.ndimage

C_KEY_DEFAULT, C_KEY_VELOCITY, C_KEY_VORTICITY, C_KEY_POSITION = 'd', 'v', 'x', 'p'
C_KEY_OBSTACLE, C_KEY_FLAGS, C_KEY_K, C_KEY_EPS = 'o', 'f', 'k', 'e'
DATA_KEY_LOW, DATA_KEY_HIGH = 0, 1
AOPS_KEY_ROTATE, AOPS_KEY_SCALE, AOPS_KEY_ROT90, AOPS_KEY_FLIP = 'rot', 'scale', 'rot90', 'flip'

C_LAYOUT = {
    'dens': 'd',
    'dens_vel': 'd,vx,vy,vz',
    'dens_vel_obs': 'd,vx,vy,vz,o',
    'dens_vel_obs_flags': 'd,vx,vy,vz,o,f',
    'dens_vel_flags': 'd,vx,vy,vz,f',
}

random.seed(42)        # the reference seeds Python's generator when the module is imported (:49)

_SCALARS = (C_KEY_DEFAULT, C_KEY_OBSTACLE, C_KEY_FLAGS, C_KEY_K, C_KEY_EPS)
_VECTORS = (C_KEY_VELOCITY, C_KEY_VORTICITY)
_VEC_RE = re.compile(r'^([vx])(.*)([xyz])$')

# the 4 / 24 rotations of a square / cube as sequences of quarter turns in the planes (axis a -> axis b)
_Z, _NZ, _X, _Y, _NX, _NY = (2, 1), (1, 2), (1, 0), (0, 2), (0, 1), (2, 0)
CUBE_ROTATIONS = {
    2: ((), (_Z,), (_Z, _Z), (_NZ,)),
    3: ((), (_X,), (_Y,), (_X, _X), (_X, _Y), (_Y, _X), (_Y, _Y), (_NX,), (_X, _X, _Y), (_X, _Y, _X), (_X, _Y, _Y),
        (_Y, _X, _X), (_Y, _Y, _X), (_NY,), (_NX, _Y), (_X, _X, _Y, _X), (_X, _X, _Y, _Y), (_X, _Y, _X, _X), (_X, _NY),
        (_Y, _NX), (_NY, _X), (_NX, _Y, _X), (_X, _Y, _NX), (_X, _NY, _X)),
}


class TilecreatorError(Exception):
    """Tilecreator errors"""


# ------------------------------------------------------------------------------------------------
# channel layouts
# ------------------------------------------------------------------------------------------------
class ChannelMap(object):
    """'d,vx,vy,vz,o': scalar channels by kind and vector triples [ix, iy, iz] by kind (parseChannels, :937-1057)"""

    def __init__(self, text, dim):
        self.keys = [k.strip() for k in text.lower().split(',')]
        self.by_kind = {k: [] for k in _SCALARS + _VECTORS}
        for i, key in enumerate(self.keys):
            if not key:
                raise TilecreatorError('empty channel key.')
            if key[0] in _SCALARS:
                if len(key) != 1:
                    raise TilecreatorError('channel %d: unknown channel key "%s".' % (i, key))
                self.by_kind[key].append(i)
                continue
            m = _VEC_RE.match(key)
            if m is None:
                raise TilecreatorError('channel %d: unknown channel key "%s".' % (i, key))
            kind, label, axis = m.groups()
            comps = [kind + label + a for a in 'xyz']
            if any(self.keys.count(c) > 1 for c in comps):
                raise TilecreatorError('duplicate velocity channel with label "%s".' % label)
            needed = comps if dim == 3 else comps[:2]
            if any(c not in self.keys for c in needed):
                raise TilecreatorError('missing velocity channel with label "%s".' % label)
            if axis == 'x':
                if comps[2] not in self.keys:
                    raise TilecreatorError('vector "%s%s" has no z component (the reference requires one in 2D too).' % (kind, label))
                self.by_kind[kind].append([self.keys.index(c) for c in comps])

    @property
    def count(self):
        return len(self.keys)

    def vectors(self):
        """every [ix, iy, iz] triple that follows the grid transforms (velocities and vorticities)"""
        return [t for kind in _VECTORS for t in self.by_kind[kind]]


def _as3(value, dim, what):
    if np.isscalar(value):
        return np.asarray([value] * 3)
    value = list(value)
    if len(value) == 2 and dim == 2:
        return np.asarray([1] + value)
    if len(value) == 3:
        return np.asarray(value)
    raise TilecreatorError('%s mismatch.' % what)


# ------------------------------------------------------------------------------------------------
# augmentation: array work on a {DATA_KEY_LOW: array, DATA_KEY_HIGH: array} pair
# ------------------------------------------------------------------------------------------------
class _Augment(object):
    """Applies drawn augmentation parameters.  `maps` = {key: ChannelMap}, `labels` = keys whose array is a
    label (left untouched).  Vector components are addressed by index arrays over all coherent frames packed
    in the channel axis (frame f, channel c -> f * channels + c)."""

    order, fill = 1, 'constant'

    def __init__(self, maps, labels):
        self.maps, self.labels = maps, labels

    def _components(self, key, n_channels):
        """index arrays (ix, iy, iz), each covering every vector of every packed frame"""
        m = self.maps[key]
        frames = n_channels // m.count
        trip = [[f * m.count + t[a] for f in range(frames) for t in m.vectors()] for a in range(3)]
        return [np.asarray(t, dtype=np.intp) for t in trip]

    def _each(self, pair):
        return [k for k in pair if k not in self.labels]

    def _turning_components(self, key, n_channels):
        """Components that rotations act on.  Reference behaviour, kept: its rotation helpers build NEW channel
        arrays, and the caller stores them back only for packed coherent frames (special_aug, :648-668) -- so a
        single-frame tile (tile_t = 1) keeps its vector components unrotated under `rotate` and `rotate90`, while
        flips and scalings (done in place there) always apply."""
        comp = self._components(key, n_channels)
        if n_channels // self.maps[key].count <= 1:
            return [c[:0] for c in comp]
        return comp

    def flip(self, pair, axis):
        """mirror along grid axis (0 z, 1 y, 2 x); the matching vector component changes sign"""
        for k in self._each(pair):
            a = np.flip(pair[k], axis)
            comp = self._components(k, a.shape[-1])[2 - axis]
            if comp.size:
                a = a.copy()
                a[..., comp] *= -1
            pair[k] = a
        return pair

    def quarter_turn(self, pair, plane):
        """np.rot90 in the plane (a, b) of grid axes; vector components (x,y,z <-> axes 2,1,0) turn with it"""
        for k in self._each(pair):
            a = np.rot90(pair[k], axes=plane)
            comp = self._turning_components(k, a.shape[-1])
            ca, cb = comp[2 - plane[0]], comp[2 - plane[1]]
            if ca.size:
                a = a.copy()
                old_a = a[..., ca].copy()
                a[..., ca] = -a[..., cb]
                a[..., cb] = old_a
            pair[k] = a
        return pair

    def resample(self, pair, factor, dim):
        """zoom to round(factor * resolution) of the LOW array (both arrays use that ratio, :808-845);
        vectors scale with the grid"""
        ref = np.array(pair[DATA_KEY_LOW].shape)
        want = [1 if dim == 2 else factor, factor, factor, 1]
        ratio = np.round(ref * want) / ref
        for k in self._each(pair):
            a = scipy.ndimage.zoom(pair[k], ratio, order=self.order, mode=self.fill, cval=0.0)
            comp = np.concatenate(self._components(k, a.shape[-1]))
            if comp.size:
                a[..., comp] *= factor
            pair[k] = a
        return pair

    def rotate(self, pair, matrix):
        """matrix: 4x4 homogeneous rotation in (z,y,x) order.  Vectors are rotated (which promotes the array to
        float64, as in the reference), then every channel is resampled about the array centre."""
        rot3 = matrix[:3, :3]
        for k in self._each(pair):
            a = pair[k]
            ix, iy, iz = self._turning_components(k, a.shape[-1])
            if ix.size:
                zyx = np.stack([a[..., iz], a[..., iy], a[..., ix]], axis=0).astype(np.float64)     # [3, ..., nvec]
                turned = np.tensordot(rot3, zyx, axes=([1], [0]))
                a = a.astype(np.float64)
                a[..., iz], a[..., iy], a[..., ix] = turned[0], turned[1], turned[2]
            pair[k] = self.about_centre(a, matrix.T)
        return pair

    def about_centre(self, a, matrix, data_dim=3):
        if a.ndim != 4:
            raise TilecreatorError('Data shape mismatch.')
        centre = np.array(a.shape) / 2 - np.array([0.5, 0.5, 0.5, 0])
        shift_in, shift_out = np.eye(4), np.eye(4)
        shift_in[:3, 3] = centre[:3]
        shift_out[:3, 3] = -centre[:3]
        m = shift_in.dot(matrix).dot(shift_out)
        planes = [scipy.ndimage.affine_transform(a[..., c], m[:data_dim, :data_dim], m[:data_dim, data_dim],
                                                 order=self.order, mode=self.fill, cval=0.)
                  for c in range(a.shape[-1])]
        return np.stack(planes, axis=-1)


def draw_rotation(dim):
    """random rotation as a homogeneous 4x4 in (z,y,x) order: an angle in 2D, a unit quaternion in 3D (:677-698)"""
    m = np.eye(4)
    if dim == 2:
        theta = np.pi * np.random.uniform(0, 2)
        c, s = np.cos(theta), np.sin(theta)
        m[1:3, 1:3] = [[c, -s], [s, c]]
        return m
    q = np.random.normal(size=4)
    q /= np.linalg.norm(q)
    p = np.outer(q, q) * 2
    m[:3, :3] = [[1 - p[2, 2] - p[3, 3], p[1, 2] - p[3, 0], p[1, 3] + p[2, 0]],
                 [p[1, 2] + p[3, 0], 1 - p[1, 1] - p[3, 3], p[2, 3] - p[1, 0]],
                 [p[1, 3] - p[2, 0], p[2, 3] + p[1, 0], 1 - p[1, 1] - p[2, 2]]]
    return m


# ------------------------------------------------------------------------------------------------
# the tile creator
# ------------------------------------------------------------------------------------------------
class TileCreator(object):

    def __init__(self, tileSizeLow, simSizeLow=64, upres=2, dim=2, dim_t=1, overlapping=0, densityMinimum=0.02,
                 premadeTiles=False, partTrain=0.9, partTest=0.1, partVal=0, channelLayout_low=C_LAYOUT['dens_vel'],
                 channelLayout_high=C_LAYOUT['dens'], highIsLabel=False, loadPN=False, padding=0):
        if dim not in (2, 3):
            self.TCError('Data dimension must be 2 or 3.')
        self.dim, self.dim_t = dim, dim_t
        self.tileSizeLow = _as3(tileSizeLow, dim, 'Tile size')
        self.simSizeLow = _as3(simSizeLow, dim, 'Simulation size')
        if np.isscalar(upres):
            self.upres = upres
            smallest = upres
        else:
            self.upres = [1] + list(upres) if (dim == 2 and len(upres) == 2) else list(upres)
            smallest = upres[1]
        if smallest < 1:
            self.TCError('Upres must be at least 1.')
        self.tileSizeHigh = self.tileSizeLow * self.upres
        self.simSizeHigh = self.simSizeLow * self.upres
        if dim == 2:
            for a in (self.tileSizeLow, self.tileSizeHigh, self.simSizeLow, self.simSizeHigh):
                a[0] = 1
        if (self.simSizeLow < self.tileSizeLow).any():
            self.TCError('Tile size {} can not be larger than sim size {}.'.format(self.tileSizeLow, self.simSizeLow))
        if densityMinimum < 0.:
            self.TCError('densityMinimum can not be negative.')
        if loadPN:
            self.TCError('prev and next tiles not supported.')
        self.densityMinimum, self.premadeTiles, self.useDataAug = densityMinimum, premadeTiles, False
        self.hasPN, self.padding = loadPN, padding

        self.maps = {DATA_KEY_LOW: ChannelMap(channelLayout_low, dim), DATA_KEY_HIGH: ChannelMap(channelLayout_high, dim)}
        self.c_low, self.c_high = self.maps[DATA_KEY_LOW].keys, self.maps[DATA_KEY_HIGH].keys
        self.c_lists = {k: m.by_kind for k, m in self.maps.items()}
        self.data_flags = {}
        for key, is_label in ((DATA_KEY_LOW, False), (DATA_KEY_HIGH, highIsLabel)):
            flags = {'isLabel': is_label, 'channels': self.maps[key].count, C_KEY_POSITION: False}
            for kind in _VECTORS + (C_KEY_OBSTACLE, C_KEY_FLAGS, C_KEY_K, C_KEY_EPS):
                flags[kind] = len(self.c_lists[key][kind]) > 0
            self.data_flags[key] = flags
        print('TileCreator: dim {}, dim_t {}; low {} (vectors {}), high {} (vectors {})'.format(
            dim, dim_t, self.c_low, self.maps[DATA_KEY_LOW].vectors(), self.c_high, self.maps[DATA_KEY_HIGH].vectors()))

        n_low, n_high = self.maps[DATA_KEY_LOW].count, self.maps[DATA_KEY_HIGH].count
        self.tile_shape_low = np.append(self.tileSizeLow, [n_low])
        self.tile_shape_high = np.append(self.tileSizeHigh, [n_high])
        self.frame_shape_low = np.append(self.simSizeLow, [n_low])
        self.frame_shape_high = np.append(self.simSizeHigh, [n_high])
        self.densityThreshold = self.densityMinimum * np.prod(self.tile_shape_low[:3])
        self.data = {DATA_KEY_LOW: [], DATA_KEY_HIGH: []}
        whole = partTrain + partTest + partVal
        self.part_train, self.part_test, self.part_validation = partTrain / whole, partTest / whole, partVal / whole
        self._aug = _Augment(self.maps, [k for k, f in self.data_flags.items() if f['isLabel']])

    def TCError(self, msg):
        raise TilecreatorError(msg)

    def parseChannels(self, channelString):
        m = ChannelMap(channelString, self.dim)
        return m.keys, m.by_kind

    # ------------------------------------------------------------------ augmentation switches (:227-317)
    def initDataAugmentation(self, rot=2, minScale=0.85, maxScale=1.15, flip=True):
        self.useDataAug = True
        self.do_rotation, self.do_rot90 = rot == 2, rot == 1
        self.cube_rot = {d: [list(seq) for seq in CUBE_ROTATIONS[d]] for d in CUBE_ROTATIONS}
        self.scaleFactor = [minScale, maxScale]
        self.do_scaling = not (minScale == 1 and maxScale == 1)
        self.do_flip = flip
        self.interpolation_order, self.fill_mode = _Augment.order, _Augment.fill
        active = [n for n, on in (('rotation', self.do_rotation), ('rot90', self.do_rot90), ('scaling', self.do_scaling),
                                  ('flip', self.do_flip)) if on]
        print('data augmentation: ' + ', '.join(active))

    # ------------------------------------------------------------------ frames (:321-392)
    def addData(self, low, high, flip_vel_z=True):
        low, high = np.asarray(low), np.asarray(high)
        if low.ndim != high.ndim:
            self.TCError('Data shape mismatch. Dimensions: %d vs %d' % (low.ndim, high.ndim))
        if low.ndim not in (4, 5):
            self.TCError('Input must be single 3D data or sequence of 3D data.')
        for arr, key, name in ((low, DATA_KEY_LOW, 'low'), (high, DATA_KEY_HIGH, 'high')):
            if arr.shape[-1] != self.dim_t * self.data_flags[key]['channels']:
                self.TCError('(Dim_t * Channels) configured for tilecreator ({}-res) do not match the channels of the data: {}'.format(
                    name, [arr.shape[-1], self.dim_t, self.data_flags[key]['channels']]))
        if low.ndim == 5:
            if low.shape[0] != high.shape[0]:
                self.TCError('unequal amount of low ({}) and high ({}) data.'.format(low.shape[0], high.shape[0]))
        else:
            low, high = low[None], high[None]
        if flip_vel_z and self.dim == 3:
            # sign convention of the siggraph-2018 3D data: channel 3 (vz) of every packed frame is negated; the
            # multiplication by a float64 vector also makes the stored low frames float64, as in the reference (:353-358)
            sign = np.ones(self.data_flags[DATA_KEY_LOW]['channels'] * self.dim_t)
            sign[3::self.data_flags[DATA_KEY_LOW]['channels']] = -1.0
            low = low * sign.reshape((1, 1, 1, 1, -1))
            print('Note - flipped Z coord of velocities! Only for sigg18-3d-data! disable for other data...')
        shapes = (list(low.shape[1:]), list(high.shape[1:]))
        if self.premadeTiles:
            if self.dim_t != 1:
                self.TCError('Currently, Dim_t = {} > 1 is not supported by premade tiles'.format(self.dim_t))
            want = (list(self.tile_shape_low), list(self.tile_shape_high))
            what = 'Tile'
        else:
            shapes[0][-1] //= self.dim_t
            shapes[1][-1] //= self.dim_t
            want = (list(self.frame_shape_low), list(self.frame_shape_high))
            what = 'Frame'
        if shapes[0] != want[0] or shapes[1] != want[1]:
            self.TCError('{} shape mismatch: is - specified\n\tlow: {} - {}\n\thigh {} - {} (dim_t {})'.format(
                what, shapes[0], want[0], shapes[1], want[1], self.dim_t))
        self.data[DATA_KEY_LOW].extend(low)
        self.data[DATA_KEY_HIGH].extend(high)
        self.splitSets()

    def splitSets(self):
        n = len(self.data[DATA_KEY_LOW])
        train = int(n * self.part_train)
        test = train + int(n * self.part_test)
        self.setBorders = [train, test, n]
        print('frames: {} training, {} testing, {} validation'.format(train, test - train, n - test))

    def clearData(self):
        self.data = {DATA_KEY_LOW: [], DATA_KEY_HIGH: []}

    # ------------------------------------------------------------------ regular tilings (:403-450, 886-931)
    def createTiles(self, data, tileShape, strides=-1):
        """all tiles of a regular grid with the given strides (default: the tile size), z-major order"""
        step = list(tileShape[:3]) if (np.isscalar(strides) and strides <= 0) else \
            ([strides] * 3 if np.isscalar(strides) else list(strides))
        pad = [self.padding] * 3 + [0]
        if data.shape[0] <= 1:
            pad[0], step[0] = 0, 1
        counts = [(data.shape[i] - tileShape[i]) // step[i] + 1 for i in range(3)]
        tiles = []
        for tz, ty, tx in itertools.product(*[range(c) for c in counts]):
            z, y, x = tz * step[0], ty * step[1], tx * step[2]
            t = data[z:z + tileShape[0], y:y + tileShape[1], x:x + tileShape[2], :]
            tiles.append(np.pad(t, [(p, p) for p in pad], 'edge') if self.padding > 0 else t)
        return np.array(tiles)

    def cutTile(self, data, tileShape, offset=[0, 0, 0]):
        size = [int(v) for v in np.asarray(tileShape)[:3]]
        o = [int(v) for v in np.asarray(offset)[:3]]
        if any(data.shape[i] < size[i] + o[i] for i in range(3)):
            self.TCError('Can\'t cut tile with shape {} and offset {} from data with shape {}.'.format(size, o, data.shape))
        tile = data[o[0]:o[0] + size[0], o[1]:o[1] + size[1], o[2]:o[2] + size[2], :]
        if list(tile.shape[:3]) != size:
            self.TCError('Wrong tile shape after cutting. is: {}. goal: {}.'.format(tile.shape, size))
        return tile

    def concatTiles(self, tiles, frameShape, tileBorder=[0, 0, 0, 0]):
        """inverse of createTiles for a [nz, ny, nx] grid of tiles, optionally dropping a border of every tile"""
        if tiles.ndim != 5 or len(frameShape) != 3 or len(tileBorder) != 4:
            self.TCError('Data shape mismatch.')
        nz, ny, nx = [int(v) for v in frameShape]
        if nz * ny * nx != len(tiles):
            self.TCError('given tiles do not match required tiles.')
        border = np.asarray(tileBorder)
        if (border > 0).any():
            inner = tiles.shape[1:] - 2 * border
            tiles = np.asarray([self.cutTile(t, inner, border) for t in tiles])
        grid = tiles.reshape((nz, ny, nx) + tiles.shape[1:])
        # [nz, ny, nx, tz, ty, tx, c] -> [nz, tz, ny, ty, nx, tx, c]
        grid = grid.transpose(0, 3, 1, 4, 2, 5, 6)
        s = grid.shape
        return grid.reshape(s[0] * s[1], s[2] * s[3], s[4] * s[5], s[6])

    def getFrameTiles(self, index):
        low, high = self.getDatum(index)
        return self.createTiles(low, self.tile_shape_low), self.createTiles(high, self.tile_shape_high)

    # ------------------------------------------------------------------ density test (:905-925)
    def getTileDensity(self, tile):
        dens = tile[..., :1] if self.data_flags[DATA_KEY_LOW]['channels'] > 1 else tile
        return dens.sum(dtype=np.float64)

    def hasMinDensity(self, tile):
        return self.getTileDensity(tile) >= self.densityMinimum * tile.shape[0] * tile.shape[1] * tile.shape[2]

    # ------------------------------------------------------------------ random access (:457-642)
    def getDatum(self, index, tile_t=1):
        """copies of frame index // dim_t, restricted to tile_t coherent sub-frames from index % dim_t"""
        frame, first = divmod(index, self.dim_t) if self.dim_t > 1 else (index, 0)
        out = []
        for key, width in ((DATA_KEY_LOW, self.tile_shape_low[-1]), (DATA_KEY_HIGH, self.tile_shape_high[-1])):
            out.append(np.copy(self.data[key][frame][..., first * width:(first + tile_t) * width]))
        return out[0], out[1]

    def getRandomDatum(self, isTraining=True, tile_t=1):
        lo, hi = (0, self.setBorders[0]) if isTraining else (self.setBorders[0], self.setBorders[1])
        frame = random.randrange(lo, hi)
        first = 0
        if tile_t < self.dim_t:
            first = random.randrange(0, self.dim_t - tile_t)
        else:
            tile_t = self.dim_t
        return self.getDatum(frame * self.dim_t + first, tile_t)

    def _tile_geometry(self, low_shape, tileShapeLow, bounds):
        """(start, stop) of the low-res offset range per axis, high-res tile shape, offset multipliers"""
        size_low = np.copy(self.tile_shape_low) if tileShapeLow is None else np.asarray(tileShapeLow)
        if np.isscalar(self.upres):
            size_high = size_low * self.upres
            mult = np.array([self.upres] * 3)
        else:
            # non-scalar upres: the reference hard-codes the 8x-along-x case here (:594, :610)
            size_high = size_low * np.array((1, 1, 8, 1))
            mult = np.array([1, 1, 4])
        start = np.floor(bounds)
        stop = np.asarray(low_shape) - size_low + 1 - start
        if self.dim == 2:
            start[0], stop[0], mult[0], size_high[0] = 0, 1, 1, 1
        if (stop - start)[:3].min() < 0:
            self.TCError('Can\'t cut tile {} from frame {} with bounds {}.'.format(size_low, low_shape, start))
        return start.astype(int), stop.astype(int), size_low, size_high, mult

    def getRandomTile(self, low, high, tileShapeLow=None, bounds=[0, 0, 0, 0]):
        """a random low/high tile pair with at least the minimum mean density (up to 19 draws); `bounds` keeps the
        offsets away from the frame border (mirrored parts after a rotation)"""
        if low.ndim != 4 or high.ndim != 4 or (tileShapeLow is not None and len(tileShapeLow) != 4):
            self.TCError('Data shape mismatch.')
        start, stop, size_low, size_high, mult = self._tile_geometry(low.shape, tileShapeLow, bounds)
        is_label = self.data_flags[DATA_KEY_HIGH]['isLabel']
        low_tile = high_tile = None
        for _ in range(19):
            off = np.asarray([random.randrange(int(start[a]), int(stop[a])) for a in range(3)])
            low_tile = self.cutTile(low, size_low, off)
            high_tile = high if is_label else self.cutTile(high, size_high, off * mult)
            if self.hasMinDensity(low_tile):
                break
        return low_tile, high_tile

    def selectRandomTiles(self, selectionSize, isTraining=True, augment=False, tile_t=1):
        """-> low [selectionSize, z, y, x, channels * tile_t], high likewise; z = 1 in 2D"""
        have = self.setBorders[0] if isTraining else self.setBorders[1] - self.setBorders[0]
        if have < 1:
            self.TCError('no training data.' if isTraining else 'no test data.')
        if tile_t > self.dim_t:
            self.TCError('not enough coherent frames. Requested {}, given {}'.format(tile_t, self.dim_t))
        lows, highs = [], []
        for _ in range(selectionSize):
            if augment and self.useDataAug:
                low, high = self.generateTile(isTraining, tile_t)
            else:
                low, high = self.getRandomDatum(isTraining, tile_t)
                if not self.premadeTiles:
                    low, high = self.getRandomTile(low, high)
            lows.append(low)
            highs.append(high)
        return np.asarray(lows), np.asarray(highs)

    def generateTile(self, isTraining=True, tile_t=1):
        """one augmented pair.  Draw order (part of the contract): frame [, sub-frame], scale factor, offsets of the
        oversized crop, rotation, offsets of the final crop, quarter-turn sequence, flip axis."""
        low, high = self.getRandomDatum(isTraining, tile_t)
        pair = {DATA_KEY_LOW: low, DATA_KEY_HIGH: high}
        if not self.premadeTiles:
            factor = None
            if self.do_scaling or self.do_rotation:
                grow = 1.5 if self.do_rotation else 1
                if self.do_scaling:
                    factor = np.random.uniform(self.scaleFactor[0], self.scaleFactor[1])
                    grow /= factor
                big = np.ceil(self.tile_shape_low * grow)
                if self.dim == 2:
                    big[0] = 1
                pair[DATA_KEY_LOW], pair[DATA_KEY_HIGH] = self.getRandomTile(low, high, big.astype(int))
            if factor is not None:
                pair = self._aug.resample(pair, factor, self.dim)
            margin = np.zeros(4)
            if self.do_rotation:
                margin = np.array(pair[DATA_KEY_LOW].shape) * 0.16
                pair = self._aug.rotate(pair, draw_rotation(self.dim))
            pair[DATA_KEY_LOW], pair[DATA_KEY_HIGH] = self.getRandomTile(pair[DATA_KEY_LOW], pair[DATA_KEY_HIGH], bounds=margin)
        if self.do_rot90:
            for plane in CUBE_ROTATIONS[self.dim][np.random.choice(len(CUBE_ROTATIONS[self.dim]))]:
                pair = self._aug.quarter_turn(pair, plane)
        if self.do_flip:
            axis = np.random.choice(4)
            if axis < 3:
                pair = self._aug.flip(pair, axis)
        for key, goal in ((DATA_KEY_LOW, self.tile_shape_low), (DATA_KEY_HIGH, self.tile_shape_high)):
            want = list(goal[:3]) + [goal[3] * tile_t]
            if list(pair[key].shape) != want:
                self.TCError('Wrong tile shape after data augmentation. is: {}. goal: {}.'.format(pair[key].shape, want))
        return pair[DATA_KEY_LOW], pair[DATA_KEY_HIGH]

    # kept for callers of the reference's augmentation entry points
    def flip(self, data, axes, isFrame=True):
        for axis in (axes if isFrame else [a + 1 for a in axes]):
            data = self._aug.flip(data, int(axis))
        return data

    def rotate90(self, data, axes):
        if len(axes) != 2:
            self.TCError('need 2 axes for rotate90.')
        return self._aug.quarter_turn(data, tuple(axes))

    def scale(self, data, factor):
        return self._aug.resample(data, factor, self.dim)

    def rotate(self, data):
        return self._aug.rotate(data, draw_rotation(self.dim))

    def applyTransform(self, data, transform_matrix, data_dim=3):
        return self._aug.about_centre(data, transform_matrix, data_dim)

    # ------------------------------------------------------------------ coherent batches (:1382-1412)
    def selectRandomTempoTiles(self, selectionSize, isTraining=True, augment=False, n_t=3, dt=0.25):
        """n_t coherent frames per sample: flattened low tiles [n_t * (selectionSize // n_t), -1], high tiles and the
        semi-Lagrangian look-up positions of every frame (dt * (+1, 0, -1) for n_t = 3)"""
        samples = int(max(1, selectionSize // n_t))
        low, high = self.selectRandomTiles(samples, isTraining, augment, tile_t=n_t)
        rows = samples * n_t
        tl, th = self.tileSizeLow, self.tileSizeHigh

        def unpack(batch, t):   # [sample, z, y, x, frame * c] -> [sample * frame, z, y, x, c]
            b = batch.reshape((samples, t[0], t[1], t[2], n_t, -1))
            return np.moveaxis(b, 4, 1).reshape((rows, t[0], t[1], t[2], -1))

        low = unpack(low, tl)
        vel = low[..., self.c_lists[DATA_KEY_LOW][C_KEY_VELOCITY][0]].reshape((rows, tl[0], tl[1], tl[2], 3))
        steps = np.array([i * dt for i in range(n_t // 2, -n_t // 2, -1)] * samples, dtype=np.float32)
        steps = steps.reshape((-1,) + (1,) * (3 if self.dim == 2 else 4))
        pos = getSemiLagrPosBatch(vel, steps, th[1]).reshape((rows, -1))
        return low.reshape((rows, -1)), unpack(high, th).reshape((rows, -1)), pos


# ------------------------------------------------------------------------------------------------
# advection look-up positions for the temporal discriminator (:1293-1378)
# ------------------------------------------------------------------------------------------------
def gridInterpolBatch(macgridbatch, targetshape, order=1):
    """resample [b,z,y,x,c] to targetshape on cell-centred coordinates; batch and channel axes are not mixed"""
    src = macgridbatch.shape
    if len(targetshape) != 5 or len(src) != 5 or targetshape[-1] != src[-1]:
        raise TilecreatorError('gridInterpolBatch: shapes %s -> %s' % (src, targetshape))
    axes = [np.linspace(0, targetshape[0] - 1, targetshape[0])]
    axes += [np.linspace(0.5, targetshape[k] - 0.5, targetshape[k]) * (float(src[k]) / targetshape[k]) for k in (1, 2, 3)]
    axes.append(np.linspace(0, targetshape[4] - 1, targetshape[4]))
    return scipy.ndimage.map_coordinates(macgridbatch, np.meshgrid(*axes, indexing='ij'), order=order, mode='nearest')


def getMACGridCenteredBatch(macgrid_batch, is3D):
    """staggered (MAC) velocities -> cell centres, components returned in grid order (z,y,x; 2D: y,x)"""
    b, nz, ny, nx, _ = macgrid_batch.shape

    def next_along(axis, comp, n):      # component `comp` of the neighbour one cell up `axis` (clamped at the end)
        idx = list(range(1, n)) + [n - 1]
        return macgrid_batch.take(idx, axis=axis)[..., comp].reshape([b, nz, ny, nx, 1])

    if is3D:
        up = np.concatenate((next_along(1, 2, nz), next_along(2, 1, ny), next_along(3, 0, nx)), axis=-1)
        return (0.5 * (macgrid_batch[..., ::-1] + up)).reshape([b, nz, ny, nx, 3])
    up = np.concatenate((next_along(2, 1, ny), next_along(3, 0, nx)), axis=4)
    return (0.5 * (macgrid_batch[..., -2::-1] + up)).reshape([b, ny, nx, 2])


def getSemiLagrPosBatch(macgrid_batch, dt, cube_len_output=-1):
    """positions x - v(x) * dt on a cube_len_output^2 grid, [b,y,x,2] (2D only: the reference's 3D branch reads an
    undefined name, :1360)"""
    if macgrid_batch.ndim != 5 or macgrid_batch.shape[-1] != 3:
        raise TilecreatorError('getSemiLagrPosBatch expects [b,z,y,x,3]')
    b, nz, ny, nx, _ = macgrid_batch.shape
    if nz > 1:
        raise NotImplementedError('3D look-up positions are undefined in the reference (tilecreator_t.py:1360)')
    if cube_len_output == -1:
        cube_len_output = nx
    fx, fy = float(nx) / cube_len_output, float(ny) / cube_len_output
    ox, oy = int(nx / fx + 0.5), int(ny / fy + 0.5)
    yy, xx = np.meshgrid(np.linspace(0.5, oy - 0.5, oy), np.linspace(0.5, ox - 0.5, ox), indexing='ij')
    cells = np.stack((yy, xx), axis=-1).reshape([1, oy, ox, 2])
    if cube_len_output == nx:
        return cells - getMACGridCenteredBatch(macgrid_batch, False) * dt
    fine = gridInterpolBatch(macgrid_batch, [b, 1, oy, ox, 3], 1)
    return cells - (getMACGridCenteredBatch(fine, False) / fx) * dt


# ------------------------------------------------------------------------------------------------
# previews (Pillow instead of scipy.misc.toimage, :1125-1157)
# ------------------------------------------------------------------------------------------------
def savePngsGrayscale(tiles, path, imageCounter=0, tiles_in_image=[1, 1], channels=[0], save_gif=False,
                      plot_vel_x_y=False):
    """tiles [tile,y,x,c] -> mosaics of tiles_in_image (rows, cols); one 8-bit PNG per mosaic and channel"""
    from PIL import Image
    rows, cols = tiles_in_image
    tiles = np.asarray(tiles)
    if len(tiles) % (rows * cols) != 0:
        print('ERROR: number of tiles does not match tiles per image')
        return
    count = len(tiles) // (rows * cols)
    mosaics = tiles.reshape((count, rows, cols) + tiles.shape[1:])
    mosaics = mosaics.transpose(0, 1, 3, 2, 4, 5).reshape((count, rows * tiles.shape[1], cols * tiles.shape[2], -1))
    for m, img in enumerate(mosaics):
        for c in channels:
            number = imageCounter * count + m
            name = 'img_{:04d}.png'.format(number) if len(channels) == 1 else 'img_{:04d}_c{:04d}.png'.format(number, c)
            Image.fromarray((np.clip(img[..., c], 0.0, 1.0) * 255).astype(np.uint8)).save(path + name)
