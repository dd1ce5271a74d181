"""In-memory low/high frame pairs -> training tiles, with the interface of the reference's
``tools_wscale/tilecreator_t.py`` (TileCreator :60, selectRandomTiles :457, generateTile :491,
getRandomTile :576, augmentation :648-879, selectRandomTempoTiles :1382, getSemiLagrPosBatch :1345).

Same constructor arguments, method names, return shapes / dtypes, error class, and the same
sequence of random draws (Python ``random`` seeded 42 at import for frame / offset choices,
``numpy.random`` for the augmentation parameters), so seeded runs reproduce the reference's batches
(pinned by tests/golden fixtures generated with the reference module).

Deliberate differences: ``rot=1`` (90-degree rotations) draws the cube rotation by index, which is
what the reference's ``np.random.choice`` on a ragged list did before numpy 1.24 broke it
(tilecreator_t.py:527); PNG helpers use Pillow instead of the removed ``scipy.misc``.
"""
from random import randrange, seed

import numpy as np
import scipy.ndimage

C_KEY_DEFAULT = 'd'
C_KEY_VELOCITY = 'v'
C_KEY_VORTICITY = 'x'
C_KEY_POSITION = 'p'
C_KEY_OBSTACLE = 'o'
C_KEY_FLAGS = 'f'
C_KEY_K = 'k'
C_KEY_EPS = 'e'

DATA_KEY_LOW = 0
DATA_KEY_HIGH = 1

AOPS_KEY_ROTATE = 'rot'
AOPS_KEY_SCALE = 'scale'
AOPS_KEY_ROT90 = 'rot90'
AOPS_KEY_FLIP = 'flip'

seed(42)        # tilecreator_t.py:49

C_LAYOUT = {
    'dens': C_KEY_DEFAULT,
    'dens_vel': 'd,vx,vy,vz',
    'dens_vel_obs': 'd,vx,vy,vz,o',
    'dens_vel_obs_flags': 'd,vx,vy,vz,o,f',
    'dens_vel_flags': 'd,vx,vy,vz,f',
}

_SCALAR_KEYS = (C_KEY_DEFAULT, C_KEY_OBSTACLE, C_KEY_FLAGS, C_KEY_K, C_KEY_EPS)
_VECTOR_KEYS = (C_KEY_VELOCITY, C_KEY_VORTICITY)


class TilecreatorError(Exception):
    ''' Tilecreator errors '''


def _size3(value, dim, what):
    if np.isscalar(value):
        return np.asarray([value, value, value])
    if len(value) == 2 and dim == 2:
        return np.asarray([1] + list(value))
    if len(value) == 3:
        return np.asarray(value)
    raise TilecreatorError('%s mismatch.' % what)


class TileCreator(object):

    def __init__(self, tileSizeLow, simSizeLow=64, upres=2, dim=2, dim_t=1, overlapping=0, densityMinimum=0.02,
                 premadeTiles=False, partTrain=0.9, partTest=0.1, partVal=0, channelLayout_low=C_LAYOUT['dens_vel'],
                 channelLayout_high=C_LAYOUT['dens'], highIsLabel=False, loadPN=False, padding=0):
        self.dim_t = dim_t
        if dim != 2 and dim != 3:
            self.TCError('Data dimension must be 2 or 3.')
        self.dim = dim
        self.tileSizeLow = _size3(tileSizeLow, dim, 'Tile size')
        self.simSizeLow = _size3(simSizeLow, dim, 'Simulation size')
        if np.isscalar(upres):
            self.upres = upres
            if upres < 1:
                self.TCError('Upres must be at least 1.')
        else:
            self.upres = [1] + list(upres) if (dim == 2 and len(upres) == 2) else list(upres)
            if upres[1] < 1:
                self.TCError('Upres must be at least 1.')
        self.tileSizeHigh = self.tileSizeLow * self.upres
        self.simSizeHigh = self.simSizeLow * self.upres
        if self.dim == 2:
            self.tileSizeLow[0] = self.tileSizeHigh[0] = self.simSizeLow[0] = self.simSizeHigh[0] = 1
        if np.less(self.simSizeLow, self.tileSizeLow).any():
            self.TCError('Tile size {} can not be larger than sim size {}.'.format(self.tileSizeLow, self.simSizeLow))
        if densityMinimum < 0.:
            self.TCError('densityMinimum can not be negative.')
        self.densityMinimum = densityMinimum
        self.premadeTiles = premadeTiles
        self.useDataAug = False

        self.c_lists = {}
        self.c_low, self.c_lists[DATA_KEY_LOW] = self.parseChannels(channelLayout_low)
        self.c_high, self.c_lists[DATA_KEY_HIGH] = self.parseChannels(channelLayout_high)
        print('Dimension: {}, time dimension: {}'.format(self.dim, self.dim_t))
        for label, key, layout in (('Low', DATA_KEY_LOW, self.c_low), ('High', DATA_KEY_HIGH, self.c_high)):
            print('{}-res data:'.format(label))
            print('  channel layout: {}'.format(layout))
            print('  default channels: {}'.format(self.c_lists[key][C_KEY_DEFAULT]))
            if len(self.c_lists[key][C_KEY_VELOCITY]) > 0:
                print('  velocity channels: {}'.format(self.c_lists[key][C_KEY_VELOCITY]))
        self.data_flags = {}
        for key, layout, label in ((DATA_KEY_LOW, self.c_low, False), (DATA_KEY_HIGH, self.c_high, highIsLabel)):
            flags = {'isLabel': label, 'channels': len(layout), C_KEY_POSITION: False}
            for k in (C_KEY_VELOCITY, C_KEY_VORTICITY, C_KEY_OBSTACLE, C_KEY_FLAGS, C_KEY_K, C_KEY_EPS):
                flags[k] = len(self.c_lists[key][k]) > 0
            self.data_flags[key] = flags
        if loadPN:
            self.TCError('prev and next tiles not supported.')
        self.hasPN = loadPN
        self.padding = padding

        self.tile_shape_low = np.append(self.tileSizeLow, [self.data_flags[DATA_KEY_LOW]['channels']])
        self.tile_shape_high = np.append(self.tileSizeHigh, [self.data_flags[DATA_KEY_HIGH]['channels']])
        self.frame_shape_low = np.append(self.simSizeLow, [self.data_flags[DATA_KEY_LOW]['channels']])
        self.frame_shape_high = np.append(self.simSizeHigh, [self.data_flags[DATA_KEY_HIGH]['channels']])
        self.densityThreshold = (self.densityMinimum * self.tile_shape_low[0] * self.tile_shape_low[1]
                                 * self.tile_shape_low[2])
        self.data = {DATA_KEY_LOW: [], DATA_KEY_HIGH: []}
        total = partTrain + partTest + partVal
        self.part_train = partTrain / total
        self.part_test = partTest / total
        self.part_validation = partVal / total

    # ------------------------------------------------------------------ augmentation set-up (:227-317)
    def initDataAugmentation(self, rot=2, minScale=0.85, maxScale=1.15, flip=True):
        self.useDataAug = True
        ops = {AOPS_KEY_ROTATE: self.rotateVelocities, AOPS_KEY_SCALE: self.scaleVelocities,
               AOPS_KEY_ROT90: self.rotate90Velocities, AOPS_KEY_FLIP: self.flipVelocities}
        self.aops = {key: {op: {C_KEY_VELOCITY: fn, C_KEY_VORTICITY: fn} for op, fn in ops.items()}
                     for key in (DATA_KEY_LOW, DATA_KEY_HIGH)}
        msg = 'data augmentation: '
        self.do_rotation = rot == 2
        self.do_rot90 = rot == 1
        if self.do_rotation:
            msg += 'rotation, '
        if self.do_rot90:
            msg += 'rot90, '
            z, nz, x, y, nx, ny = (2, 1), (1, 2), (1, 0), (0, 2), (0, 1), (2, 0)
            self.cube_rot = {2: [[], [z], [z, z], [nz]],
                             3: [[], [x], [y], [x, x], [x, y], [y, x], [y, y], [nx], [x, x, y], [x, y, x], [x, y, y],
                                 [y, x, x], [y, y, x], [ny], [nx, y], [x, x, y, x], [x, x, y, y], [x, y, x, x], [x, ny],
                                 [y, nx], [ny, x], [nx, y, x], [x, y, nx], [x, ny, x]]}
        self.scaleFactor = [minScale, maxScale]
        self.do_scaling = not (minScale == 1 and maxScale == 1)
        if self.do_scaling:
            msg += 'scaling, '
        self.do_flip = flip
        if self.do_flip:
            msg += 'flip'
        print(msg + '.')
        self.interpolation_order = 1
        self.fill_mode = 'constant'

    # ------------------------------------------------------------------ data (:321-392)
    def addData(self, low, high, flip_vel_z=True):
        low = np.asarray(low)
        high = np.asarray(high)
        if len(low.shape) != len(high.shape):
            self.TCError('Data shape mismatch. Dimensions: %d vs %d' % (len(low.shape), len(high.shape)))
        if not (len(low.shape) == 4 or len(low.shape) == 5):
            self.TCError('Input must be single 3D data or sequence of 3D data.')
        if low.shape[-1] != self.dim_t * self.data_flags[DATA_KEY_LOW]['channels']:
            self.TCError('(Dim_t * Channels) configured for tilecreator (low-res) don\'t match (channels) of data: '
                         + format([low.shape[-1], self.dim_t, self.data_flags[DATA_KEY_LOW]['channels']]))
        if high.shape[-1] != self.dim_t * self.data_flags[DATA_KEY_HIGH]['channels']:
            self.TCError('(Dim_t * Channels) configured for tilecreator (high-res) don\'t match channels of data: '
                         + format([high.shape[-1], self.dim_t, self.data_flags[DATA_KEY_HIGH]['channels']]))
        low_shape, high_shape = low.shape, high.shape
        if len(low.shape) == 5:
            if low.shape[0] != high.shape[0]:
                self.TCError('unequal amount of low ({}) and high ({}) data.'.format(low.shape[1], high.shape[1]))
            low_shape, high_shape = low_shape[1:], high_shape[1:]
        else:
            low, high = [low], [high]
        if flip_vel_z and self.dim == 3:        # sign of vz of the siggraph-2018 3D data (:353-358)
            flipz = [1.0] * self.data_flags[DATA_KEY_LOW]['channels']
            flipz[3] = -1.0
            low = np.asarray(low) * np.array(flipz * self.dim_t).reshape((1, 1, 1, 1, -1))
            print("Note - flipped Z coord of velocities! Only for sigg18-3d-data! disable for other data...")
        if self.premadeTiles:
            if self.dim_t != 1:
                self.TCError('Currently, Dim_t = {} > 1 is not supported by premade tiles'.format(self.dim_t))
            if not np.array_equal(low_shape, self.tile_shape_low) or not np.array_equal(high_shape, self.tile_shape_high):
                self.TCError('Tile shape mismatch: is - specified\n\tlow: {} - {}\n\thigh {} - {}'.format(
                    low_shape, self.tile_shape_low, high_shape, self.tile_shape_high))
        else:
            one_low, one_high = list(low_shape), list(high_shape)
            one_low[-1] = low_shape[-1] // self.dim_t
            one_high[-1] = high_shape[-1] // self.dim_t
            if not np.array_equal(one_low, self.frame_shape_low) or not np.array_equal(one_high, self.frame_shape_high):
                self.TCError('Frame shape mismatch: is - specified\n\tlow: {} - {}\n\thigh {} - {}, given dim_t as {}'.format(
                    one_low, self.frame_shape_low, one_high, self.frame_shape_high, self.dim_t))
        self.data[DATA_KEY_LOW].extend(low)
        self.data[DATA_KEY_HIGH].extend(high)
        self.splitSets()

    def splitSets(self):
        length = len(self.data[DATA_KEY_LOW])
        end_train = int(length * self.part_train)
        end_test = end_train + int(length * self.part_test)
        self.setBorders = [end_train, end_test, length]
        print('Training set: {}'.format(self.setBorders[0]))
        print('Testing set:  {}'.format(self.setBorders[1] - self.setBorders[0]))
        print('Validation set:  {}'.format(self.setBorders[2] - self.setBorders[1]))

    def clearData(self):
        self.data = {DATA_KEY_LOW: [], DATA_KEY_HIGH: []}

    # ------------------------------------------------------------------ tiles (:403-450)
    def createTiles(self, data, tileShape, strides=-1):
        shape = data.shape
        pad = [self.padding, self.padding, self.padding, 0]
        if np.isscalar(strides):
            strides = list(tileShape) if strides <= 0 else [strides, strides, strides]
        else:
            strides = list(strides)
        if shape[0] <= 1:
            pad[0] = 0
            strides[0] = 1
        count = [(shape[i] - tileShape[i]) // strides[i] + 1 for i in range(3)]
        tiles = []
        for tz in range(count[0]):
            for ty in range(count[1]):
                for tx in range(count[2]):
                    z0, y0, x0 = tz * strides[0], ty * strides[1], tx * strides[2]
                    cur = data[z0:z0 + tileShape[0], y0:y0 + tileShape[1], x0:x0 + tileShape[2], :]
                    if self.padding > 0:
                        cur = np.pad(cur, [(p, p) for p in pad], 'edge')
                    tiles.append(cur)
        return np.array(tiles)

    def cutTile(self, data, tileShape, offset=[0, 0, 0]):
        offset = np.asarray(offset)
        tileShape = np.asarray(tileShape)
        tileShape[-1] = data.shape[-1]
        if np.less(data.shape[:3], tileShape[:3] + offset[:3]).any():
            self.TCError('Can\'t cut tile with shape {} and offset{} from data with shape {}.'.format(tileShape, offset, data.shape))
        o = [int(v) for v in offset[:3]]
        t = [int(v) for v in tileShape[:3]]
        tile = data[o[0]:o[0] + t[0], o[1]:o[1] + t[1], o[2]:o[2] + t[2], :]
        if not np.array_equal(tile.shape, tileShape):
            self.TCError('Wrong tile shape after cutting. is: {}. goal: {}.'.format(tile.shape, tileShape))
        return tile

    # ------------------------------------------------------------------ batches (:457-642)
    def selectRandomTiles(self, selectionSize, isTraining=True, augment=False, tile_t=1):
        """-> low [selectionSize, z, y, x, channels*tile_t], high [...]; z = 1 in 2D"""
        if isTraining:
            if self.setBorders[0] < 1:
                self.TCError('no training data.')
        elif (self.setBorders[1] - self.setBorders[0]) < 1:
            self.TCError('no test data.')
        if tile_t > self.dim_t:
            self.TCError('not enough coherent frames. Requested {}, given {}'.format(tile_t, self.dim_t))
        batch_low, batch_high = [], []
        for _ in range(selectionSize):
            if augment and self.useDataAug:
                low, high = self.generateTile(isTraining, tile_t)
            else:
                low, high = self.getRandomDatum(isTraining, tile_t)
                if not self.premadeTiles:
                    low, high = self.getRandomTile(low, high)
            batch_low.append(low)
            batch_high.append(high)
        return np.asarray(batch_low), np.asarray(batch_high)

    def generateTile(self, isTraining=True, tile_t=1):
        """one augmented low/high tile pair; the order of random draws is the reference's (:491-546)"""
        data = {}
        data[DATA_KEY_LOW], data[DATA_KEY_HIGH] = self.getRandomDatum(isTraining, tile_t)
        if not self.premadeTiles:
            if self.do_scaling or self.do_rotation:
                factor = 1
                if self.do_rotation:
                    factor *= 1.5
                if self.do_scaling:
                    scaleFactor = np.random.uniform(self.scaleFactor[0], self.scaleFactor[1])
                    factor /= scaleFactor
                tileShapeLow = np.ceil(self.tile_shape_low * factor)
                if self.dim == 2:
                    tileShapeLow[0] = 1
                data[DATA_KEY_LOW], data[DATA_KEY_HIGH] = self.getRandomTile(
                    data[DATA_KEY_LOW], data[DATA_KEY_HIGH], tileShapeLow.astype(int))
            if self.do_scaling:
                data = self.scale(data, scaleFactor)
            bounds = np.zeros(4)
            if self.do_rotation:
                bounds = np.array(data[DATA_KEY_LOW].shape) * 0.16
                data = self.rotate(data)
            data[DATA_KEY_LOW], data[DATA_KEY_HIGH] = self.getRandomTile(data[DATA_KEY_LOW], data[DATA_KEY_HIGH], bounds=bounds)
        if self.do_rot90:
            rots = self.cube_rot[self.dim]
            for axis in rots[np.random.choice(len(rots))]:
                data = self.rotate90(data, axis)
        if self.do_flip:
            axis = np.random.choice(4)
            if axis < 3:
                data = self.flip(data, [axis])
        target_low = np.copy(self.tile_shape_low)
        target_high = np.copy(self.tile_shape_high)
        target_low[-1] *= tile_t
        target_high[-1] *= tile_t
        if not np.array_equal(data[DATA_KEY_LOW].shape, target_low) or not np.array_equal(data[DATA_KEY_HIGH].shape, target_high):
            self.TCError('Wrong tile shape after data augmentation. is: {},{}. goal: {},{}.'.format(
                data[DATA_KEY_LOW].shape, data[DATA_KEY_HIGH].shape, target_low, target_high))
        return data[DATA_KEY_LOW], data[DATA_KEY_HIGH]

    def getRandomDatum(self, isTraining=True, tile_t=1):
        if isTraining:
            randNo = randrange(0, self.setBorders[0])
        else:
            randNo = randrange(self.setBorders[0], self.setBorders[1])
        randFrame = 0
        if tile_t < self.dim_t:
            randFrame = randrange(0, self.dim_t - tile_t)
        else:
            tile_t = self.dim_t
        return self.getDatum(randNo * self.dim_t + randFrame, tile_t)

    def getDatum(self, index, tile_t=1):
        """copies of frame `index // dim_t`, channels of tile_t coherent frames starting at index % dim_t"""
        cl, ch = self.tile_shape_low[-1], self.tile_shape_high[-1]
        b_low = (index % self.dim_t) * cl if self.dim_t > 1 else 0
        b_high = (index % self.dim_t) * ch if self.dim_t > 1 else 0
        return (np.copy(self.data[DATA_KEY_LOW][index // self.dim_t][:, :, :, b_low:b_low + tile_t * cl]),
                np.copy(self.data[DATA_KEY_HIGH][index // self.dim_t][:, :, :, b_high:b_high + tile_t * ch]))

    def getRandomTile(self, low, high, tileShapeLow=None, bounds=[0, 0, 0, 0]):
        """random low/high tile pair with at least densityMinimum mean density (20 tries); `bounds`
        excludes the frame borders (mirrored parts after a rotation)"""
        if tileShapeLow is None:
            tileShapeLow = np.copy(self.tile_shape_low)
        if np.isscalar(self.upres):
            tileShapeHigh = tileShapeLow * self.upres
            offset_up = np.array([self.upres, self.upres, self.upres])
        else:   # hard-coded in the reference for non-scalar upres (:594,610)
            tileShapeHigh = tileShapeLow * np.array((1, 1, 8, 1))
            offset_up = [1, 1, 4]
        frameShapeLow = np.asarray(low.shape)
        if len(low.shape) != 4 or len(high.shape) != 4 or len(tileShapeLow) != 4:
            self.TCError('Data shape mismatch.')
        start = np.floor(bounds)
        end = frameShapeLow - tileShapeLow + np.ones(4) - start
        if self.dim == 2:
            start[0] = 0
            end[0] = 1
            offset_up[0] = 1
            tileShapeHigh[0] = 1
        if np.amin((end - start)[:3]) < 0:
            self.TCError('Can\'t cut tile {} from frame {} with bounds {}.'.format(tileShapeLow, frameShapeLow, start))
        ok = False
        i = 1
        while (not ok) and i < 20:
            offset = np.asarray([randrange(int(start[0]), int(end[0])), randrange(int(start[1]), int(end[1])),
                                 randrange(int(start[2]), int(end[2]))])
            lowTile = self.cutTile(low, tileShapeLow, offset)
            offset *= offset_up
            if not self.data_flags[DATA_KEY_HIGH]['isLabel']:
                highTile = self.cutTile(high, tileShapeHigh, offset)
            else:
                highTile = high
            ok = self.hasMinDensity(lowTile)
            i += 1
        return lowTile, highTile

    # ------------------------------------------------------------------ augmentation ops (:648-879)
    def special_aug(self, data, ops_key, param):
        """channel-type specific part of an augmentation (velocity vectors rotate / flip / scale with the grid)"""
        for data_key in data:
            orig_shape = data[data_key].shape
            tile_t = orig_shape[-1] // self.data_flags[data_key]['channels']
            arr = data[data_key]
            if tile_t > 1:
                arr = arr.reshape((-1, tile_t, self.data_flags[data_key]['channels']))
            for c_key, op in self.aops[data_key][ops_key].items():
                if self.data_flags[data_key][c_key] and not self.data_flags[data_key]['isLabel']:
                    arr = op(arr, self.c_lists[data_key][c_key], param)
            if tile_t > 1:
                data[data_key] = arr.reshape(orig_shape)
        return data

    def rotate(self, data):
        if self.dim == 2:
            theta = np.pi * np.random.uniform(0, 2)
            c, s = np.cos(theta), np.sin(theta)
            rotation_matrix = np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]])
        else:
            quat = np.random.normal(size=4)
            quat /= np.linalg.norm(quat)
            q = np.outer(quat, quat) * 2
            rotation_matrix = np.array([[1 - q[2, 2] - q[3, 3], q[1, 2] - q[3, 0], q[1, 3] + q[2, 0], 0],
                                        [q[1, 2] + q[3, 0], 1 - q[1, 1] - q[3, 3], q[2, 3] - q[1, 0], 0],
                                        [q[1, 3] - q[2, 0], q[2, 3] + q[1, 0], 1 - q[1, 1] - q[2, 2], 0],
                                        [0, 0, 0, 1]])
        data = self.special_aug(data, AOPS_KEY_ROTATE, rotation_matrix)
        for data_key in data:
            if not self.data_flags[data_key]['isLabel']:
                data[data_key] = self.applyTransform(data[data_key], rotation_matrix.T)
        return data

    def rotateVelocities(self, datum, c_list, rotationMatrix):
        rot3 = rotationMatrix[:3, :3]
        rot2 = rotationMatrix[1:3, 1:3]
        channels = np.split(datum, datum.shape[-1], -1)
        for v in c_list:
            if len(v) == 3:     # z,y,x order to match the rotation matrix
                vel = rot3.dot(np.stack([channels[v[2]].flatten(), channels[v[1]].flatten(), channels[v[0]].flatten()]))
                channels[v[2]] = np.reshape(vel[0], channels[v[2]].shape)
                channels[v[1]] = np.reshape(vel[1], channels[v[1]].shape)
                channels[v[0]] = np.reshape(vel[2], channels[v[0]].shape)
            if len(v) == 2:
                vel = np.concatenate([channels[v[1]], channels[v[0]]], -1)
                shape = vel.shape
                vel = np.split(np.reshape(rot2.dot(np.reshape(vel, (-1, 2)).T).T, shape), 2, -1)
                channels[v[1]], channels[v[0]] = vel[0], vel[1]
        return np.concatenate(channels, -1)

    def rotate90(self, data, axes):
        if len(axes) != 2:
            self.TCError('need 2 axes for rotate90.')
        for data_key in data:
            if not self.data_flags[data_key]['isLabel']:
                data[data_key] = np.rot90(data[data_key], axes=axes)
        return self.special_aug(data, AOPS_KEY_ROT90, axes)

    def rotate90Velocities(self, datum, c_list, axes):
        if len(axes) != 2:
            self.TCError('need 2 axes for rotate90.')
        channels = np.split(datum, datum.shape[-1], -1)
        for v in c_list:        # grid axes z,y,x <-> velocity components x,y,z
            a, b = v[-axes[0] + 2], v[-axes[1] + 2]
            channels[a], channels[b] = -channels[b], channels[a]
        return np.concatenate(channels, -1)

    def flip(self, data, axes, isFrame=True):
        if not isFrame:
            axes = np.asarray(axes) + np.ones(np.asarray(axes).shape)
        for axis in axes:
            for data_key in data:
                if not self.data_flags[data_key]['isLabel']:
                    data[data_key] = np.flip(data[data_key], axis)
        return self.special_aug(data, AOPS_KEY_FLIP, axes)

    def flipVelocities(self, datum, c_list, axes):
        channels = np.split(datum, datum.shape[-1], -1)
        for v in c_list:
            if 2 in axes:
                channels[v[0]] *= (-1)
            if 1 in axes:
                channels[v[1]] *= (-1)
            if 0 in axes and len(v) == 3:
                channels[v[2]] *= (-1)
        return np.concatenate(channels, -1)

    def scale(self, data, factor):
        """resample the frame to round(factor * resolution) (:808-845)"""
        scale = [factor, factor, factor, 1]
        if self.dim == 2:
            scale[0] = 1
        shape = np.array(data[DATA_KEY_LOW].shape)
        scale = np.round(shape * scale) / shape
        if len(data[DATA_KEY_LOW].shape) == 5:
            scale = np.append([1], scale)
        for data_key in data:
            if not self.data_flags[data_key]['isLabel']:
                data[data_key] = scipy.ndimage.zoom(data[data_key], scale, order=self.interpolation_order,
                                                    mode=self.fill_mode, cval=0.0)
        return self.special_aug(data, AOPS_KEY_SCALE, factor)

    def scaleVelocities(self, datum, c_list, factor):
        channels = np.split(datum, datum.shape[-1], -1)
        for v in c_list:
            channels[v[0]] *= factor
            channels[v[1]] *= factor
            if len(v) == 3:
                channels[v[2]] *= factor
        return np.concatenate(channels, -1)

    def applyTransform(self, data, transform_matrix, data_dim=3):
        """affine transform about the frame centre, channel by channel (:858-879)"""
        if len(data.shape) != 4:
            self.TCError('Data shape mismatch.')
        offset = np.array(data.shape) / 2 - np.array([0.5, 0.5, 0.5, 0])
        to_centre = np.eye(4)
        to_centre[:3, 3] = offset[:3]
        back = np.eye(4)
        back[:3, 3] = -offset[:3]
        m = np.dot(np.dot(to_centre, transform_matrix), back)
        channels = [scipy.ndimage.affine_transform(ch, m[:data_dim, :data_dim], m[:data_dim, data_dim],
                                                   order=self.interpolation_order, mode=self.fill_mode, cval=0.)
                    for ch in np.rollaxis(data, 3, 0)]
        return np.stack(channels, axis=-1)

    # ------------------------------------------------------------------ helpers (:886-931)
    def concatTiles(self, tiles, frameShape, tileBorder=[0, 0, 0, 0]):
        if len(tiles.shape) != 5 or len(frameShape) != 3 or len(tileBorder) != 4:
            self.TCError('Data shape mismatch.')
        if frameShape[0] * frameShape[1] * frameShape[2] != len(tiles):
            self.TCError('given tiles do not match required tiles.')
        tileBorder = np.asarray(tileBorder)
        if np.less(np.zeros(4), tileBorder).any():
            shape = tiles.shape[1:] - 2 * tileBorder
            tiles = [self.cutTile(t, shape, tileBorder) for t in tiles]
        frame = []
        for z in range(frameShape[0]):
            rows = []
            for y in range(frameShape[1]):
                off = z * frameShape[1] * frameShape[2] + y * frameShape[2]
                rows.append(np.concatenate(tiles[off:off + frameShape[2]], axis=2))
            frame.append(np.concatenate(rows, axis=1))
        return np.concatenate(frame, axis=0)

    def hasMinDensity(self, tile):
        return self.getTileDensity(tile) >= (self.densityMinimum * tile.shape[0] * tile.shape[1] * tile.shape[2])

    def getTileDensity(self, tile):
        if self.data_flags[DATA_KEY_LOW]['channels'] > 1:
            tile = np.split(tile, [1], axis=-1)[0]
        return tile.sum(dtype=np.float64)

    def getFrameTiles(self, index):
        low, high = self.getDatum(index)
        return self.createTiles(low, self.tile_shape_low), self.createTiles(high, self.tile_shape_high)

    # ------------------------------------------------------------------ channel layout (:937-1057)
    def parseChannels(self, channelString):
        """'d' scalar data, 'v[label](x|y|z)' vector components that follow the grid transforms"""
        c = [k.strip() for k in channelString.lower().split(',')]
        c_types = {k: [] for k in _SCALAR_KEYS + _VECTOR_KEYS}
        for i, key in enumerate(c):
            if len(key) == 0:
                self.TCError('empty channel key.')
            kind = key[0]
            if kind in _SCALAR_KEYS:
                if key != kind:
                    self.TCError('channel {}: unknown channel key "{}".'.format(i, key))
                c_types[kind].append(i)
            elif kind in _VECTOR_KEYS:
                if key[-1] not in 'xyz' or len(key) < 2:
                    self.TCError('channel {}: unknown channel key "{}".'.format(i, key))
                label = key[1:-1]
                names = [kind + label + a for a in 'xyz']
                if any(c.count(nm) > 1 for nm in names):
                    self.TCError('duplicate velocity channel with label "{}".'.format(label))
                if c.count(names[0]) == 0 or c.count(names[1]) == 0 or (self.dim == 3 and c.count(names[2]) == 0):
                    self.TCError('missing velocity channel with label "{}".'.format(label))
                if key[-1] == 'x':
                    # like the reference, z is looked up unconditionally (ValueError in 2D layouts without it)
                    c_types[kind].append([c.index(names[0]), c.index(names[1]), c.index(names[2])])
            else:
                self.TCError('channel {}: unknown channel key "{}".'.format(i, key))
        return c, c_types

    def TCError(self, msg):
        raise TilecreatorError(msg)


# ----------------------------------------------------------------------------------------------
# batch helpers for the temporal discriminator (:1293-1412)
# ----------------------------------------------------------------------------------------------
def gridInterpolBatch(macgridbatch, targetshape, order=1):
    """resample [b,z,y,x,c] to targetshape with cell-centred coordinates, no mixing of batch / channels"""
    assert targetshape[-1] == macgridbatch.shape[-1]
    assert len(targetshape) == 5 and len(macgridbatch.shape) == 5
    axes = [np.linspace(0, targetshape[0] - 1, targetshape[0])]
    for k in (1, 2, 3):
        axes.append(np.linspace(0.5, targetshape[k] - 0.5, targetshape[k]) * (float(macgridbatch.shape[k]) / targetshape[k]))
    axes.append(np.linspace(0, targetshape[4] - 1, targetshape[4]))
    coords = np.meshgrid(*axes, indexing='ij')
    return scipy.ndimage.map_coordinates(macgridbatch, coords, order=order, mode='nearest')


def getMACGridCenteredBatch(macgrid_batch, is3D):
    """staggered (MAC) velocities -> cell centres; components come back in z,y,x (2D: y,x) order"""
    bn, zn, yn, xn, _ = macgrid_batch.shape
    nxt_x = macgrid_batch.take(list(range(1, xn)) + [xn - 1], axis=3)[..., 0].reshape([bn, zn, yn, xn, 1])
    nxt_y = macgrid_batch.take(list(range(1, yn)) + [yn - 1], axis=2)[..., 1].reshape([bn, zn, yn, xn, 1])
    if is3D:
        nxt_z = macgrid_batch.take(list(range(1, zn)) + [zn - 1], axis=1)[..., 2].reshape([bn, zn, yn, xn, 1])
        res = 0.5 * (macgrid_batch[..., ::-1] + np.concatenate((nxt_z, nxt_y, nxt_x), axis=-1))
        return res.reshape([bn, zn, yn, xn, 3])
    res = 0.5 * (macgrid_batch[..., -2::-1] + np.concatenate((nxt_y, nxt_x), axis=4))
    return res.reshape([bn, yn, xn, 2])


def getSemiLagrPosBatch(macgrid_batch, dt, cube_len_output=-1):
    """semi-Lagrangian look-up positions pos - v*dt on a cube_len_output grid: [b,y,x,2] (2D)"""
    assert len(macgrid_batch.shape) == 5
    bn, zn, yn, xn, cn = macgrid_batch.shape
    assert cn == 3
    is3D = zn > 1
    if is3D:
        raise NotImplementedError("3D positions use an undefined `factor` in the reference (tilecreator_t.py:1360)")
    if cube_len_output == -1:
        cube_len_output = xn
    fx = float(xn) / cube_len_output
    fy = float(yn) / cube_len_output
    nx, ny = int(xn / fx + 0.5), int(yn / fy + 0.5)
    y, x = np.meshgrid(np.linspace(0.5, ny - 0.5, ny), np.linspace(0.5, nx - 0.5, nx), indexing='ij')
    pos = np.stack((y, x), axis=-1).reshape([1, ny, nx, 2])
    if cube_len_output == xn:
        return pos - getMACGridCenteredBatch(macgrid_batch, is3D) * dt
    inter = gridInterpolBatch(macgrid_batch, [bn, 1, ny, nx, 3], 1)
    return pos - (getMACGridCenteredBatch(inter, is3D) / fx) * dt


def selectRandomTempoTiles(self, selectionSize, isTraining=True, augment=False, n_t=3, dt=0.25):
    """coherent batches: [n_t * (selectionSize // n_t), ...] flattened low tiles, high tiles and the
    advection look-up positions of every frame (:1382-1412)"""
    batch_sz = int(max(1, selectionSize // n_t))
    batch_low, batch_high = self.selectRandomTiles(batch_sz, isTraining, augment, tile_t=n_t)
    real = batch_sz * n_t
    tl, th = self.tileSizeLow, self.tileSizeHigh
    low = np.transpose(batch_low.reshape((batch_sz, tl[0], tl[1], tl[2], n_t, -1)), (0, 4, 1, 2, 3, 5))
    low = low.reshape((real, tl[0], tl[1], tl[2], -1))
    mac = low[:, :, :, :, self.c_lists[DATA_KEY_LOW][C_KEY_VELOCITY][0]].reshape((real, tl[0], tl[1], tl[2], 3))
    dtArray = np.array([i * dt for i in range(n_t // 2, -n_t // 2, -1)] * batch_sz, dtype=np.float32)
    dtArray = dtArray.reshape((-1, 1, 1, 1)) if self.dim == 2 else dtArray.reshape((-1, 1, 1, 1, 1))
    pos = getSemiLagrPosBatch(mac, dtArray, th[1]).reshape((real, -1))
    high = np.transpose(batch_high.reshape((batch_sz, th[0], th[1], th[2], n_t, -1)), (0, 4, 1, 2, 3, 5))
    return low.reshape((real, -1)), high.reshape((real, -1)), pos


TileCreator.selectRandomTempoTiles = selectRandomTempoTiles


# ----------------------------------------------------------------------------------------------
# image output (Pillow instead of scipy.misc.toimage, tilecreator_t.py:1125-1157)
# ----------------------------------------------------------------------------------------------
def savePngsGrayscale(tiles, path, imageCounter=0, tiles_in_image=[1, 1], channels=[0], save_gif=False,
                      plot_vel_x_y=False):
    """tiles [tile,y,x,c] -> grids of tiles_in_image (rows, cols), one PNG per image and channel, values clipped to [0,1]"""
    from PIL import Image
    per = tiles_in_image[0] * tiles_in_image[1]
    if len(tiles) % per != 0:
        print('ERROR: number of tiles does not match tiles per image')
        return
    tiles = np.asarray(tiles)
    n_img = len(tiles) // per
    for image in range(n_img):
        rows = []
        for y in range(tiles_in_image[0]):
            off = image * per + y * tiles_in_image[1]
            rows.append(np.concatenate(tiles[off:off + tiles_in_image[1]], axis=1))
        img = np.rollaxis(np.concatenate(rows, axis=0), -1, 0)
        for i in channels:
            name = 'img_{:04d}.png'.format(imageCounter * n_img + image) if len(channels) == 1 else \
                'img_{:04d}_c{:04d}.png'.format(imageCounter * n_img + image, i)
            Image.fromarray((np.clip(img[i], 0.0, 1.0) * 255).astype(np.uint8)).save(path + name)
