"""Numbered ``.uni`` / ``.npz`` frames of ``sim_%04d`` directories -> one numpy array per side (x, y).

Constructor arguments, the ``get()`` tuple, array shapes / dtypes and the error class are those of the
reference's ``tools_wscale/fluiddataloader.py`` (FluidDataLoader :21-36, file enumeration :181-267, frame
assembly :369-385, loadFiles :387-544, loadDirs :548-590, get :616-619); its outputs are pinned by
fixtures the reference module itself produced (tests/golden/tools_golden.npz).

Organisation (not the reference's): ``_enumerate`` turns the constructor arguments into a list of
``_Entry`` records (all file names of one sample, resolved up front); ``_View`` is the per-side chain
of array transforms (slice-axis view with the velocity component swap, z collapse, resize); the
loader drives entries through the views and owns allocation, slice filtering and statistics.

Reference behaviours kept on purpose: ``select_random`` never shuffles -- the
permutation the reference draws is discarded, only the generator state advances -- it truncates paired
slice sets to the first ``int(n * select_random)`` and leaves unpaired ones whole; the decision whether
to resize is a single flag shared by x and y (y's choice wins); slice storage is sized by
``select_random`` before filtering.  Slice mode with ``add_adj_idcs`` but without y data is rejected
(the reference reads an undefined variable there, :521).
"""
import math
import os
import re

import numpy as np
import scipy.ndimage

from . import uniio

FDG_DTYPE = np.float32
_NUMBERED = re.compile(r"(.*_)([\d]+)\.([\w]+)")


class FluidDataLoaderError(Exception):
    """FDL errors"""


class _Entry(object):
    """one sample: the x file, optional y file / label"""
    __slots__ = ("x", "y", "label")

    def __init__(self, x, y=None, label=None):
        self.x, self.y, self.label = x, y, label


def _pick(count, fraction):
    """positions of max(1, int(count * fraction)) evenly spaced picks out of `count` items"""
    n = max(1, int(count * fraction))
    step = float(count) / n
    return [t * step for t in range(n)]


class _View(object):
    """array transforms of one side (x or y) between the file and the storage"""

    # conv_axis -> (axes order, channel pair swapped in each packed 4-channel frame)
    SLICE_AXES = {0: (None, None), 1: ((1, 0, 2, 3), (2, 3)), 2: ((2, 1, 0, 3), (1, 3))}

    def __init__(self, loader, swap_velocities, postproc):
        self.loader, self.swap, self.postproc = loader, swap_velocities, postproc

    def __call__(self, a):
        L = self.loader
        if self.postproc is not None:
            a = self.postproc(a, L)
        if L.conv_slices:
            order, pair = self.SLICE_AXES[L.conv_axis]
            if order is not None:
                a = a.transpose(order)
                if self.swap and a.shape[3] > 3:
                    # the slicing axis takes the place of z: exchange that velocity component with vz in each of
                    # the three packed (d,vx,vy,vz) frames (:416-431); in place on the transposed view
                    for f in range(3):
                        i, j = 4 * f + pair[0], 4 * f + pair[1]
                        a[..., [i, j]] = a[..., [j, i]]
        return L.removeZComponent(a)


class FluidDataLoader(object):
    def __init__(self, print_info=1, base_path="../data/", base_path_y="../data/", simdirname="sim_%04d/", indices=[],
                 numpy_seed=17179023, filename=None, filename_index_min=0, filename_index_max=200, wildcard=None,
                 array_y=None, filename_y=None, func_y=None, data_fraction=1., shape=None, shape_y=None,
                 collapse_z=False, shuffle_on_load=False, conv_slices=False, conv_axis=0, density_threshold=0.002,
                 axis_scaling=[1, 1, 1, 1], axis_scaling_y=[0.25, 1, 1, 1], select_random=1.0, add_adj_idcs=False,
                 multi_file_list=None, multi_file_list_y=None, multi_file_idxOff=None, multi_file_idxOff_y=None,
                 postproc_func=None, postproc_func_y=None, np_load_string=None, np_load_string_y=None,
                 oldNamingScheme=False):
        args = dict(locals())
        args.pop("self")
        for name in ("numpy_seed", "np_load_string", "np_load_string_y"):
            args.pop(name)
        self.__dict__.update(args)
        self.np_load_string = "arr_0" if np_load_string is None else np_load_string
        self.np_load_string_y = self.np_load_string if np_load_string_y is None else np_load_string_y
        np.random.seed(numpy_seed)                                          # :129
        if filename is not None and wildcard is not None:
            raise FluidDataLoaderError("FluidDataLoader error: for input data loading, only specify one of: input filename, or wildcard")
        if sum(v is not None for v in (filename_y, array_y, func_y)) > 1:
            raise FluidDataLoaderError("FluidDataLoader error:  for label data loading, only specify one of: input filename, array or function")
        for names, offs, side in ((multi_file_list, multi_file_idxOff, "x"), (multi_file_list_y, multi_file_idxOff_y, "y")):
            if names is not None and offs is not None and len(names) != len(offs):
                raise FluidDataLoaderError("FluidDataLoader error: multi file list and idxOff lists for %s have to match %s"
                                           % (side, [len(names), len(offs)]))
        if print_info:
            print("FluidDataLoader init, path %s, filename %s" % (base_path, filename))
        self.x = self.y = None
        self.have_y_npz = False
        self.loadDirs()
        self.printStats()

    # ------------------------------------------------------------------ which files
    def getFilename(self, sim_index, fnbase, frame_index, file_path):
        if self.oldNamingScheme:
            return os.path.join(file_path, os.path.join(self.simdirname % (sim_index, frame_index),
                                                        fnbase % (sim_index, frame_index)))
        return os.path.join(file_path, os.path.join(self.simdirname % sim_index, fnbase % frame_index))

    def _enumerate(self, list_index):
        """entries of one simulation directory (:181-267)"""
        sim = self.indices[list_index]
        label = self.array_y[list_index] if self.array_y is not None else None
        entries = []
        if self.wildcard is not None:
            folder = os.path.join(self.base_path, self.simdirname % sim)
            names = sorted(f for f in os.listdir(folder)
                           if os.path.isfile(os.path.join(folder, f)) and re.search(self.wildcard, f))
            if not names:
                raise FluidDataLoaderError("Error - no files found in directory '%s' with wildcard '%s' " % (folder, self.wildcard))
            for pos in _pick(len(names), self.data_fraction):
                fn = names[int(pos)]
                e = _Entry(os.path.join(folder, fn), label=label)
                if self.filename_y is not None:
                    parts = self.filename_y.split("$")
                    if len(parts) != 2:
                        raise FluidDataLoaderError("Error - when using a wildcard for x, filename_y needs to contain exactly one '$' "
                                                   "where the file id string from x will be inserted to build the filename for y. "
                                                   "Current, invalid, filename_y is '%s' " % (self.filename_y))
                    e.y = os.path.join(folder, parts[0] + re.search(self.wildcard, fn).group(1) + parts[1])
                    if not os.path.isfile(e.y):
                        raise FluidDataLoaderError("Error - y file '%s' for x file '%s' doesnt exist in search dir '%s' " % (e.y, fn, folder))
                entries.append(e)
        else:
            span = self.filename_index_max - self.filename_index_min
            for t, pos in enumerate(_pick(span, self.data_fraction)):
                frame = int(self.filename_index_min + pos)
                e = _Entry(self.getFilename(sim, self.filename, frame, self.base_path), label=label)
                if self.filename_y is not None:
                    e.y = self.getFilename(sim, self.filename_y, frame, self.base_path_y)
                if self.func_y is not None:
                    e.label = self.func_y(list_index, sim, t, e.x)
                entries.append(e)
        if self.print_info:
            print("Found %d files from sim ID %s%s" % (len(entries), sim, "" if label is None else " with label %s" % (label,)))
        return entries

    def mogrifyFilenameIndex(self, fn, idxOffset):
        """the same file name with its frame number shifted, clamped to the configured index range (:349-367)"""
        m = _NUMBERED.search(fn)
        if not m:
            raise FluidDataLoaderError("FluidDataLoader error: got filename %s, but could not split up into name,4-digit and extension " % (fn))
        frame = min(max(int(m.group(2)) + idxOffset, self.filename_index_min), self.filename_index_max - 1)
        return "%s%04d.%s" % (m.group(1), frame, m.group(3))

    def loadSingleDatum(self, fn, lstr, idxOffset=0):
        if idxOffset != 0:
            fn = self.mogrifyFilenameIndex(fn, int(idxOffset))
        if self.print_info > 1:
            print("Loading: " + fn + ", " + lstr)
        ext = os.path.splitext(fn)[1]
        if ext == ".uni":
            return uniio.readUni(fn)[1]
        if ext == ".npz":
            return np.load(fn)[lstr]
        raise FluidDataLoaderError("FluidDataLoader error: got filename %s, but only .uni or .npz supported at the moment " % (fn))

    def _assemble(self, first, names, offsets, key):
        """channels of several grids side by side: `first` with names[0] replaced by each of names[1:] (:369-385)"""
        shift = (lambda i: 0) if offsets is None else (lambda i: offsets[i])
        parts = [self.loadSingleDatum(first, key, shift(0))]
        if names is not None:
            if names[0] not in first:
                raise FluidDataLoaderError("Error, input filename '%s' doesnt contain given string '%s'" % (first, names[0]))
            parts += [self.loadSingleDatum(first.replace(names[0], names[i]), key, shift(i)) for i in range(1, len(names))]
        return parts[0] if len(parts) == 1 else np.concatenate(parts, axis=parts[0].ndim - 1)

    # ------------------------------------------------------------------ array helpers
    def getDim(self, shape):
        if len(shape) == 4:
            return 2 if shape[0] == 1 else 3
        return 4 if len(shape) == 5 else -1

    def removeZComponent(self, x):
        """2D vector grids stored with three components lose the third when collapse_z is set"""
        if self.collapse_z and self.getDim(x.shape) == 2 and x.shape[3] == 3:
            return np.ascontiguousarray(x[..., :2], dtype=FDG_DTYPE)
        return x

    def _dense_enough(self, fx):
        """indices of the slices whose mean density (channel 0) reaches the threshold (:295-313)"""
        return [i for i in range(fx.shape[0]) if float(np.average(fx[i, :, :, 0:1])) >= self.density_threshold]

    def removeSlices(self, fx, fy=None):
        keep = self._dense_enough(fx)
        return fx[keep] if fy is None else (fx[keep], fy[keep])

    def addAdjSlices(self, fx):
        """density of the previous / next slice of each of the three packed frames as two more channels per frame
        (zeros at the ends; float64 like the reference's scratch array) (:315-336)"""
        n, h, w, _ = fx.shape
        frames = fx.reshape((n, h, w, 3, -1))
        c = frames.shape[4]
        out = np.zeros((n, h, w, 3, c + 2))
        out[..., :c] = frames
        out[1:, ..., c] = frames[:-1, ..., 0]
        out[:-1, ..., c + 1] = frames[1:, ..., 0]
        return out.reshape((n, h, w, -1))

    def selectRandomSamples(self, fx, fy=None):
        """see the module docstring: draws (and discards) a permutation, truncates paired data only (:338-347)"""
        np.random.shuffle(np.arange(fx.shape[0]))
        if fy is None:
            return fx
        keep = int(fx.shape[0] * self.select_random)
        return fx[:keep], fy[:keep]

    # ------------------------------------------------------------------ loading
    def _allocate(self, rows, shape):
        return np.zeros((int(rows),) + tuple(int(s) for s in shape), dtype=FDG_DTYPE)

    def loadFiles(self):
        """:387-544"""
        n = len(self._entries)
        view_x = _View(self, True, self.postproc_func)
        view_y = _View(self, False, self.postproc_func_y)
        filled = 0
        for t, e in enumerate(self._entries):
            fx = view_x(self._assemble(e.x, self.multi_file_list, self.multi_file_idxOff, self.np_load_string))
            fy = None
            if self.have_y_npz:
                fy = view_y(self._assemble(e.y, self.multi_file_list_y, self.multi_file_idxOff_y, self.np_load_string_y))
            if self.x is None:                       # the first sample fixes the storage
                self.data_shape = fx.shape
                self.do_zoom = self.shape is not None
                if self.do_zoom:
                    self.zoom_shape = [float(s) / d for s, d in zip(self.shape, self.data_shape)]
                else:
                    self.shape = fx.shape * np.asarray(self.axis_scaling)
                    if self.add_adj_idcs:
                        self.shape[3] += 6
                rows = n * self.shape[0] * self.select_random if self.conv_slices else n
                self.x = self._allocate(rows, self.shape[1:] if self.conv_slices else self.shape)
                if self.print_info:
                    print("x storage %s for %d entries of shape %s" % (self.x.shape, n, list(self.shape)))
            if fy is not None and self.y is None:
                self.data_shape_y = fy.shape * np.asarray(self.axis_scaling_y)
                self.do_zoom = self.shape_y is not None          # shared with x, as in the reference (:470-476)
                if self.do_zoom:
                    self.zoom_shape_y = [float(s) / d for s, d in zip(self.shape_y, self.data_shape_y)]
                else:
                    self.shape_y = fy.shape
                rows = self.x.shape[0] if self.conv_slices else n
                self.y = self._allocate(rows, self.shape_y[1:] if self.conv_slices else self.shape_y)
            if self.do_zoom:
                fx = scipy.ndimage.zoom(fx, self.zoom_shape, order=1)
                if fy is not None:
                    fy = scipy.ndimage.zoom(fy, self.zoom_shape_y, order=1)
            if not self.conv_slices:
                self.x[t, :] = fx
                if fy is not None:
                    self.y[t, :] = fy
                continue
            # slice mode: every slice along axis 0 becomes a sample
            fx = scipy.ndimage.zoom(fx, self.axis_scaling, order=1)
            if fy is not None:
                fy = scipy.ndimage.zoom(fy, self.axis_scaling_y, order=1)
                if self.add_adj_idcs:
                    fx = self.addAdjSlices(fx)
                fx, fy = self.selectRandomSamples(*self.removeSlices(fx, fy))
                self.y[filled:filled + fy.shape[0], :] = fy
            else:
                if self.add_adj_idcs:
                    raise FluidDataLoaderError("conv_slices with add_adj_idcs needs y data (the reference reads an undefined "
                                               "variable there, fluiddataloader.py:521)")
                fx = self.selectRandomSamples(self.removeSlices(fx))
            self.x[filled:filled + fx.shape[0], :] = fx
            filled += fx.shape[0]
        if self.conv_slices:
            if self.print_info:
                print("kept %d of %d slice slots (density_threshold %s, select_random %s)"
                      % (filled, self.x.shape[0], self.density_threshold, self.select_random))
            self.x = self.x[:filled]
            self.y = self.y[:filled]

    def loadDirs(self):
        """:548-590"""
        self._entries = [e for i in range(len(self.indices)) for e in self._enumerate(i)]
        self.xfn = [e.x for e in self._entries]
        self.yfn = [e.y for e in self._entries if e.y is not None]
        self.have_y_npz = len(self.yfn) > 0
        if self.array_y is not None or self.func_y is not None:
            self.y = [e.label for e in self._entries]
        if self.print_info > 1:
            print("\nfilenames x:\n" + "\n".join(self.xfn))
            if self.filename_y is not None:
                print("\nfilenames y:\n" + "\n".join(self.yfn))
        self.loadFiles()
        if self.collapse_z:
            if self.getDim(self.x[0].shape) == 2:
                self.x = np.reshape(self.x, [self.x.shape[0], self.shape[1], self.shape[2], self.shape[3]])
            if self.have_y_npz and self.getDim(self.y[0].shape) == 2:
                self.y = np.reshape(self.y, [self.y.shape[0], self.shape_y[1], self.shape_y[2], self.shape_y[3]])
        if self.shuffle_on_load:
            order = np.random.permutation(self.x.shape[0])
            self.x = self.x[order]
            self.xfn = [self.xfn[order[i]] for i in range(len(self.xfn))]
            if self.have_y_npz:
                self.y = self.y[order]
            if self.filename_y is not None:
                self.yfn = [self.yfn[order[i]] for i in range(len(self.yfn))]
            elif self.y is not None and not self.have_y_npz:
                self.y = [self.y[order[i]] for i in range(len(self.y))]

    # ------------------------------------------------------------------ reporting
    def arrayStats(self, values, weights=None):
        mean = np.average(values)
        return (mean, math.sqrt(np.average((values - mean) ** 2)))

    def perChannelStats(self, values, info=None):
        if values.shape[-1] > 1:
            if info:
                print(format(info))
            for c in range(values.shape[-1]):
                print("\t\t%d: %s" % (c, self.arrayStats(values[..., c])))

    def printStats(self):
        if not self.print_info:
            return
        print("Loaded %d datasets%s" % (self.x.shape[0], ", shuffled" if self.shuffle_on_load else ""))
        print("\tData shape x %s, mean & std dev %s" % (self.x.shape, self.arrayStats(self.x)))
        self.perChannelStats(self.x, "\tPer channel mean & std dev x: ")
        if self.have_y_npz:
            print("\tData shape y %s, mean & std dev %s" % (self.y.shape, self.arrayStats(self.y)))

    def get(self):
        """-> (x, y, filenames)"""
        return self.x, self.y, self.xfn

    def getFullInfo(self):
        lines = []
        for i, fn in enumerate(self.xfn):
            s = "%d/%d, file %s, shape %s, x mean %s " % (i, len(self.xfn), fn, format(self.x[i].shape), format(np.mean(self.x[i])))
            if self.filename_y is not None:
                s += ", file_y %s " % (self.yfn[i])
            if self.have_y_npz:
                s += ", shape_y %s , y mean %s " % (format(self.y[i].shape), format(np.mean(self.y[i])))
            if self.array_y is not None:
                s += ", y %s " % (format(self.y[i]))
            lines.append(s + "\n")
        return "".join(lines)
