"""Numbered ``.uni`` / ``.npz`` frames of ``sim_%04d`` directories -> one numpy array.

Same constructor arguments, ``get()`` tuple, shapes, dtypes and error class as
the reference's ``tools_wscale/fluiddataloader.py`` (FluidDataLoader :21,
loadFiles :387-544, loadDirs :548-590, get :616-619), including its quirks:
``select_random`` only truncates (the ``np.random.shuffle`` result is unused,
:338-347, though the RNG state still advances) and the slice conversion keeps
every slice in a zero-initialised array sized by ``select_random`` (:479).
"""
import glob
import math
import os
import re

import numpy as np
import scipy.ndimage

from . import uniio

FDG_DTYPE = np.float32


class FluidDataLoaderError(Exception):
    """FDL errors"""


class FluidDataLoader(object):
    def __init__(self, print_info=1, base_path="../data/", base_path_y="../data/", simdirname="sim_%04d/", indices=[],
                 numpy_seed=17179023, filename=None, filename_index_min=0, filename_index_max=200, wildcard=None,
                 array_y=None, filename_y=None, func_y=None, data_fraction=1., shape=None, shape_y=None,
                 collapse_z=False, shuffle_on_load=False, conv_slices=False, conv_axis=0, density_threshold=0.002,
                 axis_scaling=[1, 1, 1, 1], axis_scaling_y=[0.25, 1, 1, 1], select_random=1.0, add_adj_idcs=False,
                 multi_file_list=None, multi_file_list_y=None, multi_file_idxOff=None, multi_file_idxOff_y=None,
                 postproc_func=None, postproc_func_y=None, np_load_string=None, np_load_string_y=None,
                 oldNamingScheme=False):
        self.base_path, self.base_path_y, self.simdirname, self.indices = base_path, base_path_y, simdirname, indices
        self.filename, self.filename_index_min, self.filename_index_max = filename, filename_index_min, filename_index_max
        self.wildcard = wildcard
        self.multi_file_list, self.multi_file_list_y = multi_file_list, multi_file_list_y
        self.multi_file_idxOff, self.multi_file_idxOff_y = multi_file_idxOff, multi_file_idxOff_y
        self.postproc_func, self.postproc_func_y = postproc_func, postproc_func_y
        self.filename_y, self.array_y, self.func_y = filename_y, array_y, func_y
        self.data_fraction, self.shape, self.shape_y = data_fraction, shape, shape_y
        self.collapse_z, self.shuffle_on_load = collapse_z, shuffle_on_load
        self.conv_slices, self.conv_axis, self.density_threshold = conv_slices, conv_axis, density_threshold
        self.axis_scaling, self.axis_scaling_y = axis_scaling, axis_scaling_y
        self.select_random, self.add_adj_idcs = select_random, add_adj_idcs
        self.np_load_string = np_load_string if np_load_string is not None else "arr_0"
        self.np_load_string_y = np_load_string_y if np_load_string_y is not None else self.np_load_string
        np.random.seed(numpy_seed)                                          # :129
        if (self.filename is not None) + (self.wildcard is not None) > 1:
            raise FluidDataLoaderError("FluidDataLoader error: for input data loading, only specify one of: input filename, or wildcard")
        if (self.filename_y is not None) + (self.array_y is not None) + (self.func_y is not None) > 1:
            raise FluidDataLoaderError("FluidDataLoader error:  for label data loading, only specify one of: input filename, array or function")
        self.print_info = print_info
        if self.print_info:
            print("FluidDataLoader init, path %s, filename %s" % (self.base_path, self.filename))
        self.oldNamingScheme = oldNamingScheme
        for lst, off, what in ((multi_file_list, multi_file_idxOff, ""), (multi_file_list_y, multi_file_idxOff_y, " for y")):
            if off is not None and lst is not None and len(lst) != len(off):
                raise FluidDataLoaderError("FluidDataLoader error: multi file list and idxOff lists%s have to match %s"
                                           % (what, [len(lst), len(off)]))
        self.x = self.y = self.xfn = None
        self.have_y_npz = False
        self.loadDirs()
        self.printStats()

    # ------------------------------------------------------------------ file names
    def getFilename(self, sim_index, fnbase, frame_index, file_path):
        if not self.oldNamingScheme:
            return os.path.join(file_path, os.path.join(self.simdirname % sim_index, fnbase % frame_index))
        return os.path.join(file_path, os.path.join(self.simdirname % (sim_index, frame_index),
                                                    fnbase % (sim_index, frame_index)))

    def collectFilenamesFromDir(self, list_index):
        """:181-267"""
        sim_index = self.indices[list_index]
        found = 0
        labelstr = ""
        if self.wildcard is not None:
            search_dir = os.path.join(self.base_path, self.simdirname % sim_index)
            files = sorted(f for f in os.listdir(search_dir)
                           if os.path.isfile(os.path.join(search_dir, f)) and re.search(self.wildcard, f))
            if len(files) < 1:
                raise FluidDataLoaderError("Error - no files found in directory '%s' with wildcard '%s' " % (search_dir, self.wildcard))
            n = max(1, int(len(files) * self.data_fraction))
            tf = float(len(files)) / n
            for t in range(n):
                fn = files[int(t * tf)]
                self.xfn.append(os.path.join(search_dir, fn))
                found += 1
                if self.filename_y is not None:
                    parts = self.filename_y.split("$")
                    if len(parts) != 2:
                        raise FluidDataLoaderError("Error - when using a wildcard for x, filename_y needs to contain exactly one '$' where the file id string from x will be inserted to build the filename for y. Current, invalid, filename_y is '%s' " % (self.filename_y))
                    fny = os.path.join(search_dir, parts[0] + re.search(self.wildcard, fn).group(1) + parts[1])
                    if not os.path.isfile(fny):
                        raise FluidDataLoaderError("Error - y file '%s' for x file '%s' doesnt exist in search dir '%s' " % (fny, fn, search_dir))
                    self.yfn.append(fny)
                    self.have_y_npz = True
                if self.array_y is not None:
                    self.y = [] if self.y is None else self.y
                    self.y.append(self.array_y[list_index])
                    labelstr = " with label " + format(self.array_y[list_index])
        else:
            span = self.filename_index_max - self.filename_index_min
            n = max(1, int(span * self.data_fraction))
            tf = float(span) / n
            for t in range(n):
                idx = int(self.filename_index_min + t * tf)
                fn = self.getFilename(sim_index, self.filename, idx, self.base_path)
                self.xfn.append(fn)
                found += 1
                if self.filename_y is not None:
                    self.yfn.append(self.getFilename(sim_index, self.filename_y, idx, self.base_path_y))
                    self.have_y_npz = True
                if self.array_y is not None:
                    self.y = [] if self.y is None else self.y
                    self.y.append(self.array_y[list_index])
                    labelstr = " with label " + format(self.array_y[list_index])
                if self.func_y is not None:
                    self.y = [] if self.y is None else self.y
                    self.y.append(self.func_y(list_index, sim_index, t, fn))
        if self.print_info:
            print("Found " + format(found) + " files from sim ID " + format(sim_index) + labelstr)

    def mogrifyFilenameIndex(self, fn, idxOffset):
        """shift the frame number in a file name, clamped to the index range (:349-367)"""
        m = re.search(r"(.*_)([\d]+)\.([\w]+)", fn)
        if not m:
            raise FluidDataLoaderError("FluidDataLoader error: got filename %s, but could not split up into name,4-digit and extension " % (fn))
        idx = max(self.filename_index_min, min(self.filename_index_max - 1, int(m.group(2)) + idxOffset))
        return "%s%04d.%s" % (m.group(1), idx, m.group(3))

    def loadSingleDatum(self, fn, lstr, idxOffset=0):
        if idxOffset != 0:
            fn = self.mogrifyFilenameIndex(fn, int(idxOffset))
        if self.print_info > 1:
            print("Loading: " + fn + ", " + lstr)
        if fn.endswith(".npz"):
            return np.load(fn)[lstr]
        if fn.endswith(".uni"):
            return uniio.readUni(fn)[1]
        raise FluidDataLoaderError("FluidDataLoader error: got filename %s, but only .uni or .npz supported at the moment " % (fn))

    # ------------------------------------------------------------------ per-datum transforms
    def getDim(self, shape):
        if len(shape) == 4:
            return 2 if shape[0] == 1 else 3
        if len(shape) == 5:
            return 4
        return -1

    def removeZComponent(self, x):
        if not self.collapse_z or self.getDim(x.shape) != 2 or x.shape[3] != 3:
            return x
        x2d = np.zeros((1, x.shape[1], x.shape[2], 2), dtype=FDG_DTYPE)
        x2d[..., 0], x2d[..., 1] = x[..., 0], x[..., 1]
        return x2d

    def removeSlices(self, fx, fy=None):
        """drop slices whose mean density is below the threshold (:295-313)"""
        keep = [i for i in range(fx.shape[0]) if float(np.average(fx[i, :, :, 0:1])) >= self.density_threshold]
        if fy is None:
            return fx[keep]
        return fx[keep], fy[keep]

    def addAdjSlices(self, fx):
        """previous / next slice density of each of the three frames as extra channels (:315-336)"""
        s = fx.shape
        fx = fx.reshape((s[0], s[1], s[2], 3, -1))
        c = fx.shape[4]
        out = np.zeros(fx.shape[:4] + (c + 2,))
        out[..., 0:c] = fx
        out[1:, ..., c] = fx[:-1, ..., 0]
        out[:-1, ..., c + 1] = fx[1:, ..., 0]
        return out.reshape((s[0], s[1], s[2], -1))

    def selectRandomSamples(self, fx, fy=None):
        """:338-347 -- np.random.shuffle returns None, so fx[None] adds an axis and nothing is
        shuffled; x-only data is returned whole (with that extra axis folded back on store), paired
        data is truncated to select_random."""
        n = int(fx.shape[0] * self.select_random)
        np.random.shuffle(np.arange(fx.shape[0]))      # advances the RNG like the reference
        if fy is None:
            return fx
        return fx[0:n], fy[0:n]

    def _load_multi(self, basename, lst, offs, lstr):
        fx = self.loadSingleDatum(basename, lstr, 0 if offs is None else offs[0])
        if lst is not None:
            if basename.find(lst[0]) < 0:
                raise FluidDataLoaderError("Error, input filename '%s' doesnt contain given string '%s'" % (basename, lst[0]))
            for i in range(1, len(lst)):
                part = self.loadSingleDatum(basename.replace(lst[0], lst[i]), lstr, 0 if offs is None else offs[i])
                fx = np.append(fx, part, axis=len(fx.shape) - 1)
        return fx

    def loadFiles(self):
        """:387-544"""
        n = len(self.xfn)
        true_n = 0
        for t in range(n):
            fx = self._load_multi(self.xfn[t], self.multi_file_list, self.multi_file_idxOff, self.np_load_string)
            if self.postproc_func is not None:
                fx = self.postproc_func(fx, self)
            if self.conv_slices:
                if self.conv_axis == 1:
                    fx = fx.transpose(1, 0, 2, 3)
                    if fx.shape[3] > 3:
                        for i in range(3):     # swap vy <-> vz of each packed frame (:419-421)
                            fx[..., [i * 4 + 2, i * 4 + 3]] = fx[..., [i * 4 + 3, i * 4 + 2]]
                elif self.conv_axis == 2:
                    fx = fx.transpose(2, 1, 0, 3)
                    if fx.shape[3] > 3:
                        for i in range(3):     # swap vx <-> vz (:427-429)
                            fx[..., [i * 4 + 1, i * 4 + 3]] = fx[..., [i * 4 + 3, i * 4 + 1]]
            fy = None
            if self.have_y_npz:
                fy = self._load_multi(self.yfn[t], self.multi_file_list_y, self.multi_file_idxOff_y, self.np_load_string_y)
                if self.postproc_func_y is not None:
                    fy = self.postproc_func_y(fy, self)
                if self.conv_slices:
                    if self.conv_axis == 1:
                        fy = fy.transpose(1, 0, 2, 3)
                    elif self.conv_axis == 2:
                        fy = fy.transpose(2, 1, 0, 3)
            fx = self.removeZComponent(fx)
            if self.x is None:
                self.data_shape = fx.shape
                if self.shape is None:
                    self.shape = fx.shape * np.asarray(self.axis_scaling)
                    if self.add_adj_idcs:
                        self.shape[3] += 6
                    self.do_zoom = False
                else:
                    self.do_zoom = True
                    self.zoom_shape = [float(self.shape[i]) / self.data_shape[i] for i in range(len(self.shape))]
                    if self.print_info:
                        print("Zoom for x by " + format(self.zoom_shape))
                if self.print_info:
                    print("Allocating x data for " + format(n) + " entries of size " + format(self.shape))
                if self.conv_slices:
                    self.x = np.zeros(tuple([int(n * self.shape[0] * self.select_random)] + list(self.shape[1:])), dtype=FDG_DTYPE)
                else:
                    self.x = np.zeros(tuple([n] + list(self.shape)), dtype=FDG_DTYPE)
            if self.have_y_npz:
                fy = self.removeZComponent(fy)
                if self.y is None:
                    self.data_shape_y = fy.shape * np.asarray(self.axis_scaling_y)
                    if self.shape_y is None:
                        self.shape_y = fy.shape
                        self.do_zoom = False
                    else:
                        self.do_zoom = True
                        self.zoom_shape_y = [float(self.shape_y[i]) / self.data_shape_y[i] for i in range(len(self.shape_y))]
                    if self.print_info:
                        print("Allocating y data for " + format(n) + " entries of size " + format(self.shape_y))
                    if self.conv_slices:
                        self.y = np.zeros(tuple([self.x.shape[0]] + list(self.shape_y[1:])), dtype=FDG_DTYPE)
                    else:
                        self.y = np.zeros(tuple([n] + list(self.shape_y)), dtype=FDG_DTYPE)
                if self.do_zoom:
                    fy = scipy.ndimage.zoom(fy, self.zoom_shape_y, order=1)
            if self.do_zoom:
                fx = scipy.ndimage.zoom(fx, self.zoom_shape, order=1)
            if self.conv_slices:
                fx = scipy.ndimage.zoom(fx, self.axis_scaling, order=1)
                if self.have_y_npz:
                    fy = scipy.ndimage.zoom(fy, self.axis_scaling_y, order=1)
                    if self.add_adj_idcs:
                        fx = self.addAdjSlices(fx)
                    fx, fy = self.removeSlices(fx, fy)
                    fx, fy = self.selectRandomSamples(fx, fy)
                else:
                    if self.add_adj_idcs:
                        # the reference reads an undefined name here (`fxs`, :521)
                        raise FluidDataLoaderError("conv_slices with add_adj_idcs needs y data (undefined `fxs` in the reference, fluiddataloader.py:521)")
                    fx = self.removeSlices(fx)
                    fx = self.selectRandomSamples(fx)
                self.x[true_n:true_n + fx.shape[0], :] = fx
                if self.have_y_npz:
                    self.y[true_n:true_n + fy.shape[0], :] = fy
                true_n += fx.shape[0]
            else:
                self.x[t, :] = fx
                if self.have_y_npz:
                    self.y[t, :] = fy
            if self.print_info and t == 0:
                print("loadFiles: data size x " + format(self.x.shape) + ((", y " + format(self.y.shape)) if self.filename_y is not None else ""))
        if self.conv_slices:
            if self.print_info:
                print("Removed " + format(self.x.shape[0] - true_n) + " slices by checking against the density_threshold "
                      + format(self.density_threshold) + " and randomly selecting " + format(self.select_random)
                      + " percent of the remaining frames.")
            self.x = self.x[0:true_n]
            self.y = self.y[0:true_n]

    def loadDirs(self):
        """:548-590"""
        self.xfn, self.yfn = [], []
        for i in range(len(self.indices)):
            self.collectFilenamesFromDir(i)
        if self.print_info > 1:
            print("\nfilenames x:")
            print("\n".join(self.xfn))
            if self.filename_y is not None:
                print("\nfilenames y:")
                print("\n".join(self.yfn))
        self.loadFiles()
        if self.collapse_z:
            if self.getDim(self.x[0].shape) == 2:
                self.x = np.reshape(self.x, [self.x.shape[0], self.shape[1], self.shape[2], self.shape[3]])
            if self.have_y_npz and self.getDim(self.y[0].shape) == 2:
                self.y = np.reshape(self.y, [self.y.shape[0], self.shape_y[1], self.shape_y[2], self.shape_y[3]])
        if self.shuffle_on_load:
            idxr = np.random.permutation(self.x.shape[0])
            self.x = self.x[idxr]
            if self.have_y_npz:
                self.y = self.y[idxr]
            self.xfn = [self.xfn[idxr[i]] for i in range(len(self.xfn))]
            if self.filename_y is not None:
                self.yfn = [self.yfn[idxr[i]] for i in range(len(self.yfn))]
            elif self.y is not None and not self.have_y_npz:
                self.y = [self.y[idxr[i]] for i in range(len(self.y))]

    # ------------------------------------------------------------------ info
    def arrayStats(self, values, weights=None):
        average = np.average(values)
        return (average, math.sqrt(np.average((values - average) ** 2)))

    def perChannelStats(self, values, info=None):
        if values.shape[-1] > 1:
            if info:
                print(format(info))
            for c in range(values.shape[-1]):
                print("\t\t" + format(c) + ": " + format(self.arrayStats(values[..., c])))

    def printStats(self):
        if self.print_info:
            print("Loaded " + format(self.x.shape[0]) + " datasets" + (", shuffled" if self.shuffle_on_load else ""))
            print("\tData shape x " + format(self.x.shape))
            print("\tx mean & std dev: " + format(self.arrayStats(self.x)))
            self.perChannelStats(self.x, "\tPer channel mean & std dev x: ")
            if self.have_y_npz:
                print("\tData shape y " + format(self.y.shape))
                print("\ty mean & std dev: " + format(self.arrayStats(self.y)))

    def get(self):
        """-> (x, y, filenames)"""
        return self.x, self.y, self.xfn

    def getFullInfo(self):
        ret = ""
        for i in range(len(self.xfn)):
            ret += "%d/%d, file %s, shape %s" % (i, len(self.xfn), self.xfn[i], format(self.x[i].shape))
            ret += ", x mean %s " % (format(np.mean(self.x[i])))
            if self.filename_y is not None:
                ret += ", file_y %s " % (self.yfn[i])
            if self.have_y_npz:
                ret += ", shape_y %s " % (format(self.y[i].shape))
                ret += ", y mean %s " % (format(np.mean(self.y[i])))
            if self.array_y is not None:
                ret += ", y %s " % (format(self.y[i]))
            ret += "\n"
        return ret
