#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/ by importing the reference's
TensorFlow-free tool modules (uniio, fluiddataloader) from /root/reference and
running them on tiny synthetic data.  Run in the build container only:

    python tests/golden/make_golden.py

The fixtures hold inputs and expected outputs (arrays, decompressed bytes); no
reference source travels.  /root/reference does not exist on the GPU box.
"""
import gzip
import io
import os
import shutil
import sys
import tempfile

import numpy as np
import scipy.ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/tools_wscale")

import fluiddataloader as REF_FDL  # noqa: E402
import uniio as REF_UNI            # noqa: E402


def header(dx, dy, dz, vec3):
    return {
        "dimX": dx, "dimY": dy, "dimZ": dz, "gridType": 1 if not vec3 else 4, "elementType": 2 if vec3 else 1,
        "bytesPerElement": 12 if vec3 else 4, "info": b"golden".ljust(252, b"\0"), "dimT": 0, "timestamp": 1234567,
    }


def payload_bytes(path):
    with gzip.open(path, "rb") as f:
        return np.frombuffer(f.read(), dtype=np.uint8)


def main():
    rng = np.random.default_rng(20260104)
    out = {}
    tmp = tempfile.mkdtemp(prefix="golden_")
    try:
        # ---- 1. .uni codec: scalar and vec3 grids -------------------------------------------
        n = 6
        dens = rng.random((n, n, n, 1)).astype(np.float32)
        vel = rng.standard_normal((n, n, n, 3)).astype(np.float32)
        p_s, p_v = os.path.join(tmp, "s.uni"), os.path.join(tmp, "v.uni")
        REF_UNI.writeUni(p_s, header(n, n, n, False), dens)
        REF_UNI.writeUni(p_v, header(n, n, n, True), vel)
        out["uni_scalar_in"], out["uni_vec3_in"] = dens, vel
        out["uni_scalar_bytes"], out["uni_vec3_bytes"] = payload_bytes(p_s), payload_bytes(p_v)
        h, a = REF_UNI.readUni(p_s)
        assert np.array_equal(a, dens) and h["dimX"] == n
        # an old-style MNT2 stream as the reference reader understands it
        import struct
        raw = b"MNT2" + struct.pack("iiiiii256sQ", n, n, n, 1, 1, 4, b"old".ljust(256, b"\0"), 99) + dens.tobytes()
        p_old = os.path.join(tmp, "old.uni")
        with gzip.open(p_old, "wb") as f:
            f.write(raw)
        h2, a2 = REF_UNI.readUni(p_old)
        out["uni_mnt2_bytes"] = np.frombuffer(raw, dtype=np.uint8)
        out["uni_mnt2_dimT"] = np.int64(h2["dimT"])
        out["uni_mnt2_info_len"] = np.int64(len(h2["info"]))
        assert np.array_equal(a2, dens)

        # ---- 2. FluidDataLoader on a tiny sim ------------------------------------------------
        sim = os.path.join(tmp, "sim_1000")
        os.makedirs(sim)
        frames = 4
        low, hi = 8, 16
        d_low = rng.random((frames, low, low, low, 1)).astype(np.float32)
        v_low = rng.standard_normal((frames, low, low, low, 3)).astype(np.float32)
        d_hi = rng.random((frames, hi, hi, hi, 1)).astype(np.float32)
        for f in range(frames):
            REF_UNI.writeUni(os.path.join(sim, "density_low_%04d.uni" % f), header(low, low, low, False), d_low[f])
            REF_UNI.writeUni(os.path.join(sim, "velocity_low_%04d.uni" % f), header(low, low, low, True), v_low[f])
            REF_UNI.writeUni(os.path.join(sim, "density_high_%04d.uni" % f), header(hi, hi, hi, False), d_hi[f])
        out["fdl_d_low"], out["fdl_v_low"], out["fdl_d_hi"] = d_low, v_low, d_hi
        base = tmp + "/"
        # output mode of multipassGAN-out.py:133
        fl = REF_FDL.FluidDataLoader(print_info=0, base_path=base, base_path_y=base, numpy_seed=42,
                                     filename="density_low_%04d.uni", filename_index_min=0, oldNamingScheme=False,
                                     filename_y=None, filename_index_max=3, indices=[1000], data_fraction=1.0,
                                     multi_file_list=["density", "velocity"], multi_file_list_y=["density"])
        x, y, names = fl.get()
        out["fdl_out_x"] = x
        out["fdl_out_names"] = np.array([os.path.basename(s) for s in names])
        # training slice mode of multipassGAN-8x.py: 3 frames packed, conv axes 0/1/2
        mfl = ["density", "velocity"] * 3
        mol = [0, 0, 1, 1, 2, 2]
        for axis in (0, 1, 2):
            fl = REF_FDL.FluidDataLoader(print_info=0, base_path=base, base_path_y=base, numpy_seed=42, conv_slices=True,
                                         conv_axis=axis, select_random=0.5, density_threshold=0.45,
                                         axis_scaling_y=[0.5, 1, 1, 1] if axis == 0 else [1, 1, 1, 1],
                                         axis_scaling=[1, 1, 1, 1] if axis == 0 else [2, 1, 1, 1],
                                         filename="density_low_%04d.uni", oldNamingScheme=False,
                                         filename_y="density_high_%04d.uni", filename_index_max=3, filename_index_min=0,
                                         indices=[1000], data_fraction=1.0, multi_file_list=mfl, multi_file_idxOff=mol,
                                         multi_file_list_y=["density"] * 3, multi_file_idxOff_y=[0, 1, 2])
            x, y, _ = fl.get()
            out["fdl_slices_x_axis%d" % axis] = x
            out["fdl_slices_y_axis%d" % axis] = y
        # data_fraction / frame subsampling
        fl = REF_FDL.FluidDataLoader(print_info=0, base_path=base, base_path_y=base, numpy_seed=1,
                                     filename="density_low_%04d.uni", filename_index_min=0, filename_index_max=4,
                                     indices=[1000], data_fraction=0.5)
        x, _, names = fl.get()
        out["fdl_fraction_x"] = x
        out["fdl_fraction_names"] = np.array([os.path.basename(s) for s in names])

        # ---- 3. scipy.ndimage.zoom order 1 (K16) ----------------------------------------------
        v = rng.standard_normal((5, 6, 4, 2)).astype(np.float32)
        out["zoom_in"] = v
        for ax in range(3):
            for fac in (4, 8):
                z = [1, 1, 1, 1]
                z[ax] = fac
                out["zoom_ax%d_x%d" % (ax, fac)] = scipy.ndimage.zoom(v, z, order=1, mode="constant", cval=0.0)
    finally:
        shutil.rmtree(tmp)
    path = os.path.join(HERE, "tools_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
