"""numpy restatement of the volume <-> slice-batch marshalling around the
generators (``generate3DUniForNewNetwork``).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Volumes are
``[z,y,x,c]`` as stored in ``.uni`` files (tools_wscale/uniio.py:40-44).
"""
import numpy as np

from . import nets, ops

F32 = np.float32
CUTOFF = 0.0005   # multipassGAN-4x.py:1156, multipassGAN-out.py:614


def cutoff(v, thr=CUTOFF):
    """``dim_output[dim_output < 0.0005] = 0`` (multipassGAN-4x.py:1156-1157)."""
    out = np.array(v, dtype=F32, copy=True)
    out[out < F32(thr)] = 0
    return out


def swap_channels(a, c0, c1):
    """the ``temp_vel`` copies, e.g. multipassGAN-out.py:473-475."""
    tmp = np.copy(a[..., c0])
    a[..., c0] = a[..., c1]
    a[..., c1] = tmp
    return a


def add_adjacent(batch, n_ch):
    """``add_adj_idcs`` channels: previous / next slice density, zeros at the
    ends (multipassGAN-out.py:423-436)."""
    n = batch.shape[0]
    prev = np.zeros_like(batch[..., 0:1])
    nxt = np.zeros_like(batch[..., 0:1])
    prev[1:] = batch[:-1, ..., 0:1]
    nxt[:-1] = batch[1:, ..., 0:1]
    assert batch.shape[-1] == n_ch and n >= 2
    return np.concatenate([batch, prev, nxt], axis=3)


# ----------------------------------------------------------------------------
# 4x, one network per pass (multipassGAN-4x.py:1090-1169; example_run_output.py:4-8)
# ----------------------------------------------------------------------------
def pass1_4x(ps, low, up_res=4, batch_norm=True):
    """``upsamplingMode 2``, ``upsampleFirst``: zoom z, slices along z, result
    [z,y,x] with the <5e-4 cutoff of the written file (4x.py:1103,1126-1133,1156)."""
    sim = low.shape[0]
    xs = ops.zoom_axis_linear(low, 0, up_res).reshape(-1, sim, sim, low.shape[-1])
    out = nets.gen_resnet(ps, xs, up_res, 2, batch_norm)
    s = sim * up_res
    return cutoff(out.reshape(s, s, s))


def pass2_input_4x(x2, low, up_res=4, vel_scale=1.0):
    """slice batch of ``upsamplingMode 1`` (4x.py:278-283,1095,1113-1119):
    [x][z][y] planes of (pass-1 density, velocities*upRes zoomed in all axes),
    channels (d, vy, vz, vx) after the two swaps."""
    s = x2.shape[0]
    c = low.shape[-1]
    if c > 1:
        vel = low[..., 1:4].astype(F32) * F32(up_res)            # :278
        vel[..., 1:4] = F32(vel_scale) * vel[..., 1:4]            # :283 (acts on vy,vz only)
        for ax in range(3):                                       # :1095
            vel = ops.zoom_axis_linear(vel, ax, up_res)
        vol = np.concatenate([x2.reshape(s, s, s, 1), vel], axis=3)
    else:
        vol = x2.reshape(s, s, s, 1)
    nch = vol.shape[-1]
    xs = vol.reshape(1, s, s, s, nch).transpose(0, 3, 1, 2, 4).reshape(-1, s, s, nch).copy()
    if nch >= 4:
        swap_channels(xs, 2, 3)                                   # :1114-1116
        swap_channels(xs, 3, 1)                                   # :1117-1119
    return xs


def pass2_4x(ps, x2, low, up_res=4, batch_norm=True, vel_scale=1.0):
    """``upsamplingMode 1``: refine along x; result transposed back to [z,y,x]
    and cut off (4x.py:1139-1142,1156-1157)."""
    s = x2.shape[0]
    xs = pass2_input_4x(x2, low, up_res, vel_scale)
    out = nets.gen_resnet(ps, xs, up_res, 1, batch_norm)
    return cutoff(out.reshape(s, s, s).transpose(1, 2, 0))


def pass3_input_4x(x2, low, up_res=4, vel_scale=1.0):
    """slice batch of ``upsamplingMode 3`` (4x.py:278-283,1095,1121-1124): [y][z][x] planes of (second-pass
    density, velocities*upRes zoomed in all axes), channels (d, vx, vz, vy) after the one swap."""
    s = x2.shape[0]
    if low.shape[-1] > 1:
        vel = low[..., 1:4].astype(F32) * F32(up_res)            # :278
        vel[..., 1:4] = F32(vel_scale) * vel[..., 1:4]            # :283 (acts on vy,vz only)
        for ax in range(3):                                       # :1095
            vel = ops.zoom_axis_linear(vel, ax, up_res)
        vol = np.concatenate([x2.reshape(s, s, s, 1), vel], axis=3)
    else:
        vol = x2.reshape(s, s, s, 1)
    nch = vol.shape[-1]
    xs = vol.reshape(1, s, s, s, nch).transpose(0, 2, 1, 3, 4).reshape(-1, s, s, nch).copy()    # :1121
    if nch >= 4:
        swap_channels(xs, 2, 3)                                   # :1122-1124
    return xs


def pass3_4x(ps, x2, low, up_res=4, batch_norm=True, vel_scale=1.0):
    """``upsamplingMode 3``: refine along y; result back to [z,y,x] and cut off (4x.py:1144,1156-1157)."""
    s = x2.shape[0]
    out = nets.gen_resnet(ps, pass3_input_4x(x2, low, up_res, vel_scale), up_res, 3, batch_norm)
    return cutoff(out.reshape(s, s, s).transpose(1, 0, 2))


def two_pass_4x(ps1, ps2, low, up_res=4, batch_norm=True, vel_scale=1.0):
    """C1/C2 of BASELINE.json: the two ``multipassGAN-4x.py out 1`` runs of
    example_run_output.py:4-8 chained through the intermediate volume."""
    low = np.asarray(low, dtype=F32)
    low1 = np.array(low, copy=True)
    if low.shape[-1] > 1:
        low1[..., 1:4] = F32(vel_scale) * low1[..., 1:4]          # :283, run 1 (x_3d = x: all of vx,vy,vz)
    v1 = pass1_4x(ps1, low1, up_res, batch_norm)
    v2 = pass2_4x(ps2, v1, low, up_res, batch_norm, vel_scale)    # run 2 scales only vy,vz (quirk of :278-283)
    return v2, v1


# ----------------------------------------------------------------------------
# 8x, up to three networks in one process (multipassGAN-out.py:390-618)
# ----------------------------------------------------------------------------
def _gen_cfg(ps, c, xin, up_res, first, pixel_norm):
    return nets.growing_gen(ps, xin, up_res, first, c["filter_size"], c["start_fms"], c["max_fms"],
                            c.get("first_nn_arch", False) if first else False, c.get("use_res_net", True), pixel_norm)


def slice_batch_8x(low, up_res, pass_no, transpose_axis):
    """the low-res slice batch of one pass for a given ``transposeAxis``, written as the reference writes it:
    pass 1 :397-421, pass 2 :463-485, pass 3 :525-547"""
    sim = low.shape[0]
    nch = low.shape[-1]
    ta = transpose_axis

    def zoomed(axis):
        return ops.zoom_axis_linear(low, axis, up_res)

    def planes(a, order):
        return np.ascontiguousarray(a.transpose(order + (3,))).reshape(-1, sim, sim, nch)

    # which branch of the if-chain the pass takes: the reference rotates the roles of the four values
    kind = {1: {1: "y", 2: "x", 3: "x3", 0: "z"}, 2: {3: "y", 0: "x", 1: "x3", 2: "z"},
            3: {0: "y", 3: "x_odd", 2: "broken", 1: "z"}}[pass_no][ta]
    if kind == "z":
        return zoomed(0).reshape(-1, sim, sim, nch)
    if kind == "y":
        xs = planes(zoomed(1), (1, 0, 2))
        return swap_channels(xs, 3, 2) if nch >= 4 else xs
    if kind == "x":
        xs = planes(zoomed(2), (2, 1, 0))
        return swap_channels(xs, 3, 1) if nch >= 4 else xs
    if kind == "x3":
        xs = planes(zoomed(2), (2, 0, 1))
        if nch >= 4:
            t3, t2 = np.copy(xs[..., 3]), np.copy(xs[..., 2])
            xs[..., 3] = xs[..., 1]
            xs[..., 2] = t3
            xs[..., 1] = t2
        return xs
    if kind == "x_odd":
        xs = planes(zoomed(2), (0, 2, 1))                                        # :534-535
        return swap_channels(xs, 2, 1) if nch >= 4 else xs
    raise IndexError("index 13 is out of bounds for axis 3 with size %d" % nch)   # :542


def multipass_8x(ps_list, cfgs, low, up_res=8, pixel_norm=True, apply_cutoff=True, transpose_axis=0):
    """``generate3DUniForNewNetwork`` of multipassGAN-out.py.  ps_list/cfgs: one entry per loaded network
    (1..3); cfg keys: filter_size, start_fms, max_fms, add_adj, first_nn_arch, use_res_net.
    low: [z,y,x,4] with velocities already scaled by velScale (:138)."""
    sim = low.shape[0]
    s = sim * up_res
    nch = low.shape[-1]
    xs = slice_batch_8x(low, up_res, 1, transpose_axis)
    c = cfgs[0]
    if c.get("add_adj", False):
        xs = add_adjacent(xs, nch)
    out = _gen_cfg(ps_list[0], c, xs, up_res, True, pixel_norm)
    dim_output = out.reshape(s, s, s).transpose(2, 1, 0)                         # :459
    if len(ps_list) > 1:
        xs = slice_batch_8x(low, up_res, 2, transpose_axis)
        xin = nets.gen2_input(np.ascontiguousarray(dim_output).reshape(s, s, s, 1), xs, s)
        out = _gen_cfg(ps_list[1], cfgs[1], xin, up_res, False, pixel_norm)
        dim_output = out.reshape(s, s, s).transpose(1, 2, 0)                     # :521
    if len(ps_list) > 2:
        xs = slice_batch_8x(low, up_res, 3, transpose_axis)
        xin = nets.gen2_input(np.ascontiguousarray(dim_output).reshape(s, s, s, 1), xs, s)
        out = _gen_cfg(ps_list[2], cfgs[2], xin, up_res, False, pixel_norm)
        dim_output = out.reshape(s, s, s)                                        # :583
    if len(ps_list) > 1:
        dim_output = dim_output.transpose(2, 0, 1)                               # :587-588
    dim_output = dim_output.transpose(2, 1, 0)                                   # :589-590
    dim_output = np.ascontiguousarray(dim_output)
    return cutoff(dim_output) if apply_cutoff else dim_output


# ----------------------------------------------------------------------------
# 8x, one network per process (multipassGAN-8x.py:1600-1780)
# ----------------------------------------------------------------------------
def single_pass_8x(ps, cfg, low, prev=None, up_res=8, transpose_axis=0, pixel_norm=True, apply_cutoff=True):
    """one ``multipassGAN-8x.py out 1`` run: prev None = upsamplingMode 2 (first network), else the previous
    network's [z,y,x] volume (modes 1 / 3).  Returns the volume written to the .uni file."""
    sim = low.shape[0]
    s = sim * up_res
    nch = low.shape[-1]
    ta = transpose_axis
    axis = {0: 0, 1: 1, 2: 2, 3: 2}[ta]                                           # :1606-1613 / 1624-1631
    xs = ops.zoom_axis_linear(low, axis, up_res)
    order = {0: None, 1: (1, 0, 2), 2: (2, 1, 0), 3: (2, 0, 1)}[ta]
    ys = None if prev is None else np.asarray(prev, dtype=F32).reshape(s, s, s, 1)
    if order is not None:                                                        # :1644-1666
        xs = np.ascontiguousarray(xs.transpose(order + (3,)))
        if ys is not None:
            ys = np.ascontiguousarray(ys.transpose(order + (3,)))
    xs = xs.reshape(-1, sim, sim, nch)
    if nch >= 4:
        if ta == 1:
            swap_channels(xs, 3, 2)
        elif ta == 2:
            swap_channels(xs, 3, 1)
        elif ta == 3:
            t3, t2 = np.copy(xs[..., 3]), np.copy(xs[..., 2])
            xs[..., 3] = xs[..., 1]
            xs[..., 2] = t3
            xs[..., 1] = t2
    if cfg.get("add_adj", False):
        xs = add_adjacent(xs, nch)                                               # :1671-1685
    xin = xs if ys is None else nets.gen2_input(ys.reshape(s, s, s, 1), xs, s)
    out = _gen_cfg(ps, cfg, xin, up_res, prev is None, pixel_norm).reshape(s, s, s)
    back = {0: (0, 1, 2), 1: (1, 0, 2), 2: (2, 1, 0), 3: (1, 2, 0)}[ta]           # :1720-1725
    out = np.ascontiguousarray(out.transpose(back))
    return cutoff(out) if apply_cutoff else out
