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
# 8x, up to three networks in one process (multipassGAN-out.py:390-618), transposeAxis 0
# ----------------------------------------------------------------------------
def multipass_8x(ps_list, cfgs, low, up_res=8, pixel_norm=True, apply_cutoff=True):
    """``generate3DUniForNewNetwork`` of multipassGAN-out.py for
    ``transposeAxis 0``.  ps_list/cfgs: one entry per loaded network (2 or 3);
    cfg keys: filter_size, start_fms, max_fms, add_adj, first_nn_arch, use_res_net.
    low: [z,y,x,4] with velocities already scaled by velScale (:138)."""
    sim = low.shape[0]
    s = sim * up_res
    nch = low.shape[-1]
    dim_output = None
    # pass 1 (397-461)
    xs = ops.zoom_axis_linear(low, 0, up_res).reshape(-1, sim, sim, nch)
    c = cfgs[0]
    if c.get("add_adj", False):
        xs = add_adjacent(xs, nch)
    out = nets.growing_gen(ps_list[0], xs, up_res, True, c["filter_size"], c["start_fms"], c["max_fms"],
                           c.get("first_nn_arch", False), c.get("use_res_net", True), pixel_norm)
    dim_output = out.reshape(s, s, s).transpose(2, 1, 0)                         # :459 -> (x,y,z)
    # pass 2 (463-523)
    if len(ps_list) > 1:
        c = cfgs[1]
        xs = ops.zoom_axis_linear(low, 2, up_res).transpose(2, 1, 0, 3).copy()   # :471-472
        swap_channels(xs, 3, 1)                                                  # :473-475
        xin = nets.gen2_input(dim_output.reshape(s, s, s, 1), xs, s)
        out = nets.growing_gen(ps_list[1], xin, up_res, False, c["filter_size"], c["start_fms"], c["max_fms"],
                               False, c.get("use_res_net", True), pixel_norm)
        dim_output = out.reshape(s, s, s).transpose(1, 2, 0)                     # :521 -> (y,z,x)
    # pass 3 (525-585)
    if len(ps_list) > 2:
        c = cfgs[2]
        xs = ops.zoom_axis_linear(low, 1, up_res).transpose(1, 0, 2, 3).copy()   # :527-528
        swap_channels(xs, 3, 2)                                                  # :529-531
        xin = nets.gen2_input(dim_output.reshape(s, s, s, 1), xs, s)
        out = nets.growing_gen(ps_list[2], xin, up_res, False, c["filter_size"], c["start_fms"], c["max_fms"],
                               False, c.get("use_res_net", True), pixel_norm)
        dim_output = out.reshape(s, s, s)                                        # :583
    if len(ps_list) > 1:
        dim_output = dim_output.transpose(2, 0, 1)                               # :587-588
    dim_output = dim_output.transpose(2, 1, 0)                                   # :589-590
    dim_output = np.ascontiguousarray(dim_output)
    return cutoff(dim_output) if apply_cutoff else dim_output
