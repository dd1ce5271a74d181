"""Device-resident multi-pass volume pipeline (``generate3DUniForNewNetwork``).

The reference marshals every pass through host numpy (and, in the per-network
4x mode, through a gzip ``.uni`` file): GAN/multipassGAN-4x.py:1090-1169,
GAN/multipassGAN-out.py:390-618.  Here the low-res volume enters HBM once and
the final ``[z,y,x]`` volume leaves it once; the axis zoom, slice-batch
transposes, velocity channel swaps and the cutoff are HIP kernels
(``ops.axis_zoom_linear`` / ``volume_transpose`` / ``add_adjacent`` / ``cutoff``).

Every pass shards over its slice axis: rank r of R evaluates slices
``[r*S/R, (r+1)*S/R)`` and an all-gather reassembles the volume before the next
pass (``dist.Comm``); with one rank this degenerates to the plain loop.
The ``backend`` argument is the operator module (default: the HIP ``ops``).
"""
import os

import torch

from . import graph as G
from . import arch, ops
from .session import Session, VariableStore

CUTOFF = 0.0005     # multipassGAN-4x.py:1156, multipassGAN-out.py:614


class LocalComm(object):
    """single-rank stand-in of dist.Comm"""
    rank, world = 0, 1

    def all_gather_slabs(self, local, total):
        return local


EXCHANGES = ("all_gather", "all_to_all")


def _check_exchange(exchange):
    if exchange not in EXCHANGES:
        raise ValueError("exchange %r: expected one of %s" % (exchange, EXCHANGES))
    return exchange == "all_to_all"


def _hand_over(comm, out, s, perm, a2a, backend):
    """Pass p produced `out` = this rank's slab [S/R, S, S] of a volume V; pass p+1 slices Y = transpose(V, perm) along
    its axis 0 and this rank evaluates Y[lo:hi].  Returns (Y_local, V_full or None):
    all-gather: every rank receives all of V (S^3 elements) and keeps its rows of the transposed volume;
    all-to-all: only the [S/R, S, S/R] blocks that end up in Y[lo:hi] travel (S^3 / R elements per rank), V is never whole.
    (multipassGAN-out.py:459,521; multipassGAN-4x.py:1113)"""
    return _finish_hand_over(_start_hand_over(comm, out, s, perm, a2a), backend)


def _start_hand_over(comm, out, s, perm, a2a):
    if a2a and comm.world > 1 and perm[0] != 0:
        return ("a2a", comm.all_to_all_blocks_start(out, perm[0]), comm, s, perm)
    return ("gather", _start_gather(comm, out, s), comm, s, perm)


def _finish_hand_over(h, backend):
    kind, handle, comm, s, perm = h
    lo, hi = slice_range(s, comm)
    if kind == "a2a":
        part = handle.wait()                                   # [S, .., S/R, ..]: all of axis 0, this rank's range of axis perm[0]
        return backend.volume_transpose(part, perm), None
    full = handle.wait()
    y = backend.volume_transpose(full, perm) if tuple(perm) != (0, 1, 2) else full
    return y[lo:hi], full


def slice_range(total, comm):
    """contiguous slice range of this rank; ``total`` must divide by the world size"""
    if total % comm.world:
        raise ValueError("slice count %d does not divide over %d ranks" % (total, comm.world))
    per = total // comm.world
    return comm.rank * per, (comm.rank + 1) * per


# ----------------------------------------------------------------------------
# compiled generators
# ----------------------------------------------------------------------------
class Generator(object):
    """One generator network: private graph + session.  Call with device tensors
    x [N,h,w,C] (and y [N,H,W] for the later 8x generators); returns [N,H,W]."""

    def __init__(self, kind, cfg, params=None, prec=None, device="cuda:0", seed=777, prec_map=None):
        prec = ops.INFERENCE_PREC if prec is None else prec
        self.kind, self.cfg = kind, dict(cfg)
        prev = G.get_default_graph()
        self.graph = G.reset_default_graph()
        try:
            c = self.cfg
            low, up, nch = c["tile_low"], c["up_res"], c["channels"]
            self.high = low * up
            self.y = None
            if kind == "gen_resnet":
                mode = c.get("upsampling_mode", 2)
                side = low if mode == 2 else self.high
                self.x = G.placeholder([None, side * side * nch], name="x")
                self.sampler = arch.gen_resnet(self.x, low, up, nch, mode, use_batch_norm=c.get("batch_norm", True))
            elif kind == "growing_gen":
                first = c.get("first_gen", True)
                n_in = nch + (2 if c.get("add_adj", False) else 0)
                self.x = G.placeholder([None, low * low * n_in], name="x")
                src = self.x
                if not first:
                    self.y = G.placeholder([None, None], name="y")
                    src = arch.second_gen_input(self.x, self.y, low, self.high, nch)
                acfg = arch.Cfg8x(tileSizeLow=low, upRes=up, n_inputChannels=n_in if first else nch,
                                  upsampling_mode=2 if first else 1, upsampleMode=c.get("upsample_mode", 1),
                                  filterSize=c["filter_size"], start_fms=c["start_fms"], max_fms=c["max_fms"],
                                  first_nn_arch=c.get("first_nn_arch", False), use_res_net=c.get("use_res_net", True),
                                  pixel_norm=c.get("pixel_norm", True), addBicubicUpsample=c.get("add_bicubic", True))
                self.sampler = arch.growing_gen(src, acfg, use_batch_norm=c.get("batch_norm", False), output=True)
            else:
                raise ValueError("unknown generator kind %r" % (kind,))
        finally:
            G._default_graph[0] = prev
        self.sess = Session(device=device, prec=prec, graph=self.graph,
                            variables=VariableStore(device, seed=seed), prec_map=prec_map)
        if params is not None:
            self.sess.vars.load(params)
        self.sess.vars.ensure(self.graph)

    def params(self):
        return self.sess.vars.numpy()

    def clones(self, n):
        """n further copies for the pass lanes, made once"""
        have = self.__dict__.setdefault("_clones", [])
        while len(have) < n:
            have.append(self.clone())
        for c in have:
            if c._src_version != self.sess.vars.version:        # the weights were replaced (load): share the new tensors
                for name, v in self.sess.vars.values.items():
                    c.sess.vars.values[name] = v
                c.sess.vars.version += 1
                c._src_version = self.sess.vars.version
        return have[:n]

    def clone(self):
        """the same network and weights with its own session (workspaces, packed weights): a second lane"""
        g = Generator(self.kind, self.cfg, None, self.sess.prec, device=self.sess.device, prec_map=self.sess.prec_map)
        for name, v in self.sess.vars.values.items():
            g.sess.vars.values[name] = v
        g.sess.vars.version += 1
        g._src_version = self.sess.vars.version
        return g

    def __call__(self, x, y=None, out=None):
        """out: where the [n, high, high] result goes (a slice of the pass's volume), else a new tensor"""
        n = x.shape[0]
        feeds = {self.x: x.reshape(n, -1)}
        if self.y is not None:
            feeds[self.y] = y.reshape(n, -1)
        return self.sess.run_device(self.sampler, feeds, out=out).reshape(n, self.high, self.high)


# HIP streams the slice batches of one pass are dealt to (the batches are independent).  1 = the plain loop.
PASS_LANES = [max(1, int(os.environ.get("MPG_LANES", "2")))]
_NESTED = [False]          # set while two_pass_4x_batch runs whole volumes on lanes: no lanes inside lanes


def set_pass_lanes(k):
    PASS_LANES[0] = max(1, int(k))


def _run_pass(gen, xs, ys, lo, hi, batch, out=None):
    """the reference's per-pass sess.run loop (multipassGAN-out.py:443-447): slices [lo,hi) in batches.  With
    PASS_LANES > 1 the batches alternate over that many HIP streams (clones of the generator: same weights, own
    workspaces), so that the launch tails and latency-bound layers of one batch run under the next one's."""
    nb = (hi - lo + batch - 1) // batch
    lanes = 1 if (_NESTED[0] or not xs.is_cuda) else min(PASS_LANES[0], nb)
    if not xs.is_cuda:                           # the CPU rehearsal of the sharding logic (oracle generators)
        res = [gen(xs[j:min(j + batch, hi)], ys[j:min(j + batch, hi)] if ys is not None else None) for j in range(lo, hi, batch)]
        return res[0] if len(res) == 1 else torch.cat(res, dim=0)
    # every batch writes its slices of the pass's volume itself: no concatenation pass behind the generator calls
    full = torch.empty((hi - lo, gen.high, gen.high), dtype=torch.float32, device=xs.device)
    if lanes <= 1:
        for j in range(lo, hi, batch):
            k = min(j + batch, hi)
            gen(xs[j:k], ys[j:k] if ys is not None else None, out=full[j - lo:k - lo])
        return full
    gens = [gen] + gen.clones(lanes - 1)
    cur = torch.cuda.current_stream()
    streams = [_lane_stream(i) for i in range(lanes)]
    for st in streams:
        st.wait_stream(cur)
    for bi, j in enumerate(range(lo, hi, batch)):
        k = min(j + batch, hi)
        with torch.cuda.stream(streams[bi % lanes]):
            gens[bi % lanes](xs[j:k], ys[j:k] if ys is not None else None, out=full[j - lo:k - lo])
    for st in streams:
        cur.wait_stream(st)
    return full


# ----------------------------------------------------------------------------
# 4x: two networks chained (example_run_output.py:4-8)
# ----------------------------------------------------------------------------
class _Now(object):
    """an exchange that already happened (single rank)"""

    def __init__(self, full):
        self.full = full

    def wait(self):
        return self.full


def _start_gather(comm, local, total):
    if hasattr(comm, "all_gather_slabs_start"):
        return comm.all_gather_slabs_start(local, total)
    return _Now(comm.all_gather_slabs(local, total))


def _scaled_velocities(low, up_res, vel_scale, backend):
    """the three velocity channels of the low-res array times the upres factor (4x.py:278), then vy, vz times the velocity
    scale (4x.py:283 indexes the 3-channel array with 1:3) -- one marshalling kernel on the call's own stream instead of a
    slice, a multiply and an in-place multiply of the tensor library"""
    s2 = None if vel_scale == 1.0 else [1.0, vel_scale, vel_scale]
    return backend.channel_gather(low, None, [1, 2, 3], [float(up_res)] * 3, s2)


def _pass1_4x(gen1, low, up_res, batch, comm, backend, vel_scale, a2a=False):
    """pass 1 of one volume: upsamplingMode 2 -- zoom z, slices along z (4x.py:1103,1126-1133); the
    hand-over of the slabs to pass 2 is started, not awaited"""
    nch = low.shape[3]
    s = low.shape[0] * up_res
    low1 = low
    if nch > 1 and vel_scale != 1.0:                                 # 4x.py:283, first run: vx,vy,vz
        low1 = backend.channel_gather(low, None, list(range(nch)), [1.0] + [vel_scale] * 3 + [1.0] * (nch - 4))
    xs = backend.axis_zoom_linear(low1, 0, up_res)                   # [s, sim, sim, C]
    lo, hi = slice_range(s, comm)
    out1 = _run_pass(gen1, xs, None, lo, hi, batch)                  # [hi-lo, s, s] = (z, y, x)
    out1 = backend.cutoff(out1, CUTOFF)                              # 4x.py:1156-1157
    return _start_hand_over(comm, out1, s, (2, 0, 1), a2a)


def _pass2_4x(gen2, low, hand, up_res, batch, comm, backend, vel_scale):
    """pass 2: upsamplingMode 1 -- slices along x of (z, y) planes (4x.py:1113-1119).  `hand`: the started hand-over of
    pass 1's volume (_start_hand_over with perm (2, 0, 1)).  Returns (started all-gather of the output, v1 or None)."""
    nch = low.shape[3]
    s = low.shape[0] * up_res
    lo, hi = slice_range(s, comm)
    ys, v1 = _finish_hand_over(hand, backend)                        # [hi-lo][z][y]: this rank's x planes of pass 1
    if nch > 1:
        vel = _scaled_velocities(low, up_res, vel_scale, backend)
        for ax in range(3):                                          # 4x.py:1095
            vel = backend.axis_zoom_linear(vel, ax, up_res)
        # transpose(0,3,1,2,4) then the two channel swaps (d,vx,vy,vz) -> (d,vy,vz,vx); only this rank's x range
        velx = backend.volume_transpose(vel[:, :, lo:hi].contiguous(), (2, 0, 1), chan_map=[1, 2, 0])
        xin = backend.channel_gather(ys.reshape(hi - lo, s, s, 1), velx, [0, 1, 2, 3])
    else:
        xin = ys.reshape(hi - lo, s, s, 1)
    out2 = _run_pass(gen2, xin, None, 0, hi - lo, batch)             # [x-range][z][y]
    return _start_gather(comm, out2, s), v1


def refine_pass_4x(gen, low, prev, up_res=4, mode=1, batch=8, comm=None, backend=ops, vel_scale=1.0, apply_cutoff=True):
    """One refining invocation of multipassGAN-4x.py (upsamplingMode 1: planes (z,y) along x, :1113-1119,1139-1142;
    upsamplingMode 3: planes (z,x) along y, :1121-1124,1144).  prev: the [z,y,x] volume of the previous network.
    Returns the [z,y,x] volume the reference writes (cutoff :1156-1157)."""
    comm = comm or LocalComm()
    nch = low.shape[3]
    s = low.shape[0] * up_res
    lo, hi = slice_range(s, comm)
    # mode 1: transpose(0,3,1,2,4) + swaps 2<->3, 3<->1; mode 3: transpose(0,2,1,3,4) + swap 2<->3
    perm, cmap, back = ((2, 0, 1), [0, 2, 3, 1], (1, 2, 0)) if mode == 1 else ((1, 0, 2), [0, 1, 3, 2], (1, 0, 2))
    if nch > 1:
        vel = _scaled_velocities(low, up_res, vel_scale, backend)
        for ax in range(3):                                          # 4x.py:1095
            vel = backend.axis_zoom_linear(vel, ax, up_res)
        xin = backend.volume_transpose(backend.channel_gather(prev.reshape(s, s, s, 1), vel, [0, 1, 2, 3]), perm, chan_map=cmap)
    else:
        xin = backend.volume_transpose(prev.reshape(s, s, s), perm).reshape(s, s, s, 1)
    out = _run_pass(gen, xin, None, lo, hi, batch)
    vol = comm.all_gather_slabs(out, s)
    return backend.volume_transpose(vol, back, cutoff=CUTOFF if apply_cutoff else 0.0)


def two_pass_4x(gen1, gen2, low, up_res=4, batch=8, comm=None, backend=ops, vel_scale=1.0, exchange="all_gather"):
    """low: device [z,y,x,C].  Returns (final [z,y,x], pass-1 volume [z,y,x]), both with the
    <5e-4 cutoff of the files the reference writes between and after the passes.  exchange: how pass 1's slabs reach
    pass 2 on several ranks ("all_gather": the whole volume to every rank; "all_to_all": only the blocks each rank's
    planes need -- the pass-1 volume is then never assembled and the second result is None)."""
    comm = comm or LocalComm()
    hand = _pass1_4x(gen1, low, up_res, batch, comm, backend, vel_scale, _check_exchange(exchange))
    g2, v1 = _pass2_4x(gen2, low, hand, up_res, batch, comm, backend, vel_scale)
    final = backend.volume_transpose(g2.wait(), (1, 2, 0), cutoff=CUTOFF)   # [x, z, y] -> [z, y, x]: 4x.py:1142,1156
    return final, v1


def _two_pass_4x_steps(gen1, gen2, lows, finals, where, up_res, batch, comm, backend, vel_scale, a2a=False):
    """generator over the software pipeline of two_pass_4x_batch: step i issues pass 1 of volume i, pass 2 of volume
    i-1 and the final transpose of volume i-2; finals[where[k]] receives volume k"""
    n = len(lows)
    g1, g2 = [None] * n, [None] * n
    for i in range(n + 2):
        if i < n:
            g1[i] = _pass1_4x(gen1, lows[i], up_res, batch, comm, backend, vel_scale, a2a)
        j = i - 1
        if 0 <= j < n:
            g2[j], _ = _pass2_4x(gen2, lows[j], g1[j], up_res, batch, comm, backend, vel_scale)
            g1[j] = None
        k = i - 2
        if 0 <= k < n:
            finals[where[k]] = backend.volume_transpose(g2[k].wait(), (1, 2, 0), cutoff=CUTOFF)
            g2[k] = None
        yield


def two_pass_4x_batch(gen1, gen2, lows, up_res=4, batch=8, comm=None, backend=ops, vel_scale=1.0, lanes=None,
                      exchange="all_gather"):
    """The same two passes over a list of independent volumes, software-pipelined by one volume so that the
    slab exchange of volume i (RCCL all-gather on its own stream) runs under pass 1 of volume i+1 and
    pass 2 of volume i-1.  Same results as calling two_pass_4x per volume.  Returns the final volumes.

    lanes: further (gen1, gen2) pairs (`Generator.clone()`: same weights, own workspaces).  The volumes are then dealt
    round-robin to 1 + len(lanes) lanes, each issuing its pipeline on its own HIP stream, so that the launch tails and
    the latency-bound small layers of one lane run under the matrix-bound layers of another."""
    comm = comm or LocalComm()
    a2a = _check_exchange(exchange)
    n = len(lows)
    finals = [None] * n
    pairs = [(gen1, gen2)] + list(lanes or [])
    if len(pairs) == 1 or n < 2 or not lows[0].is_cuda:
        for _ in _two_pass_4x_steps(gen1, gen2, lows, finals, list(range(n)), up_res, batch, comm, backend, vel_scale, a2a):
            pass
        return finals
    cur = torch.cuda.current_stream()
    its = []
    for li, (ga, gb) in enumerate(pairs):
        idx = list(range(li, n, len(pairs)))
        if not idx:
            continue
        st = _lane_stream(li)
        st.wait_stream(cur)
        its.append((st, _two_pass_4x_steps(ga, gb, [lows[i] for i in idx], finals, idx, up_res, batch, comm, backend,
                                           vel_scale, a2a)))
    live = list(its)
    _NESTED[0] = True
    try:
        while live:                               # one pipeline step per lane in turn: every stream stays fed
            for item in list(live):
                st, it = item
                with torch.cuda.stream(st):
                    try:
                        next(it)
                    except StopIteration:
                        live.remove(item)
    finally:
        _NESTED[0] = False
    for st, _ in its:
        cur.wait_stream(st)
    for f in finals:
        f.record_stream(cur)                      # allocated on a lane's stream, consumed on the caller's
    return finals


_LANE_STREAMS = {}


def _lane_stream(i):
    dev = torch.cuda.current_device()
    if (dev, i) not in _LANE_STREAMS:
        _LANE_STREAMS[(dev, i)] = torch.cuda.Stream(device=dev)
    return _LANE_STREAMS[(dev, i)]


# ----------------------------------------------------------------------------
# 8x: up to three networks in one process (multipassGAN-out.py:390-618)
# ----------------------------------------------------------------------------
# Low-res slice batch of pass p under `transposeAxis` t (multipassGAN-out.py:397-421, 463-485, 525-547):
# (axis zoomed to the high resolution, axes order of the [z,y,x] volume, output channel k <- input channel map[k]).
# The velocity component along the slicing axis trades places with vz; transposeAxis 3 rotates all three.
_SWAP_YZ, _SWAP_XZ, _ROT3 = [0, 1, 3, 2], [0, 3, 2, 1], [0, 2, 3, 1]
SLICE_PREP = {
    1: {0: (0, None, None), 1: (1, (1, 0, 2), _SWAP_YZ), 2: (2, (2, 1, 0), _SWAP_XZ), 3: (2, (2, 0, 1), _ROT3)},
    2: {0: (2, (2, 1, 0), _SWAP_XZ), 1: (2, (2, 0, 1), _ROT3), 2: (0, None, None), 3: (1, (1, 0, 2), _SWAP_YZ)},
    # pass 3: transposeAxis 2 reads channel 13 of a 4-channel batch in the reference (:542) -> IndexError there too;
    # transposeAxis 3 regroups a [z, X, y] array into sim x sim planes without moving X to the front (:534-535)
    3: {0: (1, (1, 0, 2), _SWAP_YZ), 1: (0, None, None), 3: (2, (0, 2, 1), [0, 2, 1, 3])},
}


def slice_batch_8x(low, up_res, pass_no, transpose_axis, backend=ops):
    """[S, sim, sim, C] low-res slice batch of one pass (before the add_adj_idcs channels)"""
    sim, nch = low.shape[0], low.shape[3]
    try:
        axis, perm, cmap = SLICE_PREP[pass_no][transpose_axis]
    except KeyError:
        if pass_no == 3 and transpose_axis == 2:
            raise IndexError("index 13 is out of bounds for axis 3 with size %d (multipassGAN-out.py:542: the third "
                             "pass of transposeAxis 2 is broken in the reference as well)" % nch)
        raise ValueError("transposeAxis %r (0..3)" % (transpose_axis,))
    xs = backend.axis_zoom_linear(low, axis, up_res)
    if perm is not None:
        xs = backend.volume_transpose(xs, perm, chan_map=cmap if nch >= 4 else None)
    return xs.reshape(-1, sim, sim, nch)


def multipass_8x(gens, low, up_res=8, batches=(8, 2, 2), comm=None, backend=ops, apply_cutoff=True, transpose_axis=0,
                 exchange="all_gather"):
    """gens: 1..3 Generator objects (first one firstGen).  low: device [z,y,x,4] with velocities
    already scaled by velScale (out.py:138).  Returns the density volume after the reference's final
    axis restore (:587-590): [z,y,x] for transposeAxis 0."""
    comm = comm or LocalComm()
    a2a = _check_exchange(exchange)
    sim = low.shape[0]
    s = sim * up_res
    lo, hi = slice_range(s, comm)
    # pass 1 (397-461): slices along the zoomed axis, add_adj_idcs channels
    xs = slice_batch_8x(low, up_res, 1, transpose_axis, backend)
    if gens[0].cfg.get("add_adj", False):
        xs_r = backend.add_adjacent(xs, lo, hi - lo)
        out = _run_pass(gens[0], xs_r, None, 0, hi - lo, batches[0])
    else:
        out = _run_pass(gens[0], xs, None, lo, hi, batches[0])
    if len(gens) > 1:
        # pass 2 (463-523): conditioned on planes of dim_output.transpose(2,1,0) (:459)
        ys, _ = _hand_over(comm, out, s, (2, 1, 0), a2a, backend)
        xl = slice_batch_8x(low, up_res, 2, transpose_axis, backend)
        out = _run_pass(gens[1], xl[lo:hi], ys, 0, hi - lo, batches[1])
    if len(gens) > 2:
        # pass 3 (525-585): conditioned on the previous result .transpose(1,2,0) (:521)
        ys, _ = _hand_over(comm, out, s, (1, 2, 0), a2a, backend)
        xl = slice_batch_8x(low, up_res, 3, transpose_axis, backend)
        out = _run_pass(gens[2], xl[lo:hi], ys, 0, hi - lo, batches[2])
    vol = comm.all_gather_slabs(out, s)
    # the transposes the reference applies to its last dim_output, composed (459 / 521 / 583, then 587-590)
    perm = {1: (0, 1, 2), 2: (2, 1, 0), 3: (1, 0, 2)}[len(gens)]
    thr = CUTOFF if apply_cutoff else 0.0
    if perm == (0, 1, 2):
        return backend.cutoff(vol, CUTOFF) if apply_cutoff else vol
    return backend.volume_transpose(vol, perm, cutoff=thr)


# ----------------------------------------------------------------------------
# 8x: one network per process, volumes travel through .uni files (multipassGAN-8x.py:1600-1780;
# the commented alternative of example_run_output.py:64-70: modes 2 -> 1 -> 3)
# ----------------------------------------------------------------------------
# (axes order applied to the slice batch and to the previous volume, channel map, order that restores [z,y,x])
_SINGLE_PASS = {0: (None, None, (0, 1, 2)), 1: ((1, 0, 2), _SWAP_YZ, (1, 0, 2)), 2: ((2, 1, 0), _SWAP_XZ, (2, 1, 0)),
                3: ((2, 0, 1), _ROT3, (1, 2, 0))}
_ZOOM_AXIS = {0: 0, 1: 1, 2: 2, 3: 2}


def single_pass_8x(gen, low, prev=None, up_res=8, transpose_axis=0, batch=2, comm=None, backend=ops, apply_cutoff=True):
    """One `multipassGAN-8x.py out 1` invocation.  low: device [z,y,x,C]; prev: None for the first network
    (upsamplingMode 2) or the previous network's [z,y,x] volume as stored in its .uni file (modes 1 / 3).
    Returns the volume the reference writes: [z,y,x] with the <5e-4 cutoff (:1720-1725, 1766-1767)."""
    comm = comm or LocalComm()
    sim, nch = low.shape[0], low.shape[3]
    s = sim * up_res
    lo, hi = slice_range(s, comm)
    perm, cmap, back = _SINGLE_PASS[transpose_axis]
    xs = backend.axis_zoom_linear(low, _ZOOM_AXIS[transpose_axis], up_res)           # :1606-1613, 1624-1631
    if perm is not None:
        xs = backend.volume_transpose(xs, perm, chan_map=cmap if nch >= 4 else None)  # :1644-1664
    xs = xs.reshape(-1, sim, sim, nch)
    ys = None
    if prev is not None:
        ys = prev.reshape(s, s, s)
        if perm is not None:
            ys = backend.volume_transpose(ys, perm)                                  # :1650,1657,1666
    if gen.cfg.get("add_adj", False):
        out = _run_pass(gen, backend.add_adjacent(xs, lo, hi - lo), ys[lo:hi] if ys is not None else None, 0, hi - lo, batch)
    else:
        out = _run_pass(gen, xs, ys, lo, hi, batch)
    vol = comm.all_gather_slabs(out, s)
    thr = CUTOFF if apply_cutoff else 0.0
    if back == (0, 1, 2):
        return backend.cutoff(vol, CUTOFF) if apply_cutoff else vol
    return backend.volume_transpose(vol, back, cutoff=thr)
