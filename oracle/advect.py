"""numpy restatement of ``GAN.advect`` (tools_wscale/GAN.py:173-418): semi-Lagrangian and MacCormack advection of
a 2D field by a (MAC-grid) velocity, as the temporal-coherence branch of multipassGAN-8x.py uses it
(:1199,1225: ``GAN(x).advect(x, vel_t, flags, 0.5, ADV_mode, 1.0, startBz=(batch // 3) * 3)``).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  "Parity unpinned" like every TensorFlow-computed op: the
reference holds no vectors for it and TF 1.x is not installable here.  ``semi_lagrange_loop`` is a scalar-loop
restatement of the same definition that the vectorised one (and the HIP kernel) are checked against.

Conventions the reference fixes (2D branch, square grids):
* positions: ``pos[n,i,j] = (i + 1.0, j + 1.0)`` -- cell centres 0.5.. plus a further +0.5 (:372-374), so that with zero
  velocity every output is the mean of the 2x2 cells (i..i+1, j..j+1);
* velocity: channels (x, y) of ``vel`` reordered to (y, x), resized with the legacy bilinear ``resize_images`` to the
  source resolution, multiplied by the resolution ratio, averaged with its successor along its own axis (MAC ->
  centre; the successor of the last row / column is zero, :382-384), times dt * (+1, 0, -1) over each frame triple;
* look-up: ``p = pos - vel``; indices floor(p - 0.5) and +1, clamped into the grid; weights from the CLAMPED indices,
  so they are not a partition of unity at the border (:178-203).
"""
import numpy as np

from . import ops

F32 = np.float32
THRESHOLD_FLAGS = 0.2
BIG = float(np.float32(9223372036854775807))       # float32(sys.maxsize), GAN.py:222-223


def positions(h, w):
    """:362-374 (square grids; the tile / reshape construction of the reference only forms a mesh when h == w)"""
    assert h == w, "GAN.advect builds its position grid by tiling a range: meaningful for square fields only"
    i, j = np.meshgrid(np.arange(h, dtype=F32), np.arange(w, dtype=F32), indexing="ij")
    return np.stack([i + F32(1.0), j + F32(1.0)], axis=-1)[None]             # [1,h,w,2] = (y, x)


def centred_velocity(vel, h, w, dt):
    """:376-396: [N,hv,wv,>=2] (x,y,..) -> [N,h,w,2] (y,x) displacement per frame of each triple"""
    n = vel.shape[0]
    up = F32(max(h / vel.shape[1], w / vel.shape[2]))
    v = np.stack([vel[..., 1], vel[..., 0]], axis=-1).astype(F32)
    v = ops.resize_bilinear_tf1(v, h, w) * up
    nxt = np.zeros_like(v)
    nxt[:, :-1, :, 0] = v[:, 1:, :, 0]            # tf.contrib.image.transform [1,0,0,0,1,1,0,0]: row i+1, zero past the end
    nxt[:, :, :-1, 1] = v[:, :, 1:, 1]            # [1,0,1,0,1,0,0,0]: column j+1
    v = F32(0.5) * (v + nxt)
    steps = np.array([dt, 0.0, -dt] * (n // 3), dtype=F32)
    assert steps.shape[0] == n, "the batch holds whole frame triples"
    return v * steps.reshape(n, 1, 1, 1)


def semi_lagrange(source, vel, pos):
    """:175-204.  source [N,H,W,C], vel [N,H,W,2] (y,x), pos [1 or N,H,W,2]"""
    n, h, w, _ = source.shape
    p = (pos.astype(F32) - vel).astype(F32)
    q = p - F32(0.5)
    lo = np.floor(q).astype(np.int32)
    hi = lo + 1
    lim = np.array([h - 1, w - 1], dtype=np.int32)
    lo = np.minimum(np.maximum(lo, 0), lim)
    hi = np.minimum(np.maximum(hi, 0), lim)
    out = np.zeros(source.shape, dtype=np.float64)
    b = np.arange(n)[:, None, None]
    for corner in range(4):
        use_hi = np.array([bool(corner & 1), bool(corner & 2)])
        idx = np.where(use_hi, hi, lo)
        wgt = np.prod(F32(1.0) - np.abs(q - idx.astype(F32)), axis=-1, keepdims=True, dtype=F32)
        out += source[b, idx[..., 0], idx[..., 1], :].astype(np.float64) * wgt
    return out.astype(F32)


def semi_lagrange_loop(source, vel, pos):
    """the same definition, one output element at a time"""
    n, h, w, c = source.shape
    out = np.zeros(source.shape, dtype=np.float64)
    for b in range(n):
        for i in range(h):
            for j in range(w):
                pb = pos[b if pos.shape[0] > 1 else 0, i, j]
                qy = F32(F32(pb[0] - vel[b, i, j, 0]) - F32(0.5))
                qx = F32(F32(pb[1] - vel[b, i, j, 1]) - F32(0.5))
                y0, x0 = int(np.floor(qy)), int(np.floor(qx))
                ys = [min(max(y0, 0), h - 1), min(max(y0 + 1, 0), h - 1)]
                xs = [min(max(x0, 0), w - 1), min(max(x0 + 1, 0), w - 1)]
                for cy in (0, 1):
                    for cx in (0, 1):
                        wy = F32(1.0) - abs(F32(qy - F32(ys[cy])))
                        wx = F32(1.0) - abs(F32(qx - F32(xs[cx])))
                        out[b, i, j] += source[b, ys[cy], xs[cx]].astype(np.float64) * F32(wy * wx)
    return out.astype(F32)


def maccormack_correct(flags, source, forward, backward, strength=1.0):
    """:206-209"""
    cond = flags.reshape(source.shape[0], source.shape[1], source.shape[2], 1) < F32(THRESHOLD_FLAGS)
    return np.where(cond, forward + F32(strength) * F32(0.5) * (source - backward), forward).astype(F32)


def maccormack_clamp(flags, vel, intermed, source, forward, pos, start_bz):
    """doClampComponent (:213-343): the corrected value is kept only inside the [min, max] of the (fluid) cells around the
    truncated look-up position; channel 0 of `source` / `flags` is what the reference gathers."""
    assert start_bz == source.shape[0], "the reference's tf.where needs startBz == batch size"
    return np.where(_rejected(flags, vel, intermed, source, pos), forward, intermed).astype(F32)


def advect(source, vel, flags, dt, order, strength=0.0, start_bz=15):
    """GAN.advect (:347-418).  source [N,H,W,C]; vel [N,hv,wv,>=2]; flags [N,H,W,1] (order 2 only)"""
    source = np.asarray(source, dtype=F32)
    n, h, w, _ = source.shape
    pos = positions(h, w)
    v = centred_velocity(np.asarray(vel, dtype=F32), h, w, dt)
    forward = semi_lagrange(source, v, pos)
    if order != 2:
        return forward
    backward = semi_lagrange(forward, -v, pos)
    flags = np.asarray(flags, dtype=F32).reshape(n, h, w, 1)
    corrected = maccormack_correct(flags, source, forward, backward, strength)
    return maccormack_clamp(flags, v, corrected, source, forward, pos, start_bz)


def advect_torch(source, vel, flags, dt, order, strength=0.0, start_bz=15):
    """GAN.advect as a differentiable float64 torch expression in `source` (a torch tensor [N,H,W,C]); velocities and
    flags are data.  Index / weight arithmetic in float32 exactly as in `semi_lagrange`; the gathers, the correction and
    the tf.where selections are torch ops, so autograd yields what TensorFlow's gradient of the op graph yields
    (selections pass the gradient of the chosen branch; comparisons carry none)."""
    import torch
    n, h, w, c = source.shape
    pos = positions(h, w)
    v = centred_velocity(np.asarray(vel, dtype=F32), h, w, dt)

    def sl(src, vv):
        p = (pos.astype(F32) - vv).astype(F32)
        q = p - F32(0.5)
        lo = np.floor(q).astype(np.int64)
        hi = lo + 1
        lim = np.array([h - 1, w - 1], dtype=np.int64)
        lo = np.minimum(np.maximum(lo, 0), lim)
        hi = np.minimum(np.maximum(hi, 0), lim)
        b = torch.arange(n).reshape(n, 1, 1)
        out = 0
        for corner in range(4):
            use_hi = np.array([bool(corner & 1), bool(corner & 2)])
            idx = np.where(use_hi, hi, lo)
            wgt = np.prod(F32(1.0) - np.abs(q - idx.astype(F32)), axis=-1, keepdims=True, dtype=F32)
            out = out + src[b, torch.as_tensor(idx[..., 0]), torch.as_tensor(idx[..., 1]), :] * torch.as_tensor(wgt.astype(np.float64))
        return out

    forward = sl(source, v)
    if order != 2:
        return forward
    backward = sl(forward, -v)
    flags = np.asarray(flags, dtype=F32).reshape(n, h, w, 1)
    fluid = torch.as_tensor(flags < F32(THRESHOLD_FLAGS))
    corrected = torch.where(fluid, forward + strength * 0.5 * (source - backward), forward)
    # the clamp decides on float32 values, as the graph does
    f32 = lambda t: t.detach().numpy().astype(F32)                    # noqa: E731
    assert start_bz == n, "the reference's tf.where needs startBz == batch size"
    reject = torch.as_tensor(_rejected(flags, v, f32(corrected), f32(source), pos))
    return torch.where(reject, forward, corrected)


def _rejected(flags, vel, intermed, source, pos):
    """cond_complete of doClampComponent (:269-343)"""
    n, h, w, _ = source.shape
    cur = np.trunc(np.broadcast_to(pos, vel.shape).astype(F32) - vel).astype(np.int32)
    i0 = np.clip(cur[..., 0], 0, h - 1)
    j0 = np.clip(cur[..., 1], 0, w - 1)
    b = np.broadcast_to(np.arange(n)[:, None, None], i0.shape)
    lo = np.full(source.shape[:3] + (1,), BIG, dtype=F32)
    hi = np.full(source.shape[:3] + (1,), -BIG - 1, dtype=F32)
    lo_i, hi_i = lo.copy(), hi.copy()
    top = w - 1                                         # every index component is clipped to grid_res[2] - 1 (:270-272)
    for di, dj in ((0, 0), (1, 0), (0, 1), (1, 1)):
        if (di, dj) == (0, 0):
            bb, ii, jj = b, i0, j0                      # indices_0 is not clipped again (:269)
        else:
            bb, ii, jj = np.clip(b, 0, top), np.clip(i0 + di, 0, top), np.clip(j0 + dj, 0, top)
        s = source[bb, ii, jj, 0][..., None]
        fl = flags[bb, ii, jj, 0][..., None] < F32(THRESHOLD_FLAGS)
        lo = np.where(fl, np.minimum(lo, s), lo)
        hi = np.where(fl, np.maximum(hi, s), hi)
    return (intermed < lo) | (intermed > hi) | (lo == lo_i) | (hi == hi_i)
