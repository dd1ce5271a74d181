"""numpy restatement of the TF 1.x ops on the reference's hot path (NHWC, float32).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  All citations are
relative to ``/root/reference``.  Arithmetic is carried in float64 and rounded
to float32 at each op boundary, so that the oracle sits closer to the exact
result than either TF's fp32 kernels or the HIP path.
"""
import math

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------
# SAME padding (tf.nn.conv2d(..., padding="SAME"), tools_wscale/GAN.py:686-691)
# ----------------------------------------------------------------------------
def same_pad(n, k, s):
    """TF SAME: out = ceil(n/s); total pad = max((out-1)*s + k - n, 0); the
    extra element goes to the bottom/right."""
    out = -(-n // s)
    pad = max((out - 1) * s + k - n, 0)
    return out, pad // 2, pad - pad // 2


def conv2d_same(x, w, stride=(1, 1)):
    """``tf.nn.conv2d(x, W, strides, "SAME")`` (GAN.py:686-691).

    x: [N,H,W,Cin] float32, w: [kh,kw,Cin,Cout] (HWIO, GAN.py:93).
    Accumulates tap by tap in float64.
    """
    x = np.asarray(x)
    w = np.asarray(w)
    n, h, wd, cin = x.shape
    kh, kw, cin2, cout = w.shape
    assert cin == cin2, (x.shape, w.shape)
    sh, sw = stride
    oh, pt, pb = same_pad(h, kh, sh)
    ow, pl, pr = same_pad(wd, kw, sw)
    xp = np.zeros((n, h + pt + pb, wd + pl + pr, cin), dtype=np.float64)
    xp[:, pt:pt + h, pl:pl + wd, :] = x
    out = np.zeros((n, oh, ow, cout), dtype=np.float64)
    w64 = w.astype(np.float64)
    for dy in range(kh):
        for dx in range(kw):
            patch = xp[:, dy:dy + (oh - 1) * sh + 1:sh, dx:dx + (ow - 1) * sw + 1:sw, :]
            out += patch.reshape(-1, cin).dot(w64[dy, dx]).reshape(n, oh, ow, cout)
    return out.astype(F32)


def conv2d_transpose_same(x, w, stride=(1, 1)):
    """``tf.nn.conv2d_transpose(x, W, output_shape=[N, H*sh, W*sw, Cout], strides, "SAME")`` (GAN.py:703-708).

    x: [N,H,W,Cin]; w: [kh,kw,Cout,Cin] (TensorFlow's transposed-filter layout).  Written from the definition -- the
    gradient of ``conv2d_same`` on the OUTPUT grid: every input pixel scatters x * W into the padded output, the SAME
    padding of the forward convolution (``same_pad`` on the output size) is cropped.  float64 accumulation."""
    x = np.asarray(x)
    w = np.asarray(w)
    n, h, wd, cin = x.shape
    kh, kw, cout, cin2 = w.shape
    assert cin == cin2, (x.shape, w.shape)
    sh, sw = stride
    oh, ow = h * sh, wd * sw
    fo_h, pt, _ = same_pad(oh, kh, sh)
    fo_w, pl, _ = same_pad(ow, kw, sw)
    assert (fo_h, fo_w) == (h, wd)
    full = np.zeros((n, (h - 1) * sh + kh + sh, (wd - 1) * sw + kw + sw, cout), dtype=np.float64)
    w64 = w.astype(np.float64)
    x64 = x.astype(np.float64)
    for ky in range(kh):
        for kx in range(kw):
            contrib = x64.reshape(-1, cin).dot(w64[ky, kx].T).reshape(n, h, wd, cout)
            full[:, ky:ky + (h - 1) * sh + 1:sh, kx:kx + (wd - 1) * sw + 1:sw, :] += contrib
    return full[:, pt:pt + oh, pl:pl + ow, :].astype(F32)


def wscale(shape, gain=math.sqrt(2.0)):
    """Equalised-LR constant of ``GAN.weight_variable`` (GAN.py:661-668):
    float32(gain / sqrt(prod(shape[:-1])))."""
    return F32(gain / np.sqrt(np.prod(shape[:-1])))


def bias_add(x, b):
    return (x.astype(np.float64) + np.asarray(b, dtype=np.float64)).astype(F32)


def batch_norm_infer(x, gamma, beta, mean, var, eps=1e-3):
    """``tf.contrib.layers.batch_norm(is_training=False, scale=True)`` with the
    TF default epsilon 0.001 (GAN.py:108-110)."""
    x64 = x.astype(np.float64)
    inv = np.asarray(gamma, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps)
    return ((x64 - np.asarray(mean, np.float64)) * inv + np.asarray(beta, np.float64)).astype(F32)


def batch_norm_train(x, gamma, beta, eps=1e-3):
    """Training-mode batch norm: biased batch moments over N,H,W (GAN.py:110)."""
    x64 = x.astype(np.float64)
    mean = x64.mean(axis=(0, 1, 2))
    var = x64.var(axis=(0, 1, 2))
    y = (x64 - mean) / np.sqrt(var + eps) * np.asarray(gamma, np.float64) + np.asarray(beta, np.float64)
    return y.astype(F32), mean.astype(F32), var.astype(F32)


def relu(x):
    return np.maximum(x, F32(0))


def lrelu(x, leak=0.2):
    """GAN.py:733-737: f1*x + f2*|x| with f1 = 0.5(1+leak), f2 = 0.5(1-leak)."""
    f1 = 0.5 * (1 + leak)
    f2 = 0.5 * (1 - leak)
    x64 = x.astype(np.float64)
    return (f1 * x64 + f2 * np.abs(x64)).astype(F32)


def pixel_norm(x, epsilon=1e-8):
    """GAN.py:472-474: x * rsqrt(mean_c(x^2) + eps)."""
    x64 = x.astype(np.float64)
    return (x64 / np.sqrt(np.mean(x64 * x64, axis=3, keepdims=True) + epsilon)).astype(F32)


def activation(x, act):
    if act is None or act == "none":
        return x
    if act == "relu":
        return relu(x)
    if act == "lrelu":
        return lrelu(x)
    if act == "tanh":
        return np.tanh(x.astype(np.float64)).astype(F32)
    raise ValueError(act)


# ----------------------------------------------------------------------------
# legacy TF1 resize ops (align_corners=False, no half-pixel centres)
# ----------------------------------------------------------------------------
def _nearest_index(out_size, in_size):
    scale = F32(in_size) / F32(out_size)
    idx = np.floor(np.arange(out_size, dtype=F32) * scale).astype(np.int64)
    return np.minimum(idx, in_size - 1)


def resize_nearest_tf1(x, oh, ow):
    """``tf.image.resize_images(x, [oh,ow], method=1)`` / ``kb.resize_images``
    (GAN.py:517,541; multipassGAN-out.py:357): src = min(floor(dst*in/out), in-1)."""
    iy = _nearest_index(oh, x.shape[1])
    ix = _nearest_index(ow, x.shape[2])
    return x[:, iy][:, :, ix]


def resize_bilinear_tf1(x, oh, ow):
    """``tf.image.resize_images(..., method=0)`` legacy bilinear (GAN.py:541)."""
    n, h, w, c = x.shape

    def axis(out_size, in_size):
        scale = F32(in_size) / F32(out_size)
        s = np.arange(out_size, dtype=F32) * scale
        lo = np.floor(s).astype(np.int64)
        hi = np.minimum(lo + 1, in_size - 1)
        return lo, hi, (s - lo).astype(np.float64)

    y0, y1, fy = axis(oh, h)
    x0, x1, fx = axis(ow, w)
    x64 = x.astype(np.float64)
    top = x64[:, y0][:, :, x0] + (x64[:, y0][:, :, x1] - x64[:, y0][:, :, x0]) * fx[None, None, :, None]
    bot = x64[:, y1][:, :, x0] + (x64[:, y1][:, :, x1] - x64[:, y1][:, :, x0]) * fx[None, None, :, None]
    return (top + (bot - top) * fy[None, :, None, None]).astype(F32)


_BICUBIC_TABLE = 1024
_BICUBIC_A = -0.75


def _bicubic_lut():
    # TF 1.x resize_bicubic_op.cc InitCoeffsTable: float32 table of 2*(1024+1) entries
    tab = np.zeros((_BICUBIC_TABLE + 1) * 2, dtype=F32)
    a = F32(_BICUBIC_A)
    for i in range(_BICUBIC_TABLE + 1):
        x = F32(i) / F32(_BICUBIC_TABLE)
        tab[i * 2] = ((a + F32(2)) * x - (a + F32(3))) * x * x + F32(1)
        x = x + F32(1)
        tab[i * 2 + 1] = ((a * x - F32(5) * a) * x + F32(8) * a) * x - F32(4) * a
    return tab


def bicubic_taps_tf1(out_size, in_size):
    """Indices [out,4] and float32 weights [out,4] of TF 1.x ``ResizeBicubic``
    (Keys kernel A=-0.75, src = dst*in/out, fraction quantised to 1/1024,
    taps clamped to the edge).  Restated from the TF 1.x kernel definition;
    TF itself is unavailable here ("parity unpinned")."""
    lut = _bicubic_lut()
    scale = F32(in_size) / F32(out_size)
    idx = np.zeros((out_size, 4), dtype=np.int64)
    wts = np.zeros((out_size, 4), dtype=F32)
    for o in range(out_size):
        s = F32(o) * scale
        i = int(math.floor(s))
        delta = F32(s - F32(i))
        off = int(np.rint(delta * F32(_BICUBIC_TABLE)))
        wts[o] = (lut[off * 2 + 1], lut[off * 2], lut[(_BICUBIC_TABLE - off) * 2], lut[(_BICUBIC_TABLE - off) * 2 + 1])
        idx[o] = [min(max(i + d, 0), in_size - 1) for d in (-1, 0, 1, 2)]
    return idx, wts


def resize_bicubic_tf1(x, oh, ow):
    """``tf.image.resize_images(..., method=2)`` as used by
    ``avg_depool(mode=2)`` for ``addBicubicUpsample`` (multipassGAN-out.py:330)."""
    iy, wy = bicubic_taps_tf1(oh, x.shape[1])
    ix, wx = bicubic_taps_tf1(ow, x.shape[2])
    x64 = x.astype(np.float64)
    # along W for every input row, then along H (the TF kernel's order)
    tmp = np.zeros((x.shape[0], x.shape[1], ow, x.shape[3]), dtype=np.float64)
    for t in range(4):
        tmp += x64[:, :, ix[:, t], :] * wx[:, t].astype(np.float64)[None, None, :, None]
    out = np.zeros((x.shape[0], oh, ow, x.shape[3]), dtype=np.float64)
    for t in range(4):
        out += tmp[:, iy[:, t], :, :] * wy[:, t].astype(np.float64)[None, :, None, None]
    return out.astype(F32)


def resize_images_tf1(x, oh, ow, method):
    """tf.image.resize_images method ids: 0 bilinear, 1 nearest, 2 bicubic."""
    if method == 0:
        return resize_bilinear_tf1(x, oh, ow)
    if method == 1:
        return resize_nearest_tf1(x, oh, ow)
    if method == 2:
        return resize_bicubic_tf1(x, oh, ow)
    raise ValueError("resize method %r" % (method,))


def max_depool(x, height_factor=2, width_factor=2):
    """``GAN.max_depool`` -> ``kb.resize_images`` nearest replication (GAN.py:501-523)."""
    return resize_nearest_tf1(x, x.shape[1] * height_factor, x.shape[2] * width_factor)


def avg_depool(x, mode=0, scale=(2,)):
    """``GAN.avg_depool`` 2D branch (GAN.py:528-541)."""
    if len(scale) == 1:
        oh, ow = x.shape[1] * scale[0], x.shape[2] * scale[0]
    else:
        oh, ow = x.shape[1] * scale[0], x.shape[2] * scale[1]
    return resize_images_tf1(x, int(oh), int(ow), mode)


def avg_pool(x, k=2, s=2):
    """``tf.nn.avg_pool(..., VALID)`` (GAN.py:162-169)."""
    n, h, w, c = x.shape
    oh, ow = (h - k) // s + 1, (w - k) // s + 1
    acc = np.zeros((n, oh, ow, c), dtype=np.float64)
    for dy in range(k):
        for dx in range(k):
            acc += x[:, dy:dy + (oh - 1) * s + 1:s, dx:dx + (ow - 1) * s + 1:s, :]
    return (acc / (k * k)).astype(F32)


def max_pool(x, k=2, s=2):
    n, h, w, c = x.shape
    oh, ow = (h - k) // s + 1, (w - k) // s + 1
    acc = np.full((n, oh, ow, c), -np.inf, dtype=F32)
    for dy in range(k):
        for dx in range(k):
            acc = np.maximum(acc, x[:, dy:dy + (oh - 1) * s + 1:s, dx:dx + (ow - 1) * s + 1:s, :])
    return acc


def depth_to_space(x, r):
    """``tf.depth_to_space`` (GAN.py:554-560), NHWC."""
    n, h, w, c = x.shape
    co = c // (r * r)
    y = x.reshape(n, h, w, r, r, co).transpose(0, 1, 3, 2, 4, 5)
    return y.reshape(n, h * r, w * r, co)


def fully_connected(x, w, b):
    """``tf.matmul(flat, W) + b`` (GAN.py:438-456); W is [in,out], already scaled."""
    return (x.astype(np.float64).dot(np.asarray(w, np.float64)) + np.asarray(b, np.float64)).astype(F32)


def minibatch_stddev(x, group_size=4):
    """``GAN.minibatch_stddev_layer`` (GAN.py:476-488) on NHWC data."""
    g = min(group_size, x.shape[0])
    s = x.shape
    y = x.reshape(g, -1, s[1], s[2], s[3]).astype(np.float64)
    y = y - y.mean(axis=0, keepdims=True)
    y = np.sqrt((y * y).mean(axis=0) + 1e-8)
    y = y.mean(axis=(1, 2, 3), keepdims=True)
    y = np.tile(y, (g, s[1], s[2], 1)).astype(F32)
    return np.concatenate([x, y], axis=3)


# ----------------------------------------------------------------------------
# scipy.ndimage.zoom(order=1) along one axis (K16)
# ----------------------------------------------------------------------------
def zoom_axis_linear(v, axis, factor):
    """Closed form of ``scipy.ndimage.zoom(v, [..factor..], order=1,
    mode='constant')`` for one zoomed axis (multipassGAN-out.py:401-421,
    multipassGAN-4x.py:1095-1103): N = round(n*factor), src = o*(n-1)/(N-1)."""
    v = np.asarray(v)
    n = v.shape[axis]
    big = int(round(n * factor))
    if big == n:
        return v.astype(F32, copy=True)
    o = np.arange(big, dtype=np.float64)
    s = o * (n - 1) / (big - 1) if big > 1 else np.zeros(1)
    i0 = np.minimum(np.floor(s).astype(np.int64), n - 1)
    i1 = np.minimum(i0 + 1, n - 1)
    t = s - i0
    shape = [1] * v.ndim
    shape[axis] = big
    t = t.reshape(shape)
    a = np.take(v, i0, axis=axis).astype(np.float64)
    b = np.take(v, i1, axis=axis).astype(np.float64)
    return (a * (1.0 - t) + b * t).astype(F32)
