"""Operator layer: thin, checked wrappers of the C ABI on PyTorch-ROCm tensors.

PyTorch is used for device memory and streams only; every function here
enqueues hand-written HIP kernels from libmpgan_hip.so on torch's current
stream.  Tensors are NHWC float32 on the GPU.
"""
import ctypes

import torch

from . import _lib
from ._lib import G8_F16, PREC_F16F6, PREC_F16X1, PREC_F16X3  # noqa: F401  (re-exported)

DEFAULT_PREC = PREC_F16X3     # kernel-level default (weight packing, generic sessions, training)
# what the inference drivers and multipass.Generator run unless told otherwise (`prec` parameter): held to
# 5e-4 relative L2 of the oracle at the full C2 / C4 sizes by tests/test_fullsize_gpu.py (north_star: 1e-3)
INFERENCE_PREC = PREC_F16F6
# contractions up to this length run as MPG_PREC_F16X3 inside an F16F6 session (session._match_fused)
F16F6_MIN_K = 256


def parse_prec(v):
    """`prec` parameter of the GAN/ drivers: 2 / "f16f6" (default), 3 / "f16x3" (fp32-grade), 1 / "f16x1" """
    names = {"f16f6": PREC_F16F6, "f16x3": PREC_F16X3, "f16x1": PREC_F16X1, "fp32": PREC_F16X3}
    s = str(v).strip().lower()
    if s in names:
        return names[s]
    try:
        p = int(s)
    except ValueError:
        p = -1
    if p not in (PREC_F16X1, PREC_F16F6, PREC_F16X3):
        raise _lib.MpgError("prec %r: expected 1 (f16x1), 2 (f16f6) or 3 (f16x3)" % (v,))
    return p


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name="tensor", dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.MpgError("%s must be a GPU tensor (the HIP path has no CPU fallback)" % name)
    if t.dtype != dtype:
        raise _lib.MpgError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.MpgError("%s must be contiguous" % name)
    return t


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class G8(object):
    """A "G8" activation tensor (include/mpgan.h): [N][ceil(C/8)][2 planes hi, lo][H][W][8 x fp16]."""

    __slots__ = ("buf", "n", "h", "w", "c", "flavour")

    def __init__(self, buf, n, h, w, c, flavour=G8_F16):
        self.buf, self.n, self.h, self.w, self.c, self.flavour = buf, n, h, w, c, flavour

    @property
    def groups(self):
        return (self.c + 7) // 8

    @staticmethod
    def empty(n, h, w, c, device, flavour=G8_F16):
        buf = torch.empty((n, (c + 7) // 8, 2, h, w, 8), dtype=torch.float16, device=device)
        return G8(buf, n, h, w, c, flavour)


def flavour_for(prec):
    """the G8 flavour a launch of precision `prec` reads: every precision reads (hi16, lo16) since round 3"""
    return G8_F16


def f6_available(cout, segments=()):
    """MPG_PREC_F16F6 is built for every output width of the fused convolution (1..128); a segment (kh, kw, cin)
    whose LDS images do not fit at that width (7x7 with four cout tiles) answers 0 to the pack-size query"""
    if not 1 <= cout <= 128:
        return False
    lib = _lib.load()
    return all(lib.mpg_conv_pack_size(kh, kw, cin, cout, PREC_F16F6) > 0 for (kh, kw, cin) in segments)


def absmax(x):
    """device scalar max |x| (mpg_absmax)"""
    lib = _lib.load()
    x = _dev(x.contiguous(), "x")
    out = torch.empty((), dtype=torch.float32, device=x.device)
    _lib.check(lib.mpg_absmax(_stream(), _ptr(x), x.numel(), _ptr(out)), "mpg_absmax")
    return out


def to_g8(x, c_off=0, cin=None, flavour=G8_F16, amax=None):
    """fp32 NHWC x[..., c_off:c_off+cin] -> G8 (mpg_f32_to_g8)"""
    lib = _lib.load()
    x = _dev(x, "x")
    if x.dim() != 4:
        raise _lib.MpgError("to_g8: expected NHWC, got %s" % (tuple(x.shape),))
    n, h, w, c = x.shape
    cin = c - c_off if cin is None else cin
    g = G8.empty(n, h, w, cin, x.device, flavour)
    if x.numel():
        _lib.check(lib.mpg_f32_to_g8_scaled(_stream(), _ptr(x), n, h, w, c, c_off, cin, flavour, _ptr(amax), _ptr(g.buf)),
                   "mpg_f32_to_g8")
    return g


def from_g8(g):
    """G8 -> fp32 NHWC (mpg_g8_to_f32)"""
    lib = _lib.load()
    if g.flavour != G8_F16:
        raise _lib.MpgError("from_g8: only the fp16 hi/lo flavour converts back to fp32")
    y = torch.empty((g.n, g.h, g.w, g.c), dtype=torch.float32, device=g.buf.device)
    _lib.check(lib.mpg_g8_to_f32(_stream(), _ptr(g.buf), g.n, g.h, g.w, g.c, _ptr(y)), "mpg_g8_to_f32")
    return y


class PackedWeights(object):
    """Weights of one conv segment in MFMA fragment order (mpg_conv_pack_weights)."""

    __slots__ = ("buf", "kh", "kw", "cin", "cout", "prec")

    def __init__(self, buf, kh, kw, cin, cout, prec):
        self.buf, self.kh, self.kw, self.cin, self.cout, self.prec = buf, kh, kw, cin, cout, prec


def pack_conv_weights(w_hwio, wscale=1.0, cout_scale=None, c_off=0, cin=None, prec=DEFAULT_PREC):
    """Pack W[kh,kw,cin_total,cout] channels [c_off, c_off+cin) times wscale
    (GAN.weight_variable, GAN.py:664-668) times an optional per-channel scale
    (folded batch norm, GAN.py:110)."""
    lib = _lib.load()
    w = _dev(w_hwio, "w_hwio")
    kh, kw, cin_total, cout = w.shape
    cin = cin_total - c_off if cin is None else cin
    nbytes = lib.mpg_conv_pack_size(kh, kw, cin, cout, prec)
    if nbytes == 0:
        raise _lib.MpgError("mpg_conv_pack_size: unsupported conv %dx%d %d->%d" % (kh, kw, cin, cout))
    buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    cs = _dev(cout_scale, "cout_scale") if cout_scale is not None else None
    rc = lib.mpg_conv_pack_weights(_stream(), _ptr(w), kh, kw, cin_total, c_off, cin, cout, float(wscale), _ptr(cs),
                                   prec, _ptr(buf), nbytes)
    _lib.check(rc, "mpg_conv_pack_weights")
    return PackedWeights(buf, kh, kw, cin, cout, prec)


class Segment(object):
    """One K-slice of a fused convolution: G8 source (an fp32 NHWC tensor is converted on the
    fly), first channel group consumed, packed weights, fused nearest upsample."""

    __slots__ = ("x", "packed", "g_off", "up_log2", "pad_hi")

    def __init__(self, x, packed, c_off=0, up_log2=0, pad_hi=0):
        self.pad_hi = int(pad_hi)
        if isinstance(x, torch.Tensor):
            x = to_g8(x, c_off, packed.cin, flavour_for(packed.prec))
            c_off = 0
        if c_off % 8:
            raise _lib.MpgError("Segment: channel offset %d of a G8 source is not a multiple of 8" % c_off)
        self.x, self.packed, self.g_off, self.up_log2 = x, packed, c_off // 8, up_log2


def conv2d_fused(segments, out_hw, bias=None, act=None, leak=0.2, pixel_norm=False, pn_eps=1e-8,
                 post_add=None, post_add_coff=0, out=None, want_f32=True, want_g8=False, reserved=0,
                 in_amax=None):
    """y = post(act(sum_s conv_SAME(up_s(x_s), W_s) + bias)) [+ post_add]; see include/mpgan.h.
    Returns the requested outputs in the order (fp32 NHWC, G8): a single object when one is requested,
    else a tuple."""
    lib = _lib.load()
    if not 1 <= len(segments) <= _lib.MAX_SEG:
        raise _lib.MpgError("conv2d_fused: %d segments (1..%d supported)" % (len(segments), _lib.MAX_SEG))
    p0 = segments[0].packed
    h, w = out_hw
    n = segments[0].x.n
    dev = segments[0].x.buf.device
    d = _lib.ConvDesc()
    d.n, d.h, d.w, d.cout, d.nseg = n, h, w, p0.cout, len(segments)
    for i, s in enumerate(segments):
        g8, pk = s.x, s.packed
        if (pk.cout, pk.prec) != (p0.cout, p0.prec):
            raise _lib.MpgError("conv2d_fused: segments packed with different cout/prec")
        if g8.flavour != flavour_for(pk.prec):
            raise _lib.MpgError("conv2d_fused: segment %d input has G8 flavour %d, precision %d reads flavour %d"
                                % (i, g8.flavour, pk.prec, flavour_for(pk.prec)))
        if g8.n != n or g8.h << s.up_log2 != h or g8.w << s.up_log2 != w:
            raise _lib.MpgError("conv2d_fused: segment %d input %dx%dx%d does not match output %dx%dx%d (up 2^%d)"
                                % (i, g8.n, g8.h, g8.w, n, h, w, s.up_log2))
        if s.g_off * 8 + pk.cin > g8.c:
            raise _lib.MpgError("conv2d_fused: segment %d channel window [%d,%d) exceeds %d"
                                % (i, s.g_off * 8, s.g_off * 8 + pk.cin, g8.c))
        g = d.seg[i]
        g.x, g.wpack = g8.buf.data_ptr(), pk.buf.data_ptr()
        g.cin, g.cgroups, g.g_off = pk.cin, g8.groups, s.g_off
        g.kh, g.kw, g.up_log2 = pk.kh, pk.kw, s.up_log2
        g.pad_hi = s.pad_hi
    if bias is not None:
        b = _dev(bias, "bias")
        if b.numel() != p0.cout:
            raise _lib.MpgError("conv2d_fused: bias has %d entries, cout is %d" % (b.numel(), p0.cout))
        d.bias = b.data_ptr()
    d.act, d.leak = _lib.act_id(act), leak
    d.pixel_norm, d.pn_eps = int(bool(pixel_norm)), pn_eps
    if post_add is not None:
        pa = _dev(post_add, "post_add")
        if pa.dim() != 4 or tuple(pa.shape[:3]) != (n, h, w) or post_add_coff + p0.cout > pa.shape[3]:
            raise _lib.MpgError("conv2d_fused: post_add %s does not match output" % (tuple(pa.shape),))
        d.post_add, d.post_add_stride, d.post_add_coff = pa.data_ptr(), pa.shape[3], post_add_coff
    y = y8 = None
    if out is not None:
        want_f32 = True
    if want_f32:
        if out is None:
            y = torch.empty((n, h, w, p0.cout), dtype=torch.float32, device=dev)
        else:
            y = _dev(out, "out")
            if tuple(y.shape) != (n, h, w, p0.cout):
                raise _lib.MpgError("conv2d_fused: out has shape %s" % (tuple(y.shape),))
        d.y = y.data_ptr()
    if want_g8:
        y8 = G8.empty(n, h, w, p0.cout, dev, G8_F16)
        d.y_g8 = y8.buf.data_ptr()
    outs = [o for o in (y, y8) if o is not None]
    if not outs:
        raise _lib.MpgError("conv2d_fused: no output requested")
    d.prec, d.reserved = p0.prec, reserved
    if in_amax is not None:     # the inputs were converted with to_g8(..., amax=in_amax): undo the power-of-two scale
        d.in_amax = _dev(in_amax, "in_amax").data_ptr()
    _lib.check(lib.mpg_conv2d_fused(_stream(), ctypes.byref(d)), "mpg_conv2d_fused")
    return outs[0] if len(outs) == 1 else tuple(outs)


PAIR_MAX_CIN = 4


def small_pair_ok(cin, cmid, cout, ka, kb, ks=None, planner=False):
    """shapes mpg_conv2d_small_pair takes: <= 8 channels everywhere, odd filters up to 7x7, shortcut inside the input tile.
    planner=True: the shapes the launch planner fuses -- only inputs of <= 4 channels: the first convolution is recomputed
    on the halo of the second (x1.33 at 5x5 on 64x16 tiles), which one launch less pays for at 1 -> 2 -> 8 (26.8 against
    37.6 us per 8 slices of 256^2) and not at 8 -> 2 -> 1 (50.0 against 37.1 us; tools/probe_small.py)"""
    if not (1 <= cin <= 8 and 1 <= cmid <= 8 and 1 <= cout <= 8):
        return False
    if planner and cin > PAIR_MAX_CIN:
        return False
    for k in (ka, kb) + ((ks,) if ks is not None else ()):
        if k[0] % 2 == 0 or k[1] % 2 == 0 or max(k) > 7:
            return False
    return ks is None or (ks[0] <= ka[0] + kb[0] - 1 and ks[1] <= ka[1] + kb[1] - 1)


def conv2d_small_pair(x, c_off, up_log2, pk_a, pk_b, pk_s, out_hw, bias_a=None, act_a=None, leak_a=0.2, bias_b=None,
                      act_b=None, leak_b=0.2, want_f32=True, want_g8=False, out=None):
    """y = act_b(conv_b(act_a(conv_a(up(x)) + bias_a)) + conv_s(up(x)) + bias_b): a residual block of <= 8-channel
    convolutions as one launch (mpg_conv2d_small_pair); x: G8 (or fp32 NHWC, converted), pk_*: PackedWeights."""
    lib = _lib.load()
    if isinstance(x, torch.Tensor):
        x = to_g8(x, c_off, pk_a.cin)
        c_off = 0
    if c_off % 8:
        raise _lib.MpgError("conv2d_small_pair: channel offset %d of a G8 source is not a multiple of 8" % c_off)
    h, w = out_hw
    if x.h << up_log2 != h or x.w << up_log2 != w:
        raise _lib.MpgError("conv2d_small_pair: input %dx%d does not match output %dx%d (up 2^%d)" % (x.h, x.w, h, w, up_log2))
    if pk_b.cin != pk_a.cout or (pk_s is not None and (pk_s.cin != pk_a.cin or pk_s.cout != pk_b.cout)):
        raise _lib.MpgError("conv2d_small_pair: channel counts of the three convolutions do not chain")
    if len(set(p.prec for p in (pk_a, pk_b) + ((pk_s,) if pk_s is not None else ()))) != 1:
        raise _lib.MpgError("conv2d_small_pair: weights packed for different precisions")
    if c_off + pk_a.cin > x.c:
        raise _lib.MpgError("conv2d_small_pair: channel window exceeds the input")
    dev = x.buf.device
    d = _lib.SmallPairDesc()
    d.n, d.h, d.w = x.n, h, w
    d.x, d.cin, d.cgroups, d.g_off, d.up_log2 = x.buf.data_ptr(), pk_a.cin, x.groups, c_off // 8, up_log2
    d.wpack_a, d.kh_a, d.kw_a, d.cmid = pk_a.buf.data_ptr(), pk_a.kh, pk_a.kw, pk_a.cout
    d.wpack_b, d.kh_b, d.kw_b, d.cout = pk_b.buf.data_ptr(), pk_b.kh, pk_b.kw, pk_b.cout
    if pk_s is not None:
        d.wpack_s, d.kh_s, d.kw_s = pk_s.buf.data_ptr(), pk_s.kh, pk_s.kw
    if bias_a is not None:
        d.bias_a = _dev(bias_a, "bias_a").data_ptr()
    if bias_b is not None:
        d.bias_b = _dev(bias_b, "bias_b").data_ptr()
    d.act_a, d.leak_a, d.act_b, d.leak_b = _lib.act_id(act_a), leak_a, _lib.act_id(act_b), leak_b
    d.prec = pk_a.prec
    y = y8 = None
    if want_f32:
        if out is not None:
            if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != x.n * h * w * pk_b.cout or out.device != dev:
                raise _lib.MpgError("conv2d_small_pair: `out` does not hold a contiguous fp32 [%d,%d,%d,%d]" % (x.n, h, w, pk_b.cout))
            y = out.view(x.n, h, w, pk_b.cout)
        else:
            y = torch.empty((x.n, h, w, pk_b.cout), dtype=torch.float32, device=dev)
        d.y = y.data_ptr()
    if want_g8:
        y8 = G8.empty(x.n, h, w, pk_b.cout, dev, G8_F16)
        d.y_g8 = y8.buf.data_ptr()
    outs = [o for o in (y, y8) if o is not None]
    if not outs:
        raise _lib.MpgError("conv2d_small_pair: no output requested")
    _lib.check(lib.mpg_conv2d_small_pair(_stream(), ctypes.byref(d)), "mpg_conv2d_small_pair")
    return outs[0] if len(outs) == 1 else tuple(outs)


def conv2d_direct(x, w_hwio, stride=(1, 1), wscale=1.0, cout_scale=None, bias=None, act=None, leak=0.2):
    """fp32 vector-ALU convolution, any stride (tf.nn.conv2d SAME, GAN.py:686-691)."""
    lib = _lib.load()
    x = _dev(x, "x")
    w = _dev(w_hwio, "w_hwio")
    n, h, wd, cin = x.shape
    kh, kw, cin2, cout = w.shape
    if cin != cin2:
        raise _lib.MpgError("conv2d_direct: input has %d channels, weights expect %d" % (cin, cin2))
    sh, sw = stride
    oh, ow = -(-h // sh), -(-wd // sw)
    y = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x.device)
    cs = _dev(cout_scale, "cout_scale") if cout_scale is not None else None
    b = _dev(bias, "bias") if bias is not None else None
    rc = lib.mpg_conv2d_direct(_stream(), _ptr(x), n, h, wd, cin, _ptr(w), kh, kw, cout, sh, sw, float(wscale),
                               _ptr(cs), _ptr(b), _lib.act_id(act), leak, _ptr(y))
    _lib.check(rc, "mpg_conv2d_direct")
    return y


def depth_to_space(x, r):
    """tf.depth_to_space (GAN.pixel_shuffle, GAN.py:554-560): [N,H,W,C] -> [N,H*r,W*r,C/r^2]"""
    lib = _lib.load()
    x = _dev(x, "x")
    n, h, w, c = x.shape
    if c % (r * r):
        raise _lib.MpgError("depth_to_space: %d channels are not a multiple of %d^2" % (c, r))
    y = torch.empty((n, h * r, w * r, c // (r * r)), dtype=torch.float32, device=x.device)
    _lib.check(lib.mpg_depth_to_space(_stream(), _ptr(x), n, h, w, c, r, _ptr(y)), "mpg_depth_to_space")
    return y


def subpixel_filter(w_hwoi, stride):
    """The stride-s transposed convolution as ONE stride-1 SAME convolution with s*s*cout outputs followed by
    depth_to_space: output row s*q + p gathers x[q + e] * W[ky] for the taps ky with (p + pad - ky) = s*e, so each
    output phase p is a stride-1 correlation over the offsets e.  Returns V[K',K',cin,s*s*cout] (K' odd) in HWIO."""
    kh, kw, cout, cin = w_hwoi.shape
    s = int(stride)
    pad = [max(k - s, 0) // 2 for k in (kh, kw)]

    def taps(k, p0):
        out = []            # (phase, tap, offset e)
        for ph in range(s):
            for t in range(k):
                if (ph + p0 - t) % s == 0:
                    out.append((ph, t, (ph + p0 - t) // s))
        return out

    ty, tx = taps(kh, pad[0]), taps(kw, pad[1])
    ey = max([abs(e) for _, _, e in ty] + [0])
    ex = max([abs(e) for _, _, e in tx] + [0])
    v = torch.zeros((2 * ey + 1, 2 * ex + 1, cin, s * s * cout), dtype=w_hwoi.dtype, device=w_hwoi.device)
    for py, ky, dy in ty:
        for px, kx, dx in tx:
            c0 = (py * s + px) * cout
            v[ey + dy, ex + dx, :, c0:c0 + cout] = w_hwoi[ky, kx].t()
    return v


def conv2d_transpose(x, w_hwoi, stride=(1, 1), wscale=1.0, bias=None, act=None, leak=0.2, prec=None):
    """tf.nn.conv2d_transpose(x, W[kh,kw,cout,cin], [N, H*sh, W*sw, cout], strides, "SAME") + bias + act
    (GAN.deconvolutional_layer, GAN.py:566-619,703-708).  prec None: the fp32 vector-ALU kernel (any stride / filter).
    prec 1/2/3: the matrix cores -- stride 1 as the fused convolution with the mirrored filter, stride 2 as a fused
    convolution with the sub-pixel filter + depth_to_space; shapes that route cannot take fall back to the fp32 kernel."""
    lib = _lib.load()
    x = _dev(x, "x")
    w = _dev(w_hwoi, "w_hwoi")
    n, h, wd, cin = x.shape
    kh, kw, cout, cin2 = w.shape
    if cin != cin2:
        raise _lib.MpgError("conv2d_transpose: input has %d channels, filter expects %d" % (cin, cin2))
    sh, sw = stride
    b = _dev(bias, "bias") if bias is not None else None
    if prec is not None and sh == sw and sh in (1, 2):
        if sh == 1 and kh <= 7 and kw <= 7 and cout <= 128:
            v = w.flip(0, 1).permute(0, 1, 3, 2).contiguous()
            p = prec if f6_available(cout, [(kh, kw, cin)]) or prec != PREC_F16F6 else PREC_F16X3
            seg = Segment(x, pack_conv_weights(v, wscale=wscale, prec=p), pad_hi=1)
            return conv2d_fused([seg], (h, wd), bias=b, act=act, leak=leak)
        if sh == 2 and 4 * cout <= 128:
            v = subpixel_filter(w, 2)
            if v.shape[0] <= 7 and v.shape[1] <= 7:
                p = prec if f6_available(4 * cout, [(v.shape[0], v.shape[1], cin)]) or prec != PREC_F16F6 else PREC_F16X3
                seg = Segment(x, pack_conv_weights(v, wscale=wscale, prec=p))
                z = conv2d_fused([seg], (h, wd), bias=b.repeat(4) if b is not None else None, act=act, leak=leak)
                return depth_to_space(z, 2)
    y = torch.empty((n, h * sh, wd * sw, cout), dtype=torch.float32, device=x.device)
    rc = lib.mpg_conv2d_transpose(_stream(), _ptr(x), n, h, wd, cin, _ptr(w), kh, kw, cout, sh, sw, float(wscale), _ptr(b),
                                  _lib.act_id(act), leak, _ptr(y))
    _lib.check(rc, "mpg_conv2d_transpose")
    return y


def _resize(fn_name, x, oh, ow):
    lib = _lib.load()
    x = _dev(x, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, oh, ow, c), dtype=torch.float32, device=x.device)
    _lib.check(getattr(lib, fn_name)(_stream(), _ptr(x), n, h, w, c, _ptr(y), oh, ow), fn_name)
    return y


def resize_nearest(x, oh, ow):
    return _resize("mpg_resize_nearest", x, oh, ow)


def resize_bilinear(x, oh, ow):
    return _resize("mpg_resize_bilinear", x, oh, ow)


def resize_bicubic(x, oh, ow):
    return _resize("mpg_resize_bicubic", x, oh, ow)


def resize_images(x, oh, ow, method):
    """tf.image.resize_images method ids (GAN.py:541): 0 bilinear, 1 nearest, 2 bicubic."""
    if method == 0:
        return resize_bilinear(x, oh, ow)
    if method == 1:
        return resize_nearest(x, oh, ow)
    if method == 2:
        return resize_bicubic(x, oh, ow)
    raise _lib.MpgError("resize method %r not supported" % (method,))


def max_pool(x, k=2, s=2, want_arg=False):
    """tf.nn.max_pool(x, k, s, VALID) (GAN.py:152-159); with want_arg also the uint8 window positions of the maxima"""
    lib = _lib.load()
    x = _dev(x, "x")
    n, h, w, c = x.shape
    if h < k or w < k:
        raise _lib.MpgError("max_pool: window %d does not fit %dx%d" % (k, h, w))
    oh, ow = (h - k) // s + 1, (w - k) // s + 1
    y = torch.empty((n, oh, ow, c), dtype=torch.float32, device=x.device)
    arg = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=x.device) if want_arg else None
    _lib.check(lib.mpg_max_pool(_stream(), _ptr(x), n, h, w, c, k, s, _ptr(y), _ptr(arg) if want_arg else None), "mpg_max_pool")
    return (y, arg) if want_arg else y


def avg_pool2(x):
    lib = _lib.load()
    x = _dev(x, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), dtype=torch.float32, device=x.device)
    _lib.check(lib.mpg_avg_pool2(_stream(), _ptr(x), n, h, w, c, _ptr(y)), "mpg_avg_pool2")
    return y


def pixel_norm(x, eps=1e-8):
    lib = _lib.load()
    x = _dev(x, "x")
    c = x.shape[-1]
    y = torch.empty_like(x)
    _lib.check(lib.mpg_pixel_norm(_stream(), _ptr(x), x.numel() // c, c, eps, _ptr(y)), "mpg_pixel_norm")
    return y


def minibatch_stddev(x, group_size=4):
    """GAN.minibatch_stddev_layer (GAN.py:476-488): [N,H,W,C] -> [N,H,W,C+1]"""
    lib = _lib.load()
    x = _dev(x, "x")
    n, h, w, c = x.shape
    g = min(group_size, n)
    stat = torch.empty((n // g,), dtype=torch.float32, device=x.device)
    y = torch.empty((n, h, w, c + 1), dtype=torch.float32, device=x.device)
    _lib.check(lib.mpg_minibatch_stddev(_stream(), _ptr(x), n, h, w, c, group_size, _ptr(stat), _ptr(y)), "mpg_minibatch_stddev")
    return y


def add_act(a, b=None, act=None, leak=0.2):
    lib = _lib.load()
    a = _dev(a, "a")
    if b is not None:
        b = _dev(b, "b")
        if b.shape != a.shape:
            raise _lib.MpgError("add_act: shapes differ %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    y = torch.empty_like(a)
    _lib.check(lib.mpg_add_act(_stream(), _ptr(a), _ptr(b), a.numel(), _lib.act_id(act), leak, _ptr(y)), "mpg_add_act")
    return y


def axis_zoom_linear(v, axis, factor):
    """scipy.ndimage.zoom(v, factor on `axis`, order=1) (multipassGAN-out.py:421)."""
    lib = _lib.load()
    v = _dev(v, "v")
    n = v.shape[axis]
    big = int(round(n * factor))
    outer = 1
    for d in v.shape[:axis]:
        outer *= d
    inner = 1
    for d in v.shape[axis + 1:]:
        inner *= d
    shape = list(v.shape)
    shape[axis] = big
    out = torch.empty(shape, dtype=torch.float32, device=v.device)
    if v.numel() == 0:
        return out
    _lib.check(lib.mpg_axis_zoom_linear(_stream(), _ptr(v), outer, n, inner, _ptr(out), big), "mpg_axis_zoom_linear")
    return out


def volume_transpose(v, perm, chan_map=None, cutoff=0.0):
    """numpy.transpose of [d0,d1,d2(,c)] by `perm` (3 axes) with optional channel
    permutation and cutoff (multipassGAN-out.py:459,472-475,614)."""
    lib = _lib.load()
    v = _dev(v, "v")
    if v.dim() == 3:
        d0, d1, d2 = v.shape
        c = 1
        squeeze = True
    elif v.dim() == 4:
        d0, d1, d2, c = v.shape
        squeeze = False
    else:
        raise _lib.MpgError("volume_transpose: expected a 3D or 4D tensor, got %s" % (tuple(v.shape),))
    perm = [int(p) for p in perm]
    if sorted(perm) != [0, 1, 2]:
        raise _lib.MpgError("volume_transpose: perm %r is not a permutation of (0,1,2)" % (perm,))
    dims = (d0, d1, d2)
    oshape = [dims[perm[0]], dims[perm[1]], dims[perm[2]]] + ([] if squeeze else [c])
    out = torch.empty(oshape, dtype=torch.float32, device=v.device)
    if v.numel() == 0:
        return out
    p_arr = (ctypes.c_int * 3)(*perm)
    cm = None
    if chan_map is not None:
        if len(chan_map) != c:
            raise _lib.MpgError("volume_transpose: chan_map has %d entries for %d channels" % (len(chan_map), c))
        cm = (ctypes.c_int * c)(*[int(k) for k in chan_map])
    rc = lib.mpg_volume_transpose(_stream(), _ptr(v), d0, d1, d2, c, p_arr, cm, float(cutoff), _ptr(out))
    _lib.check(rc, "mpg_volume_transpose")
    return out


def add_adjacent(x, s_off=0, s_cnt=None):
    """add_adj_idcs channels for slices [s_off, s_off+s_cnt) of x [S,H,W,C] (multipassGAN-out.py:423-436)."""
    lib = _lib.load()
    x = _dev(x, "x")
    s, h, w, c = x.shape
    s_cnt = s - s_off if s_cnt is None else s_cnt
    out = torch.empty((s_cnt, h, w, c + 2), dtype=torch.float32, device=x.device)
    _lib.check(lib.mpg_add_adjacent(_stream(), _ptr(x), s, h * w, c, s_off, s_cnt, _ptr(out)), "mpg_add_adjacent")
    return out


def channel_gather(a, b, cmap, scales=None, scales2=None):
    """out[..., j] = (cat(a, b)[..., cmap[j]] * scales[j]) * scales2[j] in one pass (mpg_channel_gather): the slice / scale /
    concat steps of the velocity channels between the passes (multipassGAN-4x.py:278-283, 1113-1119)"""
    import ctypes
    lib = _lib.load()
    a = _dev(a.contiguous(), "a")
    ca, cb = a.shape[-1], 0
    if b is not None:
        b = _dev(b.contiguous(), "b")
        cb = b.shape[-1]
        if tuple(b.shape[:-1]) != tuple(a.shape[:-1]):
            raise _lib.MpgError("channel_gather: %s and %s differ outside the channel axis" % (tuple(a.shape), tuple(b.shape)))
    n = len(cmap)
    npix = a.numel() // ca
    out = torch.empty(tuple(a.shape[:-1]) + (n,), dtype=torch.float32, device=a.device)
    m = (ctypes.c_int * n)(*[int(c) for c in cmap])
    sc = (ctypes.c_float * n)(*[float(x) for x in (scales if scales is not None else [1.0] * n)])
    sc2 = (ctypes.c_float * n)(*[float(x) for x in (scales2 if scales2 is not None else [1.0] * n)])
    _lib.check(lib.mpg_channel_gather(_stream(), _ptr(a), ca, _ptr(b), cb, npix, m, sc, sc2, n, _ptr(out)), "mpg_channel_gather")
    return out


def cutoff(v, thr=0.0005, out=None):
    """v[v < thr] = 0 (multipassGAN-4x.py:1156-1157)."""
    lib = _lib.load()
    v = _dev(v, "v")
    out = torch.empty_like(v) if out is None else out
    _lib.check(lib.mpg_cutoff(_stream(), _ptr(v), v.numel(), thr, _ptr(out)), "mpg_cutoff")
    return out
