"""Thin wrappers over the training entries of the C ABI (include/mpgan.h, "Training step").
fp32 NHWC GPU tensors in, fp32 GPU tensors out; no CPU fallback."""
import torch

from . import _lib
from .ops import _dev, _ptr, _stream


def _cont(t, name):
    return _dev(t.contiguous(), name)


def conv2d_wgrad(x, dy, kh, kw, stride=(1, 1), wscale=1.0):
    """dL/dW (HWIO, W stored unscaled: GAN.py:664-668) of y = conv2d_SAME(x, W * wscale)."""
    lib = _lib.load()
    x, dy = _cont(x, "x"), _cont(dy, "dy")
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    sh, sw = stride
    if tuple(dy.shape[:3]) != (n, -(-h // sh), -(-w // sw)):
        raise _lib.MpgError("conv2d_wgrad: dy shape %s does not match x %s stride %s" % (tuple(dy.shape), tuple(x.shape), stride))
    dw = torch.empty((kh, kw, cin, cout), dtype=torch.float32, device=x.device)
    _lib.check(lib.mpg_conv2d_wgrad(_stream(), _ptr(x), n, h, w, cin, _ptr(dy), cout, kh, kw, sh, sw, float(wscale),
                                    _ptr(dw)), "mpg_conv2d_wgrad")
    return dw


def wgrad_mfma_ok(kh, kw, stride):
    return tuple(stride) == (1, 1) and 1 <= kh <= 7 and kw in (1, 3, 4, 5)


_UNIT = {}


def unit_amax(device):
    """device scalar 256.0: as `x_amax` it selects scale 1 (mpg::pow2_scale brings the maximum into [2^8, 2^9)) -- for
    forward activations, which are O(1) and need no scaling ahead of the fp16 hi/lo split"""
    key = str(device)
    if key not in _UNIT:
        _UNIT[key] = torch.full((), 256.0, dtype=torch.float32, device=device)
    return _UNIT[key]


def conv2d_wgrad_mfma(x, dy, kh, kw, wscale=1.0, prec=_lib.PREC_F16X3, dy_amax=None, x_amax=None):
    """stride-1 weight gradient on the matrix cores (mpg_conv2d_wgrad_mfma)"""
    lib = _lib.load()
    x, dy = _cont(x, "x"), _cont(dy, "dy")
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    if tuple(dy.shape[:3]) != (n, h, w):
        raise _lib.MpgError("conv2d_wgrad_mfma: dy shape %s does not match x %s" % (tuple(dy.shape), tuple(x.shape)))
    nbytes = lib.mpg_conv2d_wgrad_mfma_ws_bytes(n, h, w, cin, cout)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dw = torch.empty((kh, kw, cin, cout), dtype=torch.float32, device=x.device)
    _lib.check(lib.mpg_conv2d_wgrad_mfma(_stream(), _ptr(x), n, h, w, cin, _ptr(dy), cout, kh, kw, float(wscale), prec,
                                         _ptr(ws), nbytes, _ptr(dy_amax), _ptr(x_amax), _ptr(dw)), "mpg_conv2d_wgrad_mfma")
    return dw


def conv2d_wgrad_g8(xg, dg, kh, kw, wscale=1.0, prec=_lib.PREC_F16X3, x_amax=None, dy_amax=None):
    """the same weight gradient from G8 operands (ops.G8): the forward input as the convolution read it and the scaled
    dy of the data-gradient convolution; x_amax / dy_amax = the scalars they were scaled with (mpg_conv2d_wgrad_g8)"""
    lib = _lib.load()
    if (xg.n, xg.h, xg.w) != (dg.n, dg.h, dg.w):
        raise _lib.MpgError("conv2d_wgrad_g8: dy %s does not match x %s" % ((dg.n, dg.h, dg.w), (xg.n, xg.h, xg.w)))
    dw = torch.empty((kh, kw, xg.c, dg.c), dtype=torch.float32, device=xg.buf.device)
    _lib.check(lib.mpg_conv2d_wgrad_g8(_stream(), _ptr(xg.buf), xg.n, xg.h, xg.w, xg.c, _ptr(dg.buf), dg.c, kh, kw,
                                       float(wscale), prec, _ptr(x_amax), _ptr(dy_amax), _ptr(dw)), "mpg_conv2d_wgrad_g8")
    return dw


def conv2d_dgrad(dy, w_hwio, in_hw, stride=(1, 1), wscale=1.0):
    """dL/dx of y = conv2d_SAME(x, W * wscale) for x of spatial size in_hw."""
    lib = _lib.load()
    kh, kw, cin, cout = w_hwio.shape
    dy, w = _cont(dy, "dy"), _cont(w_hwio.permute(0, 1, 3, 2), "w_hwoi")
    n = dy.shape[0]
    h, wd = in_hw
    sh, sw = stride
    if tuple(dy.shape) != (n, -(-h // sh), -(-wd // sw), cout):
        raise _lib.MpgError("conv2d_dgrad: dy shape %s does not match input %s stride %s" % (tuple(dy.shape), in_hw, stride))
    dx = torch.empty((n, h, wd, cin), dtype=torch.float32, device=dy.device)
    _lib.check(lib.mpg_conv2d_dgrad(_stream(), _ptr(dy), n, h, wd, cin, _ptr(w), cout, kh, kw, sh, sw, float(wscale),
                                    _ptr(dx)), "mpg_conv2d_dgrad")
    return dx


def fc_forward(x, w, wscale=1.0, bias=None, act=None, leak=0.2):
    """GAN.fully_connected_layer (GAN.py:438-456): act(x @ (w * wscale) + bias), x [rows, k], w [k, cout]"""
    lib = _lib.load()
    x, w = _cont(x, "x"), _cont(w, "w")
    rows, k = x.shape
    if w.shape[0] != k:
        raise _lib.MpgError("fc_forward: x has %d columns, w %d rows" % (k, w.shape[0]))
    cout = w.shape[1]
    y = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    b = _cont(bias, "bias") if bias is not None else None
    _lib.check(lib.mpg_fc_forward(_stream(), _ptr(x), rows, k, _ptr(w), cout, float(wscale), _ptr(b), _lib.act_id(act),
                                  leak, _ptr(y)), "mpg_fc_forward")
    return y


def channel_sum(x):
    lib = _lib.load()
    x = _cont(x, "x")
    c = x.shape[-1]
    out = torch.empty((c,), dtype=torch.float32, device=x.device)
    nfl = lib.mpg_bn_partials_floats(c) + c
    partials = torch.empty((nfl,), dtype=torch.float32, device=x.device)
    _lib.check(lib.mpg_channel_sum_ordered(_stream(), _ptr(x), x.numel() // c, c, _ptr(out), _ptr(partials), nfl),
               "mpg_channel_sum_ordered")
    return out


def bn_train_fwd(x, gamma, beta, eps=1e-3, act=None, leak=0.2, moving_mean=None, moving_var=None, decay=0.999):
    """-> (y, batch_mean, batch_var[biased]); advances moving_mean / moving_var in place when given"""
    lib = _lib.load()
    x = _cont(x, "x")
    c = x.shape[-1]
    y = torch.empty_like(x)
    stats = torch.empty((2, c), dtype=torch.float32, device=x.device)
    mean, var = stats[0], stats[1]
    mm = _dev(moving_mean, "moving_mean") if moving_mean is not None else None
    mv = _dev(moving_var, "moving_var") if moving_var is not None else None
    # block sums added in block order, not by atomics: the statistics (and every ReLU mask behind them) are run-to-run stable
    nfl = lib.mpg_bn_partials_floats(c)
    partials = torch.empty((nfl,), dtype=torch.float32, device=x.device)
    _lib.check(lib.mpg_bn_train_fwd_ordered(_stream(), _ptr(x), x.numel() // c, c, _ptr(_cont(gamma, "gamma")),
                                            _ptr(_cont(beta, "beta")), float(eps), _lib.act_id(act), leak, _ptr(y), _ptr(mean),
                                            _ptr(var), _ptr(mm), _ptr(mv), float(decay), _ptr(partials), nfl),
               "mpg_bn_train_fwd_ordered")
    return y, mean, var


def bn_train_bwd(dy, x, mean, var, gamma, eps=1e-3, want_amax=False):
    """-> (dx, dgamma, dbeta [, max |dx| as a 0-dim tensor]); dy is the gradient at the normalised (pre-activation) output"""
    lib = _lib.load()
    dy, x = _cont(dy, "dy"), _cont(x, "x")
    c = x.shape[-1]
    dx = torch.empty_like(x)
    dgb = torch.empty((2, c), dtype=torch.float32, device=x.device)
    dgamma, dbeta = dgb[0], dgb[1]
    amax = torch.empty((), dtype=torch.float32, device=x.device) if want_amax else None
    nfl = lib.mpg_bn_partials_floats(c)
    partials = torch.empty((nfl,), dtype=torch.float32, device=x.device)
    _lib.check(lib.mpg_bn_train_bwd_ordered(_stream(), _ptr(dy), _ptr(x), x.numel() // c, c, _ptr(mean), _ptr(var),
                                            _ptr(_cont(gamma, "gamma")), float(eps), _ptr(dx), _ptr(dgamma), _ptr(dbeta),
                                            _ptr(amax) if want_amax else None, _ptr(partials), nfl), "mpg_bn_train_bwd_ordered")
    return (dx, dgamma, dbeta, amax) if want_amax else (dx, dgamma, dbeta)


def act_bwd(dy, y, act, leak=0.2, want_amax=False):
    """dx = dy * act'(.) [, max |dx| as a 0-dim tensor]"""
    lib = _lib.load()
    dy, y = _cont(dy, "dy"), _cont(y, "y")
    dx = torch.empty_like(dy)
    amax = torch.empty((), dtype=torch.float32, device=dy.device) if want_amax else None
    _lib.check(lib.mpg_act_bwd(_stream(), _ptr(dy), _ptr(y), dy.numel(), _lib.act_id(act), leak, _ptr(dx),
                               _ptr(amax) if want_amax else None), "mpg_act_bwd")
    return (dx, amax) if want_amax else dx


def pixel_norm_bwd(dy, x, eps=1e-8):
    lib = _lib.load()
    dy, x = _cont(dy, "dy"), _cont(x, "x")
    c = x.shape[-1]
    dx = torch.empty_like(x)
    _lib.check(lib.mpg_pixel_norm_bwd(_stream(), _ptr(dy), _ptr(x), x.numel() // c, c, float(eps), _ptr(dx)),
               "mpg_pixel_norm_bwd")
    return dx


def resize_nearest_bwd(dy, h, w):
    lib = _lib.load()
    dy = _cont(dy, "dy")
    n, oh, ow, c = dy.shape
    dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dy.device)
    _lib.check(lib.mpg_resize_nearest_bwd(_stream(), _ptr(dy), n, oh, ow, c, _ptr(dx), h, w), "mpg_resize_nearest_bwd")
    return dx


def max_pool_bwd(dy, arg, h, w, k, s):
    lib = _lib.load()
    dy = _cont(dy, "dy")
    n, _, _, c = dy.shape
    dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dy.device)
    _lib.check(lib.mpg_max_pool_bwd(_stream(), _ptr(dy), _ptr(arg), n, h, w, c, k, s, _ptr(dx)), "mpg_max_pool_bwd")
    return dx


def avg_pool2_bwd(dy, h, w):
    lib = _lib.load()
    dy = _cont(dy, "dy")
    n, _, _, c = dy.shape
    dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dy.device)
    _lib.check(lib.mpg_avg_pool2_bwd(_stream(), _ptr(dy), n, h, w, c, _ptr(dx)), "mpg_avg_pool2_bwd")
    return dx


def lerp(x, y, t):
    """x + (y - x) * clip(t, 0, 1) (multipassGAN-8x.py:598-599); x None = zeros_like(y)"""
    lib = _lib.load()
    y = _cont(y, "y")
    x = _cont(x, "x") if x is not None else None
    out = torch.empty_like(y)
    t = min(max(float(t), 0.0), 1.0)
    _lib.check(lib.mpg_lerp(_stream(), _ptr(x), _ptr(y), y.numel(), t, _ptr(out)), "mpg_lerp")
    return out


def tensor_resample(value, pos, clamp=True):
    """tensorResample (multipassGAN-4x.py:398-441): value [n,h,w,c], pos [n,h,w,2] = (y, x)"""
    lib = _lib.load()
    value, pos = _cont(value, "value"), _cont(pos, "pos")
    n, h, w, c = value.shape
    if tuple(pos.shape) != (n, h, w, 2):
        raise _lib.MpgError("tensor_resample: pos shape %s does not match value %s" % (tuple(pos.shape), tuple(value.shape)))
    out = torch.empty_like(value)
    _lib.check(lib.mpg_tensor_resample(_stream(), _ptr(value), _ptr(pos), n, h, w, c, int(bool(clamp)), _ptr(out)),
               "mpg_tensor_resample")
    return out


def tensor_resample_bwd(dy, pos, clamp=True):
    lib = _lib.load()
    dy, pos = _cont(dy, "dy"), _cont(pos, "pos")
    n, h, w, c = dy.shape
    dv = torch.empty_like(dy)
    _lib.check(lib.mpg_tensor_resample_bwd(_stream(), _ptr(dy), _ptr(pos), n, h, w, c, int(bool(clamp)), _ptr(dv)),
               "mpg_tensor_resample_bwd")
    return dv


def adam_step_staged(p, grad, m, v, mask, state, lr, total_grads, use_loss_scaling, beta1, beta2, eps=1e-8, ls_inc=0.0005,
                     ls_dec=1.0, ema_shadow=None, ema_decay=0.999):
    """one optimiser call of multipassGAN-8x.py:1305-1362 / 490-541 on flat fp32 buffers (mpg_adam_step_staged)"""
    lib = _lib.load()
    rc = lib.mpg_adam_step_staged(_stream(), _ptr(p), _ptr(grad), _ptr(m), _ptr(v), _ptr(mask), p.numel(), _ptr(state), _ptr(lr),
                                  int(total_grads), int(bool(use_loss_scaling)), float(beta1), float(beta2), float(eps),
                                  float(ls_inc), float(ls_dec), _ptr(ema_shadow), float(ema_decay))
    _lib.check(rc, "mpg_adam_step_staged")


def advect_velocity(vel, h, w, dt):
    """the (y, x) displacement field GAN.advect looks up with (GAN.py:376-396): vel [n,hv,wv,>=2] (x,y,..) -> [n,h,w,2]"""
    lib = _lib.load()
    vel = _cont(vel, "vel")
    n, hv, wv, cv = vel.shape
    out = torch.empty((n, h, w, 2), dtype=torch.float32, device=vel.device)
    _lib.check(lib.mpg_advect_velocity(_stream(), _ptr(vel), n, hv, wv, cv, h, w, float(dt), _ptr(out)), "mpg_advect_velocity")
    return out


def semi_lagrange(source, vel_c, sign=1.0):
    lib = _lib.load()
    source, vel_c = _cont(source, "source"), _cont(vel_c, "vel")
    n, h, w, c = source.shape
    if tuple(vel_c.shape) != (n, h, w, 2):
        raise _lib.MpgError("semi_lagrange: displacement field %s does not match source %s" % (tuple(vel_c.shape), tuple(source.shape)))
    out = torch.empty_like(source)
    _lib.check(lib.mpg_semi_lagrange(_stream(), _ptr(source), _ptr(vel_c), n, h, w, c, float(sign), _ptr(out)), "mpg_semi_lagrange")
    return out


def semi_lagrange_bwd(dy, vel_c, sign=1.0):
    lib = _lib.load()
    dy, vel_c = _cont(dy, "dy"), _cont(vel_c, "vel")
    n, h, w, c = dy.shape
    ds = torch.empty_like(dy)
    _lib.check(lib.mpg_semi_lagrange_bwd(_stream(), _ptr(dy), _ptr(vel_c), n, h, w, c, float(sign), _ptr(ds)), "mpg_semi_lagrange_bwd")
    return ds


class _SemiLagrangeFn(torch.autograd.Function):
    """differentiable in `source` (a fixed linear gather): what the generator loss needs of GAN.advect with order 1"""

    @staticmethod
    def forward(ctx, source, vel_c, sign):
        ctx.save_for_backward(vel_c)
        ctx.sign = sign
        return semi_lagrange(source, vel_c, sign)

    @staticmethod
    def backward(ctx, dy):
        (vel_c,) = ctx.saved_tensors
        return semi_lagrange_bwd(dy.contiguous(), vel_c, ctx.sign), None, None


class _MacCormackFn(torch.autograd.Function):
    """order 2 of GAN.advect on one channel.  TensorFlow differentiates the op graph of GAN.py:206-343 branch by branch:
    the min / max clamp and the flag test only select, so with keep = (correction survived)
        out = fwd + keep * strength/2 * (source - bwd),  fwd = SL(source, +v),  bwd = SL(fwd, -v)
    and d source = g + SL^T(+v)[dy - SL^T(-v)[g]] with g = keep * strength/2 * dy."""

    @staticmethod
    def forward(ctx, source, vel_c, flags, strength):
        lib = _lib.load()
        n, h, w, _ = source.shape
        src = _cont(source, "source")
        fwd = semi_lagrange(src, vel_c, 1.0)
        bwd = semi_lagrange(fwd, vel_c, -1.0)
        out, keep = torch.empty_like(src), torch.empty_like(src)
        _lib.check(lib.mpg_maccormack(_stream(), _ptr(src), _ptr(fwd), _ptr(bwd), _ptr(flags), _ptr(vel_c), n, h, w,
                                      float(strength), _ptr(out), _ptr(keep)), "mpg_maccormack")
        ctx.save_for_backward(vel_c, keep)
        ctx.strength = float(strength)
        return out

    @staticmethod
    def backward(ctx, dy):
        vel_c, keep = ctx.saved_tensors
        g = keep * (0.5 * ctx.strength) * dy
        d_fwd = dy - semi_lagrange_bwd(g, vel_c, -1.0)
        return g + semi_lagrange_bwd(d_fwd, vel_c, 1.0), None, None, None


def advect(source, vel, flags, dt, order, strength=0.0, start_bz=15):
    """GAN.advect (GAN.py:347-418) on [n,h,w,c] device tensors (h == w, n a multiple of 3), differentiable in source.
    order 1: semi-Lagrangian; order 2: MacCormack with the reference's clamp (one channel; `start_bz` must be the
    batch size, as the reference's tf.where requires)."""
    n, h, w, c = source.shape
    if h != w:
        raise _lib.MpgError("advect: the reference's position grid is only a mesh for square fields (GAN.py:362-374)")
    vel_c = advect_velocity(vel, h, w, dt)
    if order != 2:
        return _SemiLagrangeFn.apply(source, vel_c, 1.0)
    if c != 1 or start_bz != n:
        raise _lib.MpgError("advect order 2: one channel and startBz == batch size (tf.where in GAN.py:343 needs both)")
    flags = _cont(flags.reshape(n, h, w, 1).to(torch.float32), "flags")
    return _MacCormackFn.apply(source, vel_c, flags, float(strength))


def minibatch_stddev_bwd(dy, x, group_size):
    """gradient of GAN.minibatch_stddev_layer (GAN.py:476-488) with respect to its input"""
    lib = _lib.load()
    dy, x = _cont(dy, "dy"), _cont(x, "x")
    n, h, w, c = x.shape
    g = min(group_size, n)
    dstat = torch.empty((n // g,), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    _lib.check(lib.mpg_minibatch_stddev_bwd(_stream(), _ptr(x), _ptr(dy), n, h, w, c, group_size, _ptr(dstat), _ptr(dx)),
               "mpg_minibatch_stddev_bwd")
    return dx


def pair_reduce(a, b, mode):
    """0-dim tensor: sum |a - b| (mode 0) or sum (a - b)^2 (mode 1)"""
    lib = _lib.load()
    a = _cont(a, "a")
    b = _cont(b, "b") if b is not None else None
    out = torch.empty((), dtype=torch.float32, device=a.device)
    _lib.check(lib.mpg_pair_reduce(_stream(), _ptr(a), _ptr(b), a.numel(), mode, _ptr(out)), "mpg_pair_reduce")
    return out


def adam_step(p, grad, m, v, lr_t, beta1, beta2, eps=1e-8):
    """in-place tf.train.AdamOptimizer update of the flat fp32 buffer p; lr_t is a 1-element GPU tensor"""
    lib = _lib.load()
    if not isinstance(lr_t, torch.Tensor):
        lr_t = torch.full((1,), float(lr_t), dtype=torch.float32, device=p.device)
    for t, nm in ((p, "p"), (grad, "grad"), (m, "m"), (v, "v"), (lr_t, "lr_t")):
        _dev(t, nm)
        if not t.is_contiguous():
            raise _lib.MpgError("adam_step: %s must be contiguous" % nm)
    _lib.check(lib.mpg_adam_step(_stream(), _ptr(p), _ptr(grad), _ptr(m), _ptr(v), p.numel(), _ptr(lr_t), float(beta1),
                                 float(beta2), float(eps)), "mpg_adam_step")
