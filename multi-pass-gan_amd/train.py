"""Training step of the 4x multi-pass GAN on the HIP kernels.

The reference builds one TF graph and lets ``tf.gradients`` + ``AdamOptimizer.minimize`` walk it
(multipassGAN-4x.py:728-902, training loop :1300-1360).  Here the same ``graph`` nodes are executed
eagerly by ``TrainSession``; every layer is a ``torch.autograd.Function`` whose forward AND backward
are kernels of libmpgan_hip.so (conv forward + data gradient on the MFMA kernel, weight gradient,
batch-norm statistics, activation / resize / pool / pixel-norm backward).  torch supplies the tape,
views (reshape / concat / slice), the scalar loss reductions and device memory, as north_star puts
the losses and optimiser bookkeeping host/PyTorch side; the Adam update itself is ``mpg_adam_step``
with TensorFlow's epsilon placement.
"""
import math

import numpy as np

import torch

from . import _lib, ops, train_ops
from . import graph as G

TRAINABLE_KINDS = ("weight", "bias", "gamma", "beta")


def _mfma_ok(kh, kw, width, stride):
    """the fused MFMA kernel takes stride-1 filters up to 7x7 and 128 output channels per launch;
    wider layers (the 256-wide d_c4, multipassGAN-4x.py:614) run as two launches"""
    return stride == (1, 1) and kh <= 7 and kw <= 7 and width <= 512


def _conv_prec(prec, kh, kw, cin, cout):
    """the precision a forward / data-gradient convolution of the training step runs at when the trainer asks for `prec`:
    MPG_PREC_F16F6 (the inference default: 1e-4-grade, 1.6x faster on the wide layers) where the kernels cover the shape and
    the contraction is long enough to pay for it (the session's rule), else the fp32-grade three-product mode"""
    if prec != ops.PREC_F16F6:
        return prec
    if kh * kw * cin > ops.F16F6_MIN_K and ops.f6_available(cout, [(kh, kw, cin)]):
        return ops.PREC_F16F6
    return ops.PREC_F16X3


def _wgrad_prec(prec):
    """weight gradients contract over pixels on fp16 hi/lo splits only: three products unless one is asked for"""
    return prec if prec == ops.PREC_F16X1 else ops.PREC_F16X3


def _mfma_conv(x, w, wscale, prec, bias=None, act=None, leak=0.2, pad_hi=0, rescale=False, amax=None, keep=None):
    """conv2d_SAME(x, w * wscale) [+ bias, act] on the MFMA kernel, output channels in chunks of 128.
    rescale: x is a gradient (1e-4 .. 1e-8 in magnitude, below the fp16 normal range): it is split into
    fp16 hi/lo after a power-of-two scaling by its absolute maximum, and the sum is scaled back in the epilogue.
    keep: a list that receives the G8 form of x the kernel read (the weight gradient reads the same tensor)"""
    kh, kw, cin, cout = w.shape
    outs = []
    if not rescale:
        amax = None
        if isinstance(x, torch.Tensor):
            # a block's input feeds two convolutions (first conv and 1x1 shortcut): the second one finds the G8 form the first
            # one made, on the tensor itself (valid while the tensor is neither rewritten in place nor another storage)
            x = x.contiguous()
            tag = getattr(x, "_mpg_g8", None)
            if tag is None or tag[1] != x._version or tag[2] != x.data_ptr() or tag[3] != tuple(x.shape):
                tag = (ops.to_g8(x, 0, x.shape[3], ops.flavour_for(prec)), x._version, x.data_ptr(), tuple(x.shape))
                try:
                    x._mpg_g8 = tag
                except AttributeError:      # a tensor type that takes no attributes: convert every time
                    pass
            x = tag[0]
    elif isinstance(x, torch.Tensor):
        amax = ops.absmax(x) if amax is None else amax
        x = ops.to_g8(x.contiguous(), 0, x.shape[3], ops.flavour_for(prec), amax=amax)
    for c0 in range(0, cout, 128):
        c1 = min(cout, c0 + 128)
        wc = w if (c0 == 0 and c1 == cout) else w[..., c0:c1].contiguous()
        pk = ops.pack_conv_weights(wc, wscale=wscale, prec=_conv_prec(prec, kh, kw, cin, c1 - c0))
        bc = None if bias is None else (bias if (c0 == 0 and c1 == cout) else bias[c0:c1].contiguous())
        seg = ops.Segment(x, pk, pad_hi=pad_hi)
        if len(outs) == 0 and keep is not None:
            keep.append(seg.x)
        if len(outs) == 0 and c1 < cout and isinstance(x, torch.Tensor):
            x = seg.x                       # reuse the G8 conversion for the other chunks
        outs.append(ops.conv2d_fused([seg], (seg.x.h, seg.x.w), bias=bc, act=act, leak=leak, in_amax=amax))
    return outs[0] if len(outs) == 1 else torch.cat(outs, dim=3)


class ConvLayerFn(torch.autograd.Function):
    """act(bn_train?(conv2d_SAME(x, W * wscale) + b)) as GAN.convolutional_layer builds it
    (GAN.py:80-119); one fused conv launch (+ the batch-statistics passes when BN is on)."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, cfg):
        stride, wscale, act, leak = cfg["stride"], cfg["wscale"], cfg["act"], cfg["leak"]
        kh, kw, cin, cout = w.shape
        x = x.contiguous()
        x_g8 = None
        bn = gamma is not None
        conv_act = None if bn else act
        if cfg.get("fc"):
            lin = train_ops.fc_forward(x.reshape(x.shape[0], cin), w.detach().reshape(cin, cout), wscale, b, conv_act,
                                       leak).reshape(x.shape[0], 1, 1, cout)
        elif _mfma_ok(kh, kw, cout, stride):
            keep = [] if (ctx.needs_input_grad[1] and train_ops.wgrad_mfma_ok(kh, kw, stride) and cfg.get("wgrad_mfma", True)) else None
            lin = _mfma_conv(x, w.detach().contiguous(), wscale, cfg["prec"], b.detach() if b is not None else None,
                             conv_act, leak, keep=keep)
            x_g8 = keep[0] if keep else None
        else:
            lin = ops.conv2d_direct(x, w.detach().contiguous(), stride, wscale, None, b, conv_act, leak)
        if bn:
            mm, mv = cfg.get("moving", (None, None))
            y, mean, var = train_ops.bn_train_fwd(lin, gamma, beta, cfg["eps"], act, leak, mm, mv, cfg.get("decay", 0.999))
            cfg["batch_stats"] = (mean, var)
            ctx.save_for_backward(x, w, lin, mean, var, gamma, y)
        else:
            y = lin
            ctx.save_for_backward(x, w, y)
        ctx.cfg, ctx.bn, ctx.has_bias = cfg, bn, b is not None
        ctx.x_g8 = x_g8                      # the forward input as the kernel read it (hi/lo fp16, unscaled): the weight gradient's operand
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg = ctx.cfg
        stride, wscale, act, leak = cfg["stride"], cfg["wscale"], cfg["act"], cfg["leak"]
        x_g8, ctx.x_g8 = ctx.x_g8, None
        if ctx.bn:
            x, w, lin, mean, var, gamma, y = ctx.saved_tensors
        else:
            x, w, y = ctx.saved_tensors
        kh, kw, cin, cout = w.shape
        d = dy.contiguous()
        # max |d| (the power-of-two scale of both gradient convolutions) comes out of the kernel that produces d
        need_amax = stride == (1, 1) and not cfg.get("fc")
        d_amax = None
        if act is not None:
            if need_amax and not ctx.bn:
                d, d_amax = train_ops.act_bwd(d, y, act, leak, want_amax=True)
            else:
                d = train_ops.act_bwd(d, y, act, leak)
        dgamma = dbeta = None
        if ctx.bn:
            if need_amax:
                d, dgamma, dbeta, d_amax = train_ops.bn_train_bwd(d, lin, mean, var, gamma, cfg["eps"], want_amax=True)
            else:
                d, dgamma, dbeta = train_ops.bn_train_bwd(d, lin, mean, var, gamma, cfg["eps"])
        db = train_ops.channel_sum(d) if ctx.has_bias else None
        dw = None
        if need_amax and d_amax is None:
            d_amax = ops.absmax(d)                                                      # shared by both gradients
        wgrad_mm = ctx.needs_input_grad[1] and train_ops.wgrad_mfma_ok(kh, kw, stride) and not cfg.get("fc") and \
            cfg.get("wgrad_mfma", True)
        dgrad_mm = ctx.needs_input_grad[0] and _mfma_ok(kh, kw, cin, stride) and not cfg.get("fc")
        d_g8 = None
        if x_g8 is not None and wgrad_mm:
            # one scaled hi/lo conversion of d feeds both gradient kernels; x comes from the forward launch
            d_g8 = ops.to_g8(d, 0, cout, ops.flavour_for(cfg["prec"]), amax=d_amax)
        if ctx.needs_input_grad[1]:
            if d_g8 is not None:
                dw = train_ops.conv2d_wgrad_g8(x_g8, d_g8, kh, kw, wscale, _wgrad_prec(cfg["prec"]), None, d_amax)
            elif wgrad_mm:
                # x is the layer's forward input (an activation): split unscaled, no abs-max pass over it
                dw = train_ops.conv2d_wgrad_mfma(x, d, kh, kw, wscale, _wgrad_prec(cfg["prec"]), d_amax,
                                                 train_ops.unit_amax(x.device))
            else:
                dw = train_ops.conv2d_wgrad(x, d, kh, kw, stride, wscale)
        dx = None
        if ctx.needs_input_grad[0]:
            if dgrad_mm:
                wt = w.detach().flip(0, 1).permute(0, 1, 3, 2).contiguous()      # [kh,kw,cout,cin], taps mirrored
                dx = _mfma_conv(d if d_g8 is None else d_g8, wt, wscale, cfg["prec"], pad_hi=1, rescale=True, amax=d_amax)
            else:
                dx = train_ops.conv2d_dgrad(d, w.detach(), (x.shape[1], x.shape[2]), stride, wscale)
        return dx, dw, db, dgamma, dbeta, None


# ----------------------------------------------------------------------------------------------
# Layers whose backward is itself differentiable (WGAN-GP: the gradient penalty of multipassGAN-8x.py
# :1123-1140 differentiates |dD/dx| with respect to the discriminator weights).  Convolution, its data
# gradient and its weight gradient are closed under differentiation:
#   y  = conv(x, w)      : dx = dgrad(dy, w)     dw = wgrad(x, dy)
#   dx = dgrad(dy, w)    : d(dy) = conv(g, w)    dw = wgrad(g, dy)
#   dw = wgrad(x, dy)    : dx = dgrad(dy, G)     d(dy) = conv(x, G)
# so three Functions that call each other give gradients of any order from the same kernels.
# ----------------------------------------------------------------------------------------------
def _scaled_g8(t, prec):
    """(G8 of t scaled by the power of two of its abs-max, the abs-max), kept on the tensor: a gradient that enters both the
    data- and the weight-gradient convolution of a layer (or a second-order forward and its weight gradient) is reduced
    and converted once.  Valid while the tensor is neither rewritten in place nor another storage."""
    tag = getattr(t, "_mpg_sg8", None)
    if tag is None or tag[2] != t._version or tag[3] != t.data_ptr() or tag[4] != tuple(t.shape):
        tc = t.detach().contiguous()
        am = ops.absmax(tc)
        tag = (ops.to_g8(tc, 0, tc.shape[3], ops.flavour_for(prec), amax=am), am, t._version, t.data_ptr(), tuple(t.shape))
        try:
            t._mpg_sg8 = tag
        except AttributeError:
            pass
    return tag[0], tag[1]


def _conv_fwd(x, w, b, cfg):
    """x may be the autograd tensor itself (its scaled G8 form is cached on it); w, b detached"""
    kh, kw, cin, cout = w.shape
    if cfg.get("fc"):
        xd = x.detach()
        return train_ops.fc_forward(xd.reshape(xd.shape[0], cin), w.reshape(cin, cout), cfg["wscale"], b).reshape(
            xd.shape[0], 1, 1, cout)
    if _mfma_ok(kh, kw, cout, cfg["stride"]):
        if cfg.get("rescale_fwd") and x.dim() == 4:
            g8, am = _scaled_g8(x, cfg["prec"])
            return _mfma_conv(g8, w.contiguous(), cfg["wscale"], cfg["prec"], b, rescale=True, amax=am)
        return _mfma_conv(x.detach().contiguous(), w.contiguous(), cfg["wscale"], cfg["prec"], b)
    return ops.conv2d_direct(x.detach().contiguous(), w.contiguous(), cfg["stride"], cfg["wscale"], None, b)


def _conv_dgrad(dy, w, cfg, hw):
    kh, kw, cin, cout = w.shape
    if _mfma_ok(kh, kw, cin, cfg["stride"]) and not cfg.get("fc"):
        g8, am = _scaled_g8(dy, cfg["prec"])
        return _mfma_conv(g8, w.flip(0, 1).permute(0, 1, 3, 2).contiguous(), cfg["wscale"], cfg["prec"],
                          pad_hi=1, rescale=True, amax=am)
    return train_ops.conv2d_dgrad(dy.detach(), w, hw, cfg["stride"], cfg["wscale"])


def _conv_wgrad(x, dy, cfg, kh, kw):
    if train_ops.wgrad_mfma_ok(kh, kw, cfg["stride"]) and not cfg.get("fc"):
        xg, xam = _scaled_g8(x, cfg["prec"])
        dg, dam = _scaled_g8(dy, cfg["prec"])
        return train_ops.conv2d_wgrad_g8(xg, dg, kh, kw, cfg["wscale"], _wgrad_prec(cfg["prec"]), xam, dam)
    return train_ops.conv2d_wgrad(x.detach(), dy.detach(), kh, kw, cfg["stride"], cfg["wscale"])


class ConvFn(torch.autograd.Function):
    """y = conv2d_SAME(x, w * wscale) + b, differentiable to any order"""

    @staticmethod
    def forward(ctx, x, w, b, cfg):
        ctx.save_for_backward(x, w)
        ctx.cfg, ctx.has_bias = cfg, b is not None
        return _conv_fwd(x, w.detach(), b.detach() if b is not None else None, cfg)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = ConvDgradFn.apply(dy, w, ctx.cfg, (x.shape[1], x.shape[2])) if ctx.needs_input_grad[0] else None
        dw = ConvWgradFn.apply(x, dy, ctx.cfg, w.shape[0], w.shape[1]) if ctx.needs_input_grad[1] else None
        db = train_ops.channel_sum(dy.detach()) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None


class ConvDgradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, w, cfg, hw):
        ctx.save_for_backward(dy, w)
        ctx.cfg = cfg
        return _conv_dgrad(dy, w.detach(), cfg, hw)

    @staticmethod
    def backward(ctx, g):
        dy, w = ctx.saved_tensors
        # g is a gradient of a gradient: as small as dy itself, so the forward conv over it is abs-max scaled too
        ddy = ConvFn.apply(g, w, None, dict(ctx.cfg, rescale_fwd=True)) if ctx.needs_input_grad[0] else None
        dw = ConvWgradFn.apply(g, dy, ctx.cfg, w.shape[0], w.shape[1]) if ctx.needs_input_grad[1] else None
        return ddy, dw, None, None


class ConvWgradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dy, cfg, kh, kw):
        ctx.save_for_backward(x, dy)
        ctx.cfg = cfg
        return _conv_wgrad(x, dy, cfg, kh, kw)

    @staticmethod
    def backward(ctx, gw):
        x, dy = ctx.saved_tensors
        dx = ConvDgradFn.apply(dy, gw, ctx.cfg, (x.shape[1], x.shape[2])) if ctx.needs_input_grad[0] else None
        ddy = ConvFn.apply(x, gw, None, ctx.cfg) if ctx.needs_input_grad[1] else None
        return dx, ddy, None, None, None


class ActBwdFn(torch.autograd.Function):
    """dx = dy * act'(.) with the mask taken from the activation output: linear in dy"""

    @staticmethod
    def forward(ctx, dy, y, act, leak):
        ctx.save_for_backward(y)
        ctx.act, ctx.leak = act, leak
        return train_ops.act_bwd(dy.detach(), y, act, leak)

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return ActBwdFn.apply(g, y, ctx.act, ctx.leak), None, None, None


class Act2Fn(torch.autograd.Function):
    """act(x) for relu / lrelu (piecewise linear: the second derivative is zero almost everywhere)"""

    @staticmethod
    def forward(ctx, x, act, leak):
        y = ops.add_act(x.detach().contiguous(), None, act, leak)
        ctx.save_for_backward(y)
        ctx.act, ctx.leak = act, leak
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ActBwdFn.apply(dy, y, ctx.act, ctx.leak), None, None


class AvgPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.hw = (x.shape[1], x.shape[2])
        return ops.avg_pool2(x.detach().contiguous())

    @staticmethod
    def backward(ctx, dy):
        return AvgPoolBwd2Fn.apply(dy, ctx.hw)


class AvgPoolBwd2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, hw):
        return train_ops.avg_pool2_bwd(dy.detach(), *hw)

    @staticmethod
    def backward(ctx, g):
        return AvgPool2Fn.apply(g), None


class ResizeNearest2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        ctx.hw, ctx.ohw = (x.shape[1], x.shape[2]), (oh, ow)
        return ops.resize_nearest(x.detach().contiguous(), oh, ow)

    @staticmethod
    def backward(ctx, dy):
        return ResizeNearestBwd2Fn.apply(dy, ctx.hw), None, None


class ResizeNearestBwd2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, hw):
        ctx.ohw = (dy.shape[1], dy.shape[2])
        return train_ops.resize_nearest_bwd(dy.detach(), *hw)

    @staticmethod
    def backward(ctx, g):
        return ResizeNearest2Fn.apply(g, *ctx.ohw), None


class Lerp2Fn(torch.autograd.Function):
    """lerp with a backward made of lerps (linear in both operands)"""

    @staticmethod
    def forward(ctx, x, y, t):
        ctx.t, ctx.has_x = min(max(float(t), 0.0), 1.0), x is not None
        return train_ops.lerp(x.detach() if x is not None else None, y.detach(), ctx.t)

    @staticmethod
    def backward(ctx, dy):
        dyy = Lerp2Fn.apply(None, dy, ctx.t) if ctx.needs_input_grad[1] else None
        dx = Lerp2Fn.apply(None, dy, 1.0 - ctx.t) if (ctx.has_x and ctx.needs_input_grad[0]) else None
        return dx, dyy, None


def space_to_depth2(x):
    """[N,H,W,C] -> [N,H/2,W/2,4C], channel (r*2+s)*C + c holds x[2Y+r, 2X+s, c]"""
    n, h, w, c = x.shape
    return x.reshape(n, h // 2, 2, w // 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(n, h // 2, w // 2, 4 * c)


def strided4_as_3x3(w):
    """The 4x4 stride-2 SAME filter of the discriminators (multipassGAN-4x.py:607-613) as a 3x3 stride-1
    filter over the space-to-depth input: y[Y,X] = sum_{a,b,r,s} z[Y+a-1, X+b-1, (r,s,c)] * W[2a+r-1, 2b+s-1, c]
    (taps outside 0..3 are zero).  With it the strided convs, their data and weight gradients all run on
    the matrix-core kernels; 16 of the 36 (a,b,r,s) positions carry weights."""
    kh, kw, c, co = w.shape
    wp = torch.nn.functional.pad(w, (0, 0, 0, 0, 1, 1, 1, 1))            # zero ring: [6,6,C,Co]
    return wp.reshape(3, 2, 3, 2, c, co).permute(0, 2, 1, 3, 4, 5).reshape(3, 3, 4 * c, co)


class ActFn(torch.autograd.Function):
    """act(a [+ b]) (tf.nn.relu(tf.add(B, s)), multipassGAN-4x.py:523)"""

    @staticmethod
    def forward(ctx, a, b, act, leak):
        y = ops.add_act(a.contiguous(), b.contiguous() if b is not None else None, act, leak)
        ctx.save_for_backward(y)
        ctx.act, ctx.leak, ctx.two = act, leak, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        d = train_ops.act_bwd(dy, y, ctx.act, ctx.leak) if ctx.act is not None else dy
        return d, (d if ctx.two else None), None, None


class ResizeNearestFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        ctx.hw = (x.shape[1], x.shape[2])
        return ops.resize_nearest(x.contiguous(), oh, ow)

    @staticmethod
    def backward(ctx, dy):
        return train_ops.resize_nearest_bwd(dy, *ctx.hw), None, None


class AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.hw = (x.shape[1], x.shape[2])
        return ops.avg_pool2(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return train_ops.avg_pool2_bwd(dy, *ctx.hw)


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s):
        x = x.contiguous()
        y, arg = ops.max_pool(x, k, s, want_arg=True)
        ctx.save_for_backward(arg)
        ctx.geom = (x.shape[1], x.shape[2], k, s)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        return train_ops.max_pool_bwd(dy, arg, *ctx.geom), None, None


class MinibatchStddevFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group_size):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.group_size = group_size
        return ops.minibatch_stddev(x, group_size)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return train_ops.minibatch_stddev_bwd(dy, x, ctx.group_size), None


class PixelNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.eps = eps
        return ops.pixel_norm(x, eps)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return train_ops.pixel_norm_bwd(dy, x, ctx.eps), None


class LerpFn(torch.autograd.Function):
    """lerp(x, y, t) (multipassGAN-8x.py:598-599); x may be None (tf.zeros_like)"""

    @staticmethod
    def forward(ctx, x, y, t):
        ctx.t = min(max(float(t), 0.0), 1.0)
        ctx.has_x = x is not None
        return train_ops.lerp(x, y, ctx.t)

    @staticmethod
    def backward(ctx, dy):
        dxy = train_ops.lerp(None, dy, ctx.t)
        dx = train_ops.lerp(None, dy, 1.0 - ctx.t) if ctx.has_x else None
        return dx, dxy, None


class PairLossFn(torch.autograd.Function):
    """sum |a - b| (mode 0) / sum (a - b)^2 (mode 1) with the reduction in ``mpg_pair_reduce``; the
    backward is elementwise"""

    @staticmethod
    def forward(ctx, a, b, mode):
        ctx.save_for_backward(a, b)
        ctx.mode = mode
        return train_ops.pair_reduce(a, b, mode)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        d = a - b
        ga = (torch.sign(d) if ctx.mode == 0 else 2.0 * d) * g
        return (ga if ctx.needs_input_grad[0] else None), (-ga if ctx.needs_input_grad[1] else None), None


class ResampleFn(torch.autograd.Function):
    """tensorResample(value, pos) (multipassGAN-4x.py:398-441); pos is data, the gradient goes to value"""

    @staticmethod
    def forward(ctx, value, pos, clamp):
        ctx.save_for_backward(pos)
        ctx.clamp = clamp
        return train_ops.tensor_resample(value, pos, clamp)

    @staticmethod
    def backward(ctx, dy):
        (pos,) = ctx.saved_tensors
        return train_ops.tensor_resample_bwd(dy, pos, ctx.clamp), None, None


class TrainSession(object):
    """Eager, differentiable evaluation of ``graph`` nodes; parameters are leaf tensors keyed by
    the TF variable path.  ``run(fetches, feeds)`` returns torch tensors that carry the tape."""

    def __init__(self, variables, graph=None, prec=ops.PREC_F16X3, bn_decay=0.999, device="cuda:0"):
        self.graph = graph or G.get_default_graph()
        self.vars = variables
        if prec not in (ops.PREC_F16X3, ops.PREC_F16X1, ops.PREC_F16F6):
            raise _lib.MpgError("training precision %r: MPG_PREC_F16X3 (default, fp32-grade), F16X1 or F16F6" % (prec,))
        # F16F6: forward and data-gradient convolutions at the inference default where the kernels cover the shape
        # (_conv_prec); the weight gradients contract over pixels with fp16 hi/lo operands and stay at three products
        self.prec = prec
        self.bn_decay = bn_decay
        self.device = torch.device(device)
        self.strided_on_mfma = True      # 4x4 stride-2 convs as 3x3 stride-1 convs over space-to-depth inputs
        # variable-name prefixes whose layers must support gradients of gradients (the WGAN-GP discriminators)
        self.higher_order_scopes = ()
        self.params = {}
        self._scalar_feeds = {}

    def parameters(self):
        """name -> leaf tensor for every variable of the graph (trainable ones require grad)"""
        self.vars.ensure(self.graph)
        for name, spec in self.graph.variables.items():
            if name not in self.params:
                t = self.vars.get(name).detach().clone().contiguous()
                t.requires_grad_(spec.kind in TRAINABLE_KINDS)
                self.params[name] = t
        return self.params

    def trainable(self, tag):
        """tf.trainable_variables() filtered like the reference: `tag in var.name` (multipassGAN-4x.py:780-784)"""
        ps = self.parameters()
        return {n: p for n, p in ps.items() if p.requires_grad and tag in n}

    def sync_to_store(self):
        for name, p in self.params.items():
            self.vars.values[name] = p.detach().clone()
        self.vars.version += 1

    # ------------------------------------------------------------------ evaluation
    def run(self, fetches, feeds):
        _lib.load()
        if not torch.cuda.is_available():
            raise _lib.MpgError("no GPU visible: the training step has no CPU fallback")
        self.parameters()
        env = {}
        self._scalar_feeds = {}
        for node, t in feeds.items():
            if isinstance(node, G.Scalar):
                self._scalar_feeds[node.node] = float(t)
            else:
                env[node.id] = torch.as_tensor(t, dtype=torch.float32, device=self.device)
        users = self._count_users(fetches)
        # the UPDATE_OPS of tf.contrib.layers.batch_norm (multipassGAN-4x.py:773-776,889-899) run inside
        # mpg_bn_train_fwd: every evaluated batch-norm layer advances its moving averages once
        return [self._eval(f, env, users) for f in fetches]

    def _count_users(self, fetches):
        users, seen, stack = {}, set(), list(fetches)
        for f in fetches:
            users[f.id] = users.get(f.id, 0) + 1
        while stack:
            n = stack.pop()
            if n.id in seen:
                continue
            seen.add(n.id)
            for i in n.inputs:
                users[i.id] = users.get(i.id, 0) + 1
                stack.append(i)
        return users

    def _layer_chain(self, n, users):
        """n = [act](batch_norm?(bias_add?(conv2d|matmul))) with single-use links -> (conv, bias, bn, act, leak)"""
        cur, act, leak = n, None, 0.2
        if cur.op == "act":
            nxt = cur.inputs[0]
            if users.get(nxt.id, 0) != 1 or nxt.op not in ("batch_norm", "bias_add", "conv2d", "matmul"):
                return None
            act, leak = cur.attrs["act"], cur.attrs.get("leak", 0.2)
            cur = nxt
        bn = bias = None
        if cur.op == "batch_norm":
            bn = cur
            if users.get(cur.inputs[0].id, 0) != 1:
                return None
            cur = cur.inputs[0]
        if cur.op == "bias_add":
            bias = cur
            if users.get(cur.inputs[0].id, 0) != 1:
                return None
            cur = cur.inputs[0]
        if cur.op not in ("conv2d", "matmul"):
            return None
        return cur, bias, bn, act, leak

    def _higher(self, n):
        """does this node sit in a variable scope that needs differentiable backward passes?"""
        return any(n.scope.startswith(p) for p in self.higher_order_scopes)

    def _eval(self, n, env, users):
        if n.id in env:
            return env[n.id]
        v = self._compute(n, env, users)
        env[n.id] = v
        return v

    def _compute(self, n, env, users):
        op = n.op
        ev = lambda m: self._eval(m, env, users)   # noqa: E731
        if op == "placeholder":
            raise G.GraphError("placeholder %r was not fed" % (n,))
        if op == "variable":
            return self.params[n.attrs["var"]]
        if op in ("act", "batch_norm", "bias_add", "conv2d", "matmul"):
            chain = self._layer_chain(n, users)
            if chain is not None:
                return self._conv_layer(chain, ev)
            if op == "act":
                x = n.inputs[0]
                if x.op == "add" and users.get(x.id, 0) == 1:
                    return ActFn.apply(ev(x.inputs[0]), ev(x.inputs[1]), n.attrs["act"], n.attrs.get("leak", 0.2))
                return ActFn.apply(ev(x), None, n.attrs["act"], n.attrs.get("leak", 0.2))
            raise G.GraphError("training: no lowering for %r" % (n,))
        if op == "add":
            return ActFn.apply(ev(n.inputs[0]), ev(n.inputs[1]), None, 0.2)
        if op == "reshape":
            return ev(n.inputs[0]).reshape(n.attrs["target"])
        if op == "concat":
            return torch.cat([ev(i) for i in n.inputs], dim=-1)
        if op == "slice":
            b, s = n.attrs["begin"], n.attrs["size"]
            return ev(n.inputs[0])[..., b:b + s]
        if op == "slice_flat":
            x = ev(n.inputs[0])
            return x.reshape(x.shape[0], -1)[:, :n.attrs["count"]]
        if op == "pixel_norm":
            return PixelNormFn.apply(ev(n.inputs[0]), n.attrs["eps"])
        if op == "avg_pool":
            return (AvgPool2Fn if self._higher(n) else AvgPoolFn).apply(ev(n.inputs[0]))
        if op == "lerp":
            t = n.attrs["t"]
            t = t.value(self._scalar_feeds) if isinstance(t, G.Scalar) else float(t)
            x = None if n.attrs["zero_x"] else ev(n.inputs[0])
            y = ev(n.inputs[-1])
            return (Lerp2Fn if self._higher(n) else LerpFn).apply(x, y, t)
        if op == "random_normal":
            ref = ev(n.inputs[0])
            gen = self.__dict__.setdefault("_noise_gen", {})
            if n.id not in gen:
                gen[n.id] = torch.Generator(device=ref.device).manual_seed(1000003 * n.attrs["seed"] + 17)
            shape = tuple(ref.shape[:-1]) + (n.shape[-1],)
            return torch.randn(shape, generator=gen[n.id], device=ref.device, dtype=torch.float32) * n.attrs["stddev"]
        if op == "max_pool":
            if self._higher(n):
                raise NotImplementedError("second-order gradient through max_pool (no reference network uses it)")
            return MaxPoolFn.apply(ev(n.inputs[0]), n.attrs["k"], n.attrs["s"])
        if op == "advect":
            src, vel = ev(n.inputs[0]), ev(n.inputs[1])
            flags = ev(n.inputs[2]) if len(n.inputs) > 2 else torch.zeros_like(src[..., :1])
            at = n.attrs
            return train_ops.advect(src, vel.detach(), flags.detach(), at["dt"], at["order"], at["strength"], at["start_bz"])
        if op == "minibatch_stddev":
            x = ev(n.inputs[0])
            if x.requires_grad and self._higher(n):
                raise NotImplementedError("second-order gradient of minibatch_stddev_layer (use_mb_stddev with the "
                                          "WGAN-GP penalty; 0 in every reference run)")
            return MinibatchStddevFn.apply(x, n.attrs["group_size"])
        if op == "resize":
            x = ev(n.inputs[0])
            if n.attrs["method"] == 1:
                return (ResizeNearest2Fn if self._higher(n) else ResizeNearestFn).apply(x, n.attrs["oh"], n.attrs["ow"])
            if x.requires_grad:
                raise NotImplementedError("gradient of bilinear / bicubic resize (only applied to network inputs)")
            return ops.resize_images(x.contiguous(), n.attrs["oh"], n.attrs["ow"], n.attrs["method"])
        raise G.GraphError("training: no lowering for %r" % (n,))

    def _conv_layer(self, chain, ev):
        conv, bias, bn, act, leak = chain
        x = ev(conv.inputs[0])
        w = self.params[conv.inputs[1].attrs["var"]]
        b = self.params[bias.inputs[1].attrs["var"]] if bias is not None else None
        is_fc = conv.op == "matmul"
        if is_fc:
            nrow = x.shape[0]
            x = x.reshape(nrow, 1, 1, -1)
            w4 = w.reshape(1, 1, w.shape[0], w.shape[1])
            stride = (1, 1)
        else:
            w4, stride = w, tuple(conv.attrs["stride"])
            if stride == (2, 2) and tuple(w.shape[:2]) == (4, 4) and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 \
                    and self.strided_on_mfma:
                x, w4, stride = space_to_depth2(x), strided4_as_3x3(w), (1, 1)
        cfg = {"stride": stride, "wscale": conv.attrs["wscale"], "act": act, "leak": leak, "prec": self.prec, "fc": is_fc,
               "eps": bn.attrs["eps"] if bn is not None else 0.0}
        wname = conv.inputs[1].attrs["var"]
        if any(wname.startswith(p) for p in self.higher_order_scopes):
            if bn is not None:
                raise G.GraphError("batch norm inside a gradient-penalty network is not built (the 8x discriminators have none)")
            y = ConvFn.apply(x, w4, b, cfg)
            if act is not None:
                if act == "tanh":
                    raise G.GraphError("tanh has no higher-order lowering")
                y = Act2Fn.apply(y, act, leak)
            return y.reshape(y.shape[0], -1) if is_fc else y
        gamma = beta = None
        if bn is not None:
            if not bn.attrs["training"]:
                raise G.GraphError("TrainSession needs batch_norm(training=True) nodes (build the nets with train=True)")
            gamma = self.params[bn.inputs[1].attrs["var"]]
            beta = self.params[bn.inputs[2].attrs["var"]]
            cfg["moving"] = (self.params[bn.inputs[3].attrs["var"]], self.params[bn.inputs[4].attrs["var"]])
            cfg["decay"] = self.bn_decay
        y = ConvLayerFn.apply(x, w4, b, gamma, beta, cfg)
        return y.reshape(y.shape[0], -1) if is_fc else y


class AdamTF(object):
    """tf.train.AdamOptimizer(lr, beta1, beta2=0.999, epsilon=1e-8) over a set of leaf tensors, one
    flat fp32 buffer per optimiser: p -= lr_t * m / (sqrt(v) + eps) (``mpg_adam_step``)."""

    def __init__(self, params, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8, comm=None):
        self.comm = comm        # dist.Comm: gradients are averaged over the ranks before the update (one bucket)
        self.names = sorted(params)
        self.params = [params[n] for n in self.names]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)       # parameters alias the flat buffer
                off += k
        self._grad_views, off = [], 0
        for p in self.params:
            self._grad_views.append(self.grad[off:off + p.numel()].view(p.shape))
            off += p.numel()
        self.lr, self.b1, self.b2, self.eps, self.t = lr, beta1, beta2, eps, 0
        self.lr_t = torch.zeros(1, dtype=torch.float32, device=dev)      # read by the kernel: replayable in a graph

    def advance(self, lr=None):
        """host side of a step: t += 1 and the bias-corrected step size into device memory"""
        self.t += 1
        lr = self.lr if lr is None else lr
        self.lr_t.fill_(lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t))

    def step(self, grads, lr=None, advance=True):
        """grads: list aligned with self.params (None = zero gradient, as tf treats unconnected variables)"""
        if advance:
            self.advance(lr)
        if any(g is None for g in grads):
            self.grad.zero_()
        dst = [v for v, g in zip(self._grad_views, grads) if g is not None]
        torch._foreach_copy_(dst, [g.contiguous() for g in grads if g is not None])
        if self.comm is not None:
            self.comm.all_reduce_mean(self.grad)
        train_ops.adam_step(self.flat, self.grad, self.m, self.v, self.lr_t, self.b1, self.b2, self.eps)

    def slot_state(self, tag):
        """the optimiser's slots under the names tf.train.Saver gives them (``<variable>/Adam``, ``<variable>/Adam_1``;
        the step count as TF's ``beta1_power`` = beta1^(t+1), prefixed with `tag` because every optimiser of a graph
        owns one): what a checkpoint needs so that a resumed run continues like the reference's Saver restore"""
        out, off = {}, 0
        m, v = self.m.cpu().numpy(), self.v.cpu().numpy()
        for n, p_ in zip(self.names, self.params):
            k = p_.numel()
            out[n + "/Adam"] = m[off:off + k].reshape(tuple(p_.shape))
            out[n + "/Adam_1"] = v[off:off + k].reshape(tuple(p_.shape))
            off += k
        out[tag + "/beta1_power"] = np.float32(self.b1 ** (self.t + 1))
        out[tag + "/adam_t"] = np.int64(self.t)
        return out

    def load_slot_state(self, state, tag):
        """inverse of slot_state; variables without slots in `state` keep theirs.  Returns the number restored."""
        off, hit = 0, 0
        with torch.no_grad():
            for n, p_ in zip(self.names, self.params):
                k = p_.numel()
                if n + "/Adam" in state and n + "/Adam_1" in state:
                    self.m[off:off + k].copy_(torch.as_tensor(np.asarray(state[n + "/Adam"], np.float32).reshape(-1)))
                    self.v[off:off + k].copy_(torch.as_tensor(np.asarray(state[n + "/Adam_1"], np.float32).reshape(-1)))
                    hit += 1
                off += k
        if tag + "/adam_t" in state:
            self.t = int(state[tag + "/adam_t"])
        return hit


def stage_variable_names(names, stage, levels):
    """the variables stage z of the growing schedule optimises (multipassGAN-8x.py:1316-1321,1331-1336,1347-1352): all of
    them at the last stage, otherwise those whose TF name contains one of "1", "2", .., "2^(z+1)" as a substring"""
    if stage >= levels - 1:
        return list(names)
    keys = ["%i" % (2 ** i) for i in range(stage + 2)]
    return [n for n in names if any(k in n for k in keys)]


class StagedAdam(AdamTF):
    """The optimisers of one network of the 8x training graph (multipassGAN-8x.py:1305-1362): per growing stage an Adam
    with its OWN moments and step count over that stage's variable subset (a fresh one takes over at every stage change;
    `copyAdamVariables` (:1783-1802) builds assign ops it never runs and then re-initialises the new stage's slots, i.e.
    changes nothing), optional dynamic loss scaling (:490-541: the loss is multiplied by 2^ls_var, gradients by
    2^-ls_var / len(grads); a non-finite gradient skips the update and lowers ls_var by 1, an applied one raises it by
    0.0005), and for the generator the MovingAverageOptimizer shadows of each stage (:1356).
    One flat parameter buffer; the stage's subset is a 0/1 mask over it; every decision is taken on the device."""

    LS_INIT, LS_INC, LS_DEC = 64.0, 0.0005, 1.0

    def __init__(self, params, levels, lr=1e-4, beta1=0.0, beta2=0.99, eps=1e-8, comm=None, loss_scaling=False,
                 ema_decay=None):
        super().__init__(params, lr, beta1, beta2, eps, comm)
        dev = self.flat.device
        self.levels, self.loss_scaling, self.ema_decay = levels, bool(loss_scaling), ema_decay
        n = self.flat.numel()
        self.masks, self.counts = [], []
        for z in range(levels):
            chosen = set(stage_variable_names(self.names, z, levels))
            mask = torch.zeros(n, dtype=torch.float32, device=dev)
            off = 0
            for name, p in zip(self.names, self.params):
                if name in chosen:
                    mask[off:off + p.numel()] = 1.0
                off += p.numel()
            self.masks.append(None if len(chosen) == len(self.names) else mask)
            self.counts.append(len(chosen))
        self.ms = [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(levels)]
        self.vs = [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(levels)]
        # per stage: [ls_var, coef, ok, t, lr_t, -, -, -]; ls_var is ONE variable per network in the reference (:501-503)
        self.state = [torch.zeros(8, dtype=torch.float32, device=dev) for _ in range(levels)]
        for st in self.state:
            st[0] = self.LS_INIT
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self.shadows = None
        if ema_decay is not None:       # shadows start from the variables' initial values
            self.shadows = [self.flat.detach().clone() for _ in range(levels)]
        self._last_stage = None

    def _enter(self, stage):
        """the network has ONE ls_var (:501-503): when another stage's optimiser takes over, the value moves with it"""
        stage = self.levels - 1 if stage is None else int(stage)
        if self.loss_scaling and self._last_stage is not None and stage != self._last_stage:
            self.state[stage][0] = self.state[self._last_stage][0]
        self._last_stage = stage
        return stage

    def loss_scale(self, stage):
        """2^ls_var as a device scalar (apply_loss_scaling, :506-507); None when loss scaling is off"""
        if not self.loss_scaling:
            return None
        return torch.exp(self.state[self._enter(stage)][0] * math.log(2.0))

    def step(self, grads, lr=None, stage=None):
        stage = self._enter(stage)
        if lr is not None:
            self.lr = lr
        self.lr_dev.fill_(float(self.lr))
        if any(g is None for g in grads):
            self.grad.zero_()
        dst = [v for v, g in zip(self._grad_views, grads) if g is not None]
        torch._foreach_copy_(dst, [g.contiguous() for g in grads if g is not None])
        if self.comm is not None:
            self.comm.all_reduce_mean(self.grad)
        train_ops.adam_step_staged(self.flat, self.grad, self.ms[stage], self.vs[stage], self.masks[stage], self.state[stage],
                                   self.lr_dev, self.counts[stage], self.loss_scaling, self.b1, self.b2, self.eps,
                                   self.LS_INC, self.LS_DEC, None if self.shadows is None else self.shadows[stage],
                                   self.ema_decay if self.ema_decay is not None else 0.0)

    def slot_state(self, tag):
        """every stage's moments (TF uniquifies the slot names of the z-th optimiser of a variable: ``Adam_<2z>`` /
        ``Adam_<2z+1>``, the first pair without / with ``_1``), step counts and ls_var"""
        out = {}
        for z in range(self.levels):
            m, v = self.ms[z].cpu().numpy(), self.vs[z].cpu().numpy()
            chosen, off = set(stage_variable_names(self.names, z, self.levels)), 0
            for n, p_ in zip(self.names, self.params):
                k = p_.numel()
                if n in chosen:
                    out[n + self._slot(2 * z)] = m[off:off + k].reshape(tuple(p_.shape))
                    out[n + self._slot(2 * z + 1)] = v[off:off + k].reshape(tuple(p_.shape))
                off += k
            st = self.state[z].cpu().numpy()
            out["%s/stage%d/adam_t" % (tag, z)] = np.int64(st[3])
            out["%s/stage%d/ls_var" % (tag, z)] = np.float32(st[0])
        return out

    @staticmethod
    def _slot(i):
        return "/Adam" if i == 0 else "/Adam_%d" % i

    def load_slot_state(self, state, tag):
        hit = 0
        with torch.no_grad():
            for z in range(self.levels):
                off = 0
                for n, p_ in zip(self.names, self.params):
                    k = p_.numel()
                    km, kv = n + self._slot(2 * z), n + self._slot(2 * z + 1)
                    if km in state and kv in state:
                        self.ms[z][off:off + k].copy_(torch.as_tensor(np.asarray(state[km], np.float32).reshape(-1)))
                        self.vs[z][off:off + k].copy_(torch.as_tensor(np.asarray(state[kv], np.float32).reshape(-1)))
                        hit += 1
                    off += k
                if "%s/stage%d/adam_t" % (tag, z) in state:
                    self.state[z][3] = float(state["%s/stage%d/adam_t" % (tag, z)])
                    self.state[z][0] = float(state["%s/stage%d/ls_var" % (tag, z)])
        return hit

    def ema_params(self, stage=None):
        """name -> moving-average tensor of the stage's shadows (swapping_saver of the last stage's optimiser, :1369)"""
        stage = self.levels - 1 if stage is None else stage
        out, off = {}, 0
        for name, p in zip(self.names, self.params):
            out[name] = self.shadows[stage][off:off + p.numel()].view(p.shape)
            off += p.numel()
        return out


def sigmoid_ce(logits, label):
    """tf.nn.sigmoid_cross_entropy_with_logits, mean over the batch: max(x,0) - x*z + log(1+exp(-|x|))"""
    return (torch.clamp(logits, min=0) - logits * label + torch.log1p(torch.exp(-logits.abs()))).mean()


class Trainer4x(object):
    """The GAN training iteration of multipassGAN-4x.py (graph :728-768,880-902, loop :1317-1356):
    ``discRuns`` discriminator updates then ``genRuns`` generator updates on tile batches, spatial
    discriminator with feature losses, sigmoid cross entropy + lambda * L1 (+ lambda2 * layer loss)."""

    def __init__(self, tileSizeLow=16, upRes=4, n_inputChannels=4, batch_norm=True, upsampling_mode=2, device="cuda:0",
                 learning_rate=2e-4, beta1=0.5, lambda_l1=1.0, lambda2=0.0, lambda2_l=(1.0, 1.0, 1.0, 1.0),
                 weight_dld=1.0, bn_decay=0.999, variables=None, prec=ops.PREC_F16X3, seed=777, comm=None,
                 use_tempo=False, lambda_t=1.0, adv_flag=True, clamping=True, lambda_t_l2=0.0):
        from . import arch
        from .session import VariableStore
        self.tileSizeLow, self.upRes, self.C = tileSizeLow, upRes, n_inputChannels
        self.tileSizeHigh = tileSizeLow * upRes
        self.n_input = tileSizeLow * tileSizeLow * n_inputChannels
        if upsampling_mode in (1, 3):
            self.n_input = self.tileSizeHigh * self.tileSizeHigh * n_inputChannels
        self.n_output = self.tileSizeHigh * self.tileSizeHigh
        self.k, self.k2, self.k2_l, self.weight_dld = lambda_l1, lambda2, lambda2_l, weight_dld
        g = G.reset_default_graph()
        self.graph = g
        self.x = G.placeholder([None, self.n_input], name="x")
        self.x_disc = G.placeholder([None, self.n_input], name="x_disc")
        self.y = G.placeholder([None, self.n_output], name="y")
        self.gen_part = arch.gen_resnet(self.x, tileSizeLow, upRes, n_inputChannels, upsampling_mode=upsampling_mode,
                                        use_batch_norm=batch_norm, train=True)
        dkw = dict(tileSizeLow=tileSizeLow, upRes=upRes, n_input=self.n_input, n_inputChannels=n_inputChannels,
                   upsampling_mode=upsampling_mode, use_batch_norm=batch_norm, train=True, bn_decay=bn_decay)
        self.disc = arch.disc_binclass(self.x_disc, self.y, **dkw)
        self.gen = arch.disc_binclass(self.x_disc, self.gen_part, reuse=True, **dkw)
        # temporal discriminator (multipassGAN-4x.py:790-885): three advected frames as channels
        self.use_tempo, self.kt, self.adv_flag, self.clamping, self.n_t = use_tempo, lambda_t, adv_flag, clamping, 3
        # useTempoL2 (multipassGAN-4x.py:147-152,815-826): the l2 distance of consecutive advected generator frames
        self.ktl = float(lambda_t_l2)
        self.use_tempo_l2 = self.ktl > 1e-6
        if use_tempo or self.use_tempo_l2:
            self.x_t = G.placeholder([None, self.n_input], name="x_t")
            self.gen_part_t = arch.gen_resnet(self.x_t, tileSizeLow, upRes, n_inputChannels, upsampling_mode=upsampling_mode,
                                              reuse=True, use_batch_norm=batch_norm, train=True)
        if use_tempo:
            self.t_fake = G.placeholder([None, self.n_output * self.n_t], name="t_fake")
            self.t_real = G.placeholder([None, self.n_output * self.n_t], name="t_real")
            tk = dict(tileSizeLow=tileSizeLow, upRes=upRes, n_t_channels=self.n_t, use_batch_norm=batch_norm, train=True,
                      bn_decay=bn_decay)
            self.gen_t = arch.disc_binclass_cond_tempo(self.t_fake, reuse=False, **tk)
            self.disc_t = arch.disc_binclass_cond_tempo(self.t_real, reuse=True, **tk)
        self.sess = TrainSession(variables or VariableStore(device, seed=seed), graph=g, prec=prec, bn_decay=bn_decay,
                                 device=device)
        self.g_var = self.sess.trainable("g_")
        self.d_var = self.sess.trainable("d_")
        self.opt_d = AdamTF(self.d_var, learning_rate, beta1, comm=comm)
        self.opt_g = AdamTF(self.g_var, learning_rate, beta1, comm=comm)
        if use_tempo:
            self.t_var = {n: p for n, p in self.sess.trainable("t_").items() if n.startswith("discriminatorTempo")}
            self.opt_t = AdamTF(self.t_var, learning_rate, beta1, comm=comm)

    def optimisers(self):
        """-> [(tag, AdamTF)] in the order the reference creates them (:812-900)"""
        return [("disc", self.opt_d), ("gen", self.opt_g)] + ([("tempo", self.opt_t)] if hasattr(self, "opt_t") else [])

    def set_learning_rate(self, lr):
        """the fed learning rate of all optimisers (the decayLR schedule of :773-774 is evaluated by the caller)"""
        for _, o in self.optimisers():
            o.lr = float(lr)

    def slot_state(self):
        out = {}
        for tag, o in self.optimisers():
            out.update(o.slot_state(tag))
        return out

    def load_slot_state(self, state):
        return sum(o.load_slot_state(state, tag) for tag, o in self.optimisers())

    def losses(self, batch_xs, batch_ys):
        """-> dict of the loss tensors of multipassGAN-4x.py:744-768 (one forward of G, D(real), D(fake))"""
        feeds = {self.x: batch_xs, self.x_disc: batch_xs, self.y: batch_ys}
        fetch = [self.gen_part] + list(self.disc) + list(self.gen)
        out = self.sess.run(fetch, feeds)
        gen_part, (disc, dy1, dy2, dy3, dy4), (gen, gy1, gy2, gy3, gy4) = out[0], out[1:6], out[6:11]
        y = torch.as_tensor(batch_ys, dtype=torch.float32, device=gen_part.device).reshape(gen_part.shape)
        ones, zeros = torch.ones_like(disc), torch.zeros_like(disc)
        L = {}
        L["disc_loss_disc"] = sigmoid_ce(disc, ones)
        L["disc_loss_gen"] = sigmoid_ce(gen, zeros)
        layer = 0.0
        for kf, a, b in zip(self.k2_l, (dy1, dy2, dy3, dy4), (gy1, gy2, gy3, gy4)):
            layer = layer + (kf * 0.5) * PairLossFn.apply(a, b, 1)   # tf.nn.l2_loss
        L["disc_loss_layer"] = layer
        L["disc_loss"] = L["disc_loss_disc"] * self.weight_dld + L["disc_loss_gen"]
        L["gen_loss"] = sigmoid_ce(gen, ones)
        L["gen_l2_loss"] = 0.5 * PairLossFn.apply(y, gen_part, 1)
        L["gen_l1_loss"] = PairLossFn.apply(y, gen_part, 0) / float(gen_part.numel())
        L["gen_loss_complete"] = L["gen_loss"] + L["gen_l1_loss"] * self.k + L["disc_loss_layer"] * self.k2
        L["gen_part"] = gen_part
        return L

    # ------------------------------------------------------------------ temporal branch
    def _frames_as_channels(self, frames, y_pos):
        """[3B, n_output] frame rows (+ look-up positions [3B, 2 n_output]) -> [B, n_output * 3]: advect every
        frame to the centre time with tensorResample when adv_flag (:801-812), then pack the n_t frames of a
        tile as channels (reshape [-1, n_t, n_output], transpose (0, 2, 1), :813-814)"""
        th = self.tileSizeHigh
        v = frames.reshape(-1, th, th, 1)
        if self.adv_flag:
            pos = torch.as_tensor(y_pos, dtype=torch.float32, device=v.device).reshape(-1, th, th, 2)
            v = ResampleFn.apply(v, pos, self.clamping)
        return v.reshape(-1, self.n_t, self.n_output).permute(0, 2, 1).reshape(-1, self.n_output * self.n_t)

    def tempo_losses(self, batch_xts, batch_yts, batch_y_pos=None):
        """-> t_disc_loss, t_gen_loss (:834-866) for a coherent batch from TileCreator.selectRandomTempoTiles"""
        dev = self.sess.device
        xts = torch.as_tensor(batch_xts, dtype=torch.float32, device=dev)
        yts = torch.as_tensor(batch_yts, dtype=torch.float32, device=dev)
        gen_part_t = self.sess.run([self.gen_part_t], {self.x_t: xts})[0]
        fake = self._frames_as_channels(gen_part_t, batch_y_pos)
        L = {}
        if self.use_tempo_l2:
            # tl_gen_loss = sum_i mean((frame_i - frame_i+1)^2) over the n_t advected generator frames (:821-826)
            f = fake.reshape(-1, self.n_output, self.n_t)
            fr = [f[:, :, i].contiguous() for i in range(self.n_t)]
            tl = None
            for i in range(self.n_t - 1):
                term = PairLossFn.apply(fr[i], fr[i + 1], 1) / float(fr[i].numel())
                tl = term if tl is None else tl + term
            L["tl_gen_loss"] = tl
        if not self.use_tempo:
            return L
        real = self._frames_as_channels(yts, batch_y_pos)
        gen_t, disc_t = self.sess.run([self.gen_t, self.disc_t], {self.t_fake: fake, self.t_real: real})
        L["t_disc_loss_disc"] = sigmoid_ce(disc_t, torch.ones_like(disc_t))
        L["t_disc_loss_gen"] = sigmoid_ce(gen_t, torch.zeros_like(gen_t))
        L["t_disc_loss"] = L["t_disc_loss_disc"] * self.weight_dld + L["t_disc_loss_gen"]
        L["t_gen_loss"] = sigmoid_ce(gen_t, torch.ones_like(gen_t))
        return L

    def tempo_disc_step(self, batch_xts, batch_yts, batch_y_pos=None, advance=True):
        L = self.tempo_losses(batch_xts, batch_yts, batch_y_pos)
        grads = torch.autograd.grad(L["t_disc_loss"], self.opt_t.params, allow_unused=True)
        self.opt_t.step(grads, advance=advance)
        return L

    def gen_step_tempo(self, batch_xs, batch_ys, batch_xts, batch_yts, batch_y_pos=None, advance=True):
        """generator update with the temporal term: gen_loss_complete + lambda_t * t_gen_loss (:866-868,897-899)"""
        L = self.losses(batch_xs, batch_ys)
        Lt = self.tempo_losses(batch_xts, batch_yts, batch_y_pos)
        L.update(Lt)
        if self.use_tempo_l2:
            L["gen_loss_complete"] = L["gen_loss_complete"] + self.ktl * Lt["tl_gen_loss"]      # :826
        if self.use_tempo:
            L["gen_loss_complete"] = L["gen_loss_complete"] + self.kt * Lt["t_gen_loss"]          # :866
        grads = torch.autograd.grad(L["gen_loss_complete"], self.opt_g.params, allow_unused=True)
        self.opt_g.step(grads, advance=advance)
        return L

    def disc_step(self, batch_xs, batch_ys, advance=True):
        L = self.losses(batch_xs, batch_ys)
        grads = torch.autograd.grad(L["disc_loss"], self.opt_d.params, allow_unused=True)
        self.opt_d.step(grads, advance=advance)
        return L

    def gen_step(self, batch_xs, batch_ys, advance=True):
        L = self.losses(batch_xs, batch_ys)
        grads = torch.autograd.grad(L["gen_loss_complete"], self.opt_g.params, allow_unused=True)
        self.opt_g.step(grads, advance=advance)
        return L

    def train_step_graphed(self, batch_xs, batch_ys):
        """the same iteration (discRuns = genRuns = 1) replayed from one captured hipGraph: ~1500 launches
        of the eager step become one graph launch.  Shapes are fixed by the first call; every later
        call copies the batch into the static inputs, bumps the Adam step sizes in device memory and
        replays.  Returns (disc_loss, gen_loss_complete) device scalars of the replayed iteration."""
        dev = self.opt_d.flat.device
        xs = torch.as_tensor(batch_xs, dtype=torch.float32, device=dev)
        ys = torch.as_tensor(batch_ys, dtype=torch.float32, device=dev)
        if getattr(self, "_graph", None) is None:
            self._gx, self._gy = xs.clone(), ys.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.train_step(self._gx, self._gy)        # allocator / lazy initialisation warm-up
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            self.opt_d.advance()
            self.opt_g.advance()
            with torch.cuda.graph(self._graph):
                Ld = self.disc_step(self._gx, self._gy, advance=False)
                Lg = self.gen_step(self._gx, self._gy, advance=False)
                self._gout = (Ld["disc_loss"].detach(), Lg["gen_loss_complete"].detach())
            self._graph.replay()       # capture does not execute: run the captured iteration once
            return self._gout
        if xs.shape != self._gx.shape or ys.shape != self._gy.shape:
            raise _lib.MpgError("train_step_graphed: batch shape changed after capture")
        self._gx.copy_(xs)
        self._gy.copy_(ys)
        self.opt_d.advance()
        self.opt_g.advance()
        self._graph.replay()
        return self._gout

    def train_step(self, batch_xs, batch_ys, discRuns=1, genRuns=1, tempo=None):
        """one iteration of the reference loop (:1317-1356): returns (disc_loss, gen_loss_complete) device scalars.
        tempo = (batch_xts, batch_yts, batch_y_pos) adds the temporal discriminator update and loss term."""
        for _ in range(discRuns):
            Ld = self.disc_step(batch_xs, batch_ys)
        if tempo is not None:
            for _ in range(discRuns):
                self.tempo_disc_step(*tempo)
        for _ in range(genRuns):
            Lg = self.gen_step_tempo(batch_xs, batch_ys, *tempo) if tempo is not None else self.gen_step(batch_xs, batch_ys)
        # detached: holding a loss would keep the tape (and its stream bookkeeping) alive across iterations
        return Ld["disc_loss"].detach(), Lg["gen_loss_complete"].detach()


class Trainer8x(object):
    """One stage of the progressive-growing training of multipassGAN-8x.py (graph :1023-1144, optimisers
    :1305-1362): WGAN-GP (lambda 10, target 1, epsilon penalty 1e-3) or LSGAN or sigmoid-CE losses, L1 and
    layer losses for the generator, Adam(beta1, beta2) per network, and the 0.999 moving average of the
    generator weights (tf.contrib.opt.MovingAverageOptimizer).  ``percentage`` in [0, log2(upRes)] is fed per
    step (3.0 = the final 8x stage)."""

    def __init__(self, cfg, device="cuda:0", learning_rate=1e-4, beta1=0.0, beta2=0.99, lambda_l1=1.0, lambda2=0.0,
                 k2_ls=None, weight_dld=1.0, use_wgan_gp=True, use_LSGAN=False, variables=None,
                 prec=ops.PREC_F16X3, seed=777, comm=None, ema_decay=0.999, use_tempo=False, lambda_t=1.0,
                 adv_flag=True, clamping=True, loss_scaling=False, adv_mode=0, batch_norm=False):
        from . import arch
        from .session import VariableStore
        self.cfg = cfg
        self.k, self.k2, self.weight_dld = lambda_l1, lambda2, weight_dld
        self.use_wgan_gp, self.use_LSGAN = use_wgan_gp, use_LSGAN
        # `batchNorm 1` / `use_mb_stddev 1` (multipassGAN-8x.py:85,149,847,907; both 0 in the reference runs): the kernels
        # and their gradients exist, but the WGAN-GP penalty differentiates the critic twice and the second derivative of
        # batch statistics / of the minibatch standard deviation is not built: LSGAN and sigmoid-CE losses only
        if use_wgan_gp and (batch_norm or cfg.use_mb_stddev):
            raise _lib.MpgError("batchNorm / use_mb_stddev with use_wgan_gp 1: the gradient penalty needs second derivatives "
                                "of the batch statistics, which are not built -- train with use_wgan_gp 0")
        self.batch_norm = bool(batch_norm)
        if use_wgan_gp:
            self.wgan_lambda, self.wgan_target, self.wgan_epsilon = (150.0, 30.0, 1e-3) if use_LSGAN else (10.0, 1.0, 1e-3)
        self.currentUpres = int(round(math.log(cfg.upRes, 2)))
        g = G.reset_default_graph()
        self.graph = g
        self.percentage = G.scalar_placeholder("percentage")
        self.x = G.placeholder([None, cfg.n_input], name="x")
        self.x_disc = G.placeholder([None, cfg.n_input], name="x_disc")
        self.y_gp = G.placeholder([None, cfg.n_output], name="y_gp")
        if cfg.upsampling_mode == 2:
            self.y2 = None
            self.y_in = G.placeholder([None, cfg.n_output], name="y_in")
            x_in = self.x
        else:       # later networks: `y` carries (target, previous pass) as two channels (:1041-1060)
            self.y2 = G.placeholder([None, cfg.n_output * 2], name="y")
            x_in, self.y_in = arch.later_network_input(self.x, self.y2, cfg)
        self.gen_y = arch.growing_gen(x_in, cfg, self.percentage, use_batch_norm=self.batch_norm, train=True,
                                      currentUpres=self.currentUpres)
        dk = dict(cfg=cfg, use_batch_norm=self.batch_norm, train=True, currentUpres=self.currentUpres)
        self.disc, self.f_y = arch.growing_disc(self.y_in, self.x_disc, self.percentage, reuse=False, **dk)
        self.gen, self.f_g = arch.growing_disc(self.gen_y, self.x_disc, self.percentage, reuse=True, **dk)
        self.d_out, _ = arch.growing_disc(self.y_gp, self.x_disc, self.percentage, reuse=True, **dk)
        self.k2_ls = list(k2_ls) if k2_ls is not None else [1.0] * len(self.f_y)
        # temporal discriminator (multipassGAN-8x.py:1158-1300; final growing stage, tensorResample advection)
        self.use_tempo, self.kt, self.adv_flag, self.clamping, self.n_t = use_tempo, lambda_t, adv_flag, clamping, 3
        # 0: tensorResample on positions computed by the tile creator; 1 / 2: GAN.advect on the low-res velocity channels
        # of x_t, semi-Lagrangian / MacCormack (:1193-1199,1221-1225)
        self.adv_mode = int(adv_mode)
        if use_tempo and adv_flag and self.adv_mode and cfg.n_inputChannels != 4:
            raise _lib.MpgError("adv_mode %d reads the velocity from channels 1..3 of a 4-channel low-res tile (:1187), "
                                "got %d channels" % (self.adv_mode, cfg.n_inputChannels))
        if use_tempo:
            self.x_t = G.placeholder([None, cfg.n_input], name="x_t")
            if cfg.upsampling_mode == 2:
                self.y_t2, x_t_in = None, self.x_t
            else:   # (:1171-1172) previous pass of the three frames = channel 1 of y_t
                self.y_t2 = G.placeholder([None, cfg.n_output * 2], name="yt")
                x_t_in, _ = arch.later_network_input(self.x_t, self.y_t2, cfg)
            self.gen_ts = arch.growing_gen(x_t_in, cfg, self.percentage, reuse=True, use_batch_norm=self.batch_norm, train=True,
                                             currentUpres=self.currentUpres)
            tk = dict(cfg=cfg, n_t_channels=self.n_t, use_batch_norm=self.batch_norm, train=True, currentUpres=self.currentUpres)
            self.t_fake = G.placeholder([None, cfg.n_output * self.n_t], name="t_fake")
            self.t_real = G.placeholder([None, cfg.n_output * self.n_t], name="t_real")
            self.t_gp = G.placeholder([None, cfg.n_output * self.n_t], name="t_gp")
            self.gen_s = arch.growing_disc_tempo(self.t_fake, self.percentage, reuse=False, **tk)
            self.disc_s = arch.growing_disc_tempo(self.t_real, self.percentage, reuse=True, **tk)
            self.t_out = arch.growing_disc_tempo(self.t_gp, self.percentage, reuse=True, **tk)
        self.sess = TrainSession(variables or VariableStore(device, seed=seed), graph=g, prec=prec, device=device)
        if use_wgan_gp:
            self.sess.higher_order_scopes = ("spatial-disc", "tempo-disc")
        self.g_var = self.sess.trainable("g_")
        self.d_var = self.sess.trainable("d_")
        lv = self.currentUpres
        self.loss_scaling = bool(loss_scaling)
        self.opt_d = StagedAdam(self.d_var, lv, learning_rate, beta1, beta2, comm=comm, loss_scaling=loss_scaling)
        self.opt_g = StagedAdam(self.g_var, lv, learning_rate, beta1, beta2, comm=comm, loss_scaling=loss_scaling,
                                ema_decay=ema_decay)
        if use_tempo:
            self.t_var = {n: p for n, p in self.sess.trainable("t_").items() if n.startswith("tempo-disc")}
            self.opt_t = StagedAdam(self.t_var, lv, learning_rate, beta1, beta2, comm=comm, loss_scaling=loss_scaling)
        self.ema_decay = ema_decay
        self.rng = torch.Generator(device="cpu").manual_seed(seed)

    # ------------------------------------------------------------------ losses
    def _adv(self, logits, real):
        """(discriminator term, generator term) of one critic output"""
        if self.use_LSGAN:
            tgt = 1.0 if real else 0.0
            return 0.5 * ((logits - tgt) ** 2).mean()
        if self.use_wgan_gp:
            return (-logits).mean() if real else logits.mean()
        return sigmoid_ce(logits, torch.ones_like(logits) if real else torch.zeros_like(logits))

    def losses(self, batch_xs, batch_ys, percentage=3.0, lerp_factor=None, need_gp=True):
        """-> dict of loss tensors (multipassGAN-8x.py:1082-1142); batch_ys at full tileSizeHigh resolution"""
        dev = self.sess.device
        xs = torch.as_tensor(batch_xs, dtype=torch.float32, device=dev)
        ys = torch.as_tensor(batch_ys, dtype=torch.float32, device=dev)
        if self.y2 is None:
            ys = self._to_full_res(ys)
            feeds = {self.x: xs, self.x_disc: xs, self.y_in: ys, self.percentage: percentage}
        else:
            feeds = {self.x: xs, self.x_disc: xs, self.y2: ys, self.percentage: percentage}
            ys = ys.reshape(-1, self.cfg.n_output, 2)[:, :, 0].contiguous()        # the target channel
        out = self.sess.run([self.gen_y, self.disc, self.gen] + list(self.f_y) + list(self.f_g), feeds)
        nf = len(self.f_y)
        gen_y, disc, gen = out[0], out[1], out[2]
        f_y, f_g = out[3:3 + nf], out[3 + nf:3 + 2 * nf]
        L = {"gen_y": gen_y}
        layer = 0.0
        for kf, a, b in zip(self.k2_ls, f_y, f_g):
            layer = layer + (kf * 0.5) * PairLossFn.apply(a, b, 1)
        L["disc_loss_layer"] = layer
        L["d_loss_y"], L["d_loss_g"] = self._adv(disc, True), self._adv(gen, False)
        disc_loss = L["d_loss_y"] * self.weight_dld + L["d_loss_g"]
        L["gen_l2_loss"] = 0.5 * PairLossFn.apply(ys, gen_y, 1)
        L["l1_loss"] = PairLossFn.apply(ys, gen_y, 0) / float(gen_y.numel())
        L["g_loss_d"] = self._adv(gen, True)
        if self.use_wgan_gp and need_gp:
            if lerp_factor is None:
                lerp_factor = torch.rand((xs.shape[0], 1), generator=self.rng)
            lf = torch.as_tensor(lerp_factor, dtype=torch.float32, device=dev).reshape(-1, 1)
            # d(penalty)/d(generator) is never used: the discriminator step only updates d_var (:1340-1350)
            y_gp = (lf * ys + (1.0 - lf) * gen_y.detach()).requires_grad_(True)
            d_out = self.sess.run([self.d_out], {self.x_disc: xs, self.y_gp: y_gp, self.percentage: percentage})[0]
            (grads_d,) = torch.autograd.grad(d_out.mean(), y_gp, create_graph=True)
            norm = torch.sqrt(((grads_d + 1e-4) ** 2).sum(dim=1))
            L["grad_penalty_d"] = (self.wgan_lambda * (norm - self.wgan_target) ** 2).mean()
            L["epsilon_penalty_d"] = (disc ** 2).mean()
            disc_loss = disc_loss + L["epsilon_penalty_d"] * self.wgan_epsilon + L["grad_penalty_d"]
        L["disc_loss"] = disc_loss
        L["gen_loss_complete"] = L["g_loss_d"] + L["l1_loss"] * self.k + L["disc_loss_layer"] * self.k2
        return L

    def _to_full_res(self, ys):
        """targets of an earlier growing stage come at tileSizeLow * 2^stage: nearest resize to tileSizeHigh
        (tf.image.resize_images(..., method=1), multipassGAN-8x.py:1055-1058)"""
        th = self.cfg.tileSizeHigh
        if ys.shape[1] == th * th:
            return ys
        cur = int(round(math.sqrt(ys.shape[1])))
        if cur * cur != ys.shape[1] or th % cur:
            raise _lib.MpgError("targets of %d values per tile do not fit tileSizeHigh %d" % (ys.shape[1], th))
        return ops.resize_nearest(ys.reshape(-1, cur, cur, 1).contiguous(), th, th).reshape(-1, th * th)

    # ------------------------------------------------------------------ temporal branch
    def _frames_as_channels(self, frames, y_pos, xts=None, cur_size=None):
        """advection look-up at the CURRENT stage's resolution (the positions come at tileSizeLow * 2^stage):
        generated frames are nearest-downsampled to it, resampled, and resized back (:1178-1200)"""
        th = self.cfg.tileSizeHigh
        frames = frames.reshape(frames.shape[0], -1)
        cur = int(round(math.sqrt(frames.shape[1])))
        if self.adv_flag:
            if self.adv_mode:
                tl = self.cfg.tileSizeLow
                pc = cur_size                                # currentTileSizeX (:1180-1184): the fed targets' resolution
                pos = None
            else:
                pos = torch.as_tensor(y_pos, dtype=torch.float32, device=frames.device)
                pc = int(round(math.sqrt(pos.shape[1] // 2)))
            v = frames.reshape(-1, cur, cur, 1)
            if cur != pc:                                   # generator output (full size) -> current size
                k = cur // pc
                v = v[:, ::k, ::k, :].contiguous()
            if pos is None:
                vel_t = xts.reshape(-1, tl, tl, 4)[..., 1:4].contiguous()
                n = v.shape[0]                              # startBz = (batch // 3) * 3 = the rows of a tempo batch
                v = train_ops.advect(v, vel_t, torch.zeros_like(v), 0.5, self.adv_mode, 1.0, start_bz=n)
            else:
                v = ResampleFn.apply(v, pos.reshape(-1, pc, pc, 2), self.clamping)
            if pc != th:
                v = (ResizeNearest2Fn if self.use_wgan_gp else ResizeNearestFn).apply(v, th, th)
        else:
            v = self._to_full_res(frames).reshape(-1, th, th, 1)
        return v.reshape(-1, self.n_t, self.cfg.n_output).permute(0, 2, 1).reshape(-1, self.cfg.n_output * self.n_t)

    def tempo_losses(self, batch_xts, batch_yts, batch_y_pos=None, percentage=3.0, lerp_factor=None, need_gp=True):
        """t_disc_loss / g_loss_t of multipassGAN-8x.py:1216-1300 for [3B, .] coherent frame rows (final stage)"""
        dev = self.sess.device
        xts = torch.as_tensor(batch_xts, dtype=torch.float32, device=dev)
        yts = torch.as_tensor(batch_yts, dtype=torch.float32, device=dev)
        if self.y_t2 is None:
            gen_ts = self.sess.run([self.gen_ts], {self.x_t: xts, self.percentage: percentage})[0]
        else:       # rows of (target, previous pass) pairs: the real frames are channel 0 (:1218-1219)
            gen_ts = self.sess.run([self.gen_ts], {self.x_t: xts, self.y_t2: yts, self.percentage: percentage})[0]
            yts = yts.reshape(-1, self.cfg.n_output, 2)[:, :, 0].contiguous()
        # resolution of the current growing stage = that of the fed targets (tileSizeLow * 2^ceil(percentage), :1180-1181)
        cur_size = int(round(math.sqrt(yts.reshape(yts.shape[0], -1).shape[1]))) if self.cfg.upsampling_mode == 2 else self.cfg.tileSizeHigh
        fake = self._frames_as_channels(gen_ts, batch_y_pos, xts, cur_size)
        real = self._frames_as_channels(yts, batch_y_pos, xts, cur_size)
        gen_s, disc_s = self.sess.run([self.gen_s, self.disc_s], {self.t_fake: fake, self.t_real: real,
                                                                   self.percentage: percentage})
        L = {"t_loss_y": self._adv(disc_s, True), "t_loss_g": self._adv(gen_s, False)}
        t_disc_loss = L["t_loss_y"] * self.weight_dld + L["t_loss_g"]
        if self.use_wgan_gp and need_gp:
            if lerp_factor is None:
                lerp_factor = torch.rand((fake.shape[0], 1), generator=self.rng)
            lf = torch.as_tensor(lerp_factor, dtype=torch.float32, device=dev).reshape(-1, 1)
            y_gp = (lf * real + (1.0 - lf) * fake.detach()).requires_grad_(True)
            t_out = self.sess.run([self.t_out], {self.t_gp: y_gp, self.percentage: percentage})[0]
            (grads_t,) = torch.autograd.grad(t_out.mean(), y_gp, create_graph=True)
            # [B, n_output, n_t]: the norm runs over the pixels of each frame (reduce_sum(axis=1), :1287)
            gt = grads_t.reshape(-1, self.cfg.n_output, self.n_t)
            norm = torch.sqrt(((gt + 1e-4) ** 2).sum(dim=1))
            L["grad_penalty_t"] = (self.wgan_lambda * (norm - self.wgan_target) ** 2).mean()
            L["epsilon_penalty_t"] = (disc_s ** 2).mean()
            t_disc_loss = t_disc_loss + L["epsilon_penalty_t"] * self.wgan_epsilon + L["grad_penalty_t"]
        L["t_disc_loss"] = t_disc_loss
        L["g_loss_t"] = self._adv(gen_s, True)
        return L

    def optimisers(self):
        return [("disc", self.opt_d), ("gen", self.opt_g)] + ([("tempo", self.opt_t)] if hasattr(self, "opt_t") else [])

    def slot_state(self):
        out = {}
        for tag, o in self.optimisers():
            out.update(o.slot_state(tag))
        return out

    def load_slot_state(self, state):
        return sum(o.load_slot_state(state, tag) for tag, o in self.optimisers())

    @property
    def ema(self):
        """moving averages of the generator variables as the last stage's optimiser keeps them (the model_ema checkpoint)"""
        return list(self.opt_g.ema_params().values())

    def _update(self, opt, loss, stage):
        """calc_gradients + apply_updates (:510-541): d(loss * 2^ls_var) for ALL of the network's variables; the stage's
        mask selects the ones its optimiser owns (unconnected ones count as zeros, :516)"""
        scale = opt.loss_scale(opt.levels - 1 if stage is None else stage)
        grads = torch.autograd.grad(loss if scale is None else loss * scale, opt.params, allow_unused=True)
        opt.step(grads, stage=stage)

    def tempo_disc_step(self, batch_xts, batch_yts, batch_y_pos=None, percentage=3.0, lerp_factor=None, stage=None):
        L = self.tempo_losses(batch_xts, batch_yts, batch_y_pos, percentage, lerp_factor)
        self._update(self.opt_t, L["t_disc_loss"], stage)
        return L

    def disc_step(self, batch_xs, batch_ys, percentage=3.0, lerp_factor=None, stage=None):
        L = self.losses(batch_xs, batch_ys, percentage, lerp_factor)
        self._update(self.opt_d, L["disc_loss"], stage)
        return L

    def gen_step(self, batch_xs, batch_ys, percentage=3.0, tempo=None, stage=None):
        L = self.losses(batch_xs, batch_ys, percentage, need_gp=False)
        if tempo is not None:
            Lt = self.tempo_losses(tempo[0], tempo[1], tempo[2], percentage, need_gp=False)
            L.update(Lt)
            L["gen_loss_complete"] = L["gen_loss_complete"] + self.kt * Lt["g_loss_t"]        # :1302
        self._update(self.opt_g, L["gen_loss_complete"], stage)     # the moving averages move inside the optimiser call
        return L

    def train_step(self, batch_xs, batch_ys, percentage=3.0, discRuns=1, genRuns=1, tempo=None, stage=None):
        """stage: index of the growing stage's optimisers, log2(currentUpres) - 1 (:1978); None = the last one"""
        for _ in range(discRuns):
            Ld = self.disc_step(batch_xs, batch_ys, percentage, stage=stage)
        if tempo is not None:
            for _ in range(discRuns):
                self.tempo_disc_step(tempo[0], tempo[1], tempo[2], percentage, stage=stage)
        for _ in range(genRuns):
            Lg = self.gen_step(batch_xs, batch_ys, percentage, tempo, stage=stage)
        return Ld["disc_loss"].detach(), Lg["gen_loss_complete"].detach()
