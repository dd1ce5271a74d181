"""numpy restatement of the reference's generator / discriminator graphs.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Parameters live in a
flat dict keyed by the TF variable path without the ``:0`` suffix, e.g.
``generator/g_cA0/weight`` (unscaled N(0,1) weight, GAN.py:668), ``.../bias``
(GAN.py:683) and, for the 4x nets, ``.../{gamma,beta,moving_mean,
moving_variance}`` (``tf.contrib.layers.batch_norm`` in the conv's own scope,
GAN.py:110).  ``ParamSource`` creates missing entries deterministically so the
same seeds give the same networks in the oracle and in the product.
"""
import math

import numpy as np

from . import ops

F32 = np.float32
SQRT2 = math.sqrt(2.0)


class ParamSource(object):
    """Holds / creates parameters.  Synthetic init per SURVEY.md section 8d:
    weight ~ N(0,1) (GAN.py:668), bias = 0.1 (GAN.py:683); BN gamma/beta/
    moving stats perturbed so that batch norm is not the identity."""

    def __init__(self, params=None, seed=777, bn_seed=4321):
        self.params = {} if params is None else params
        self.seed = seed
        self.bn_seed = bn_seed
        self.order = []

    def _rng(self, name, base):
        # name-keyed stream => independent of creation order
        h = 0
        for ch in name:
            h = (h * 131 + ord(ch)) % (2 ** 31 - 1)
        return np.random.default_rng([base, h])

    def get(self, name, shape, kind):
        if name not in self.params:
            if kind == "weight":
                v = self._rng(name, self.seed).standard_normal(shape).astype(F32)
            elif kind == "bias":
                v = np.full(shape, 0.1, dtype=F32)
            elif kind == "gamma":
                v = (1.0 + 0.1 * self._rng(name, self.bn_seed).standard_normal(shape)).astype(F32)
            elif kind == "beta":
                v = (0.1 * self._rng(name, self.bn_seed).standard_normal(shape)).astype(F32)
            elif kind == "moving_mean":
                v = (0.1 * self._rng(name, self.bn_seed).standard_normal(shape)).astype(F32)
            elif kind == "moving_variance":
                v = (1.0 + 0.2 * self._rng(name, self.bn_seed).random(shape)).astype(F32)
            else:
                raise ValueError(kind)
            self.params[name] = v
        v = self.params[name]
        assert tuple(v.shape) == tuple(shape), (name, v.shape, shape)
        if name not in self.order:
            self.order.append(name)
        return v


def conv_layer(ps, scope, x, cout, k, act=None, stride=1, batch_norm=False, gain=SQRT2):
    """``GAN.convolutional_layer`` (GAN.py:80-119): conv(SAME) -> +bias ->
    [BN inference] -> activation.  Returns (activated, linear)."""
    cin = x.shape[-1]
    shape = (k, k, cin, cout)
    w = ps.get(scope + "/weight", shape, "weight")
    b = ps.get(scope + "/bias", (cout,), "bias")
    w_eff = (w.astype(np.float64) * np.float64(ops.wscale(shape, gain))).astype(F32)
    y = ops.conv2d_same(x, w_eff, (stride, stride))
    y = ops.bias_add(y, b)
    if batch_norm:
        y = ops.batch_norm_infer(
            y,
            ps.get(scope + "/gamma", (cout,), "gamma"),
            ps.get(scope + "/beta", (cout,), "beta"),
            ps.get(scope + "/moving_mean", (cout,), "moving_mean"),
            ps.get(scope + "/moving_variance", (cout,), "moving_variance"),
        )
    return ops.activation(y, act), y


# ----------------------------------------------------------------------------
# 4x generator (multipassGAN-4x.py:505-569)
# ----------------------------------------------------------------------------
def res_block_4x(ps, x, rb_id, s1, s2, batch_norm, k=5, prefix="generator/"):
    """``resBlock`` of multipassGAN-4x.py:505-526:
    relu(convB(relu(convA(x))) + conv1x1(x))."""
    a, _ = conv_layer(ps, prefix + "g_cA%d" % rb_id, x, s1, k, "relu", 1, batch_norm)
    _, b = conv_layer(ps, prefix + "g_cB%d" % rb_id, a, s2, k, None, 1, batch_norm)
    _, s = conv_layer(ps, prefix + "g_s%d" % rb_id, x, s2, 1, None, 1, batch_norm)
    return ops.relu((b.astype(np.float64) + s.astype(np.float64)).astype(F32))


def gen_resnet(ps, x, up_res=4, upsampling_mode=2, batch_norm=True):
    """``gen_resnet`` (multipassGAN-4x.py:528-569).  x: [N,h,w,C] NHWC.
    Returns [N,H,W,1]; the reference flattens to [N,H*W] (line 566)."""
    c = x.shape[-1]
    if upsampling_mode == 2:
        inp = ops.max_depool(x, up_res, up_res)      # :554
    elif upsampling_mode in (1, 3):
        inp = x                                       # :556
    elif upsampling_mode == 0:
        inp = ops.max_depool(x, 1, up_res)           # :558
    else:
        raise ValueError(upsampling_mode)
    ru1 = res_block_4x(ps, inp, 0, c * 2, c * 8, batch_norm)   # :560
    ru2 = res_block_4x(ps, ru1, 1, 128, 128, batch_norm)       # :561
    ru3 = res_block_4x(ps, ru2, 2, 32, 8, batch_norm)          # :563
    ru4 = res_block_4x(ps, ru3, 3, 2, 1, False)                # :564
    return ru4


# ----------------------------------------------------------------------------
# 8x growing generator, output mode (multipassGAN-out.py:220-338)
# ----------------------------------------------------------------------------
def res_block_8x(ps, scope, x, s1, s2, name, k, pixel_norm=True, batch_norm=False):
    """``resBlock`` of multipassGAN-out.py:220-237.  Returns (result,
    gan_layer) where gan_layer is what ``GAN.layer`` holds afterwards: the
    last pixel_norm output, or the 1x1 shortcut conv when pixelNorm is off."""
    a, _ = conv_layer(ps, scope + "g_cA_" + name, x, s1, k, "relu", 1, batch_norm)
    if pixel_norm:
        a = ops.pixel_norm(a)
    _, b = conv_layer(ps, scope + "g_cB_" + name, a, s2, k, None, 1, batch_norm)
    _, s = conv_layer(ps, scope + "g_s_" + name, x, s2, 1, None, 1, batch_norm)
    r = ops.relu((b.astype(np.float64) + s.astype(np.float64)).astype(F32))
    layer = s
    if pixel_norm:
        r = ops.pixel_norm(r)
        layer = r
    return r, layer


def growing_gen(ps, x, up_res=8, first_gen=True, filter_size=3, start_fms=256, max_fms=256,
                first_nn_arch=False, use_res_net=True, pixel_norm=True, batch_norm=False,
                upsample_mode=1, add_bicubic_upsample=True, prefix="generator/"):
    """``growing_gen`` in output mode (``output=True``; multipassGAN-out.py:286-338)
    with ``growBlockGen`` (239-284).  x: [N,h,w,C] (first generator, h = low
    res) or [N,H,W,C+1] (later generators: previous pass density first)."""
    k = filter_size
    cur = int(round(math.log(up_res, 2)))
    gan_layer = x            # GAN(_in).layer, out.py:299
    if first_nn_arch:
        x_g = x
    elif use_res_net:
        half = min(max_fms, start_fms // 2)
        x_g, gan_layer = res_block_8x(ps, prefix, x, 16, half // 8, "1", k, pixel_norm, False)          # :308
        x_g, gan_layer = res_block_8x(ps, prefix, x_g, half // 4, half // 2, "2", k, pixel_norm, False)  # :309
    else:
        x_g, _ = conv_layer(ps, prefix + "g_cA1", x, 32, k, "lrelu", 1, batch_norm)                       # :311
        if pixel_norm:
            x_g = ops.pixel_norm(x_g)
        x_g, _ = conv_layer(ps, prefix + "g_cB1", x_g, min(start_fms // 2, max_fms), k, "lrelu", 1, batch_norm)  # :314
        if pixel_norm:
            x_g = ops.pixel_norm(x_g)
        gan_layer = x_g
    dens = None
    for j in range(1, cur + 1):
        fms = min(int(start_fms / (2 ** j)), max_fms)     # :320
        upres = 2 ** j
        scope = prefix + "genBlock%d/" % upres
        if first_gen:
            # gan.avg_depool(mode=upsampleMode) acts on GAN.layer (out.py:243, GAN.py:528-541)
            in_depool = ops.avg_depool(gan_layer, mode=upsample_mode, scale=(2,))
        else:
            in_depool = x_g
        if first_nn_arch:
            if upres == 2:
                names = ["first", "second", "third", "fourth", "fifth"]
                widths = [(fms, fms)] * 5
            elif upres == 4:
                names = ["first", "second", "third"]
                widths = [(fms * 2, fms), (fms, fms), (fms, fms)]
            else:
                names = ["first", "second"]
                widths = [(fms * 2, fms), (fms, fms)]
            outp = in_depool
            for nm, (s1, s2) in zip(names, widths):
                outp, gan_layer = res_block_8x(ps, scope, outp, s1, s2, nm, k, pixel_norm, batch_norm)
        elif use_res_net:
            outp, gan_layer = res_block_8x(ps, scope, in_depool, fms, fms, "first", k, pixel_norm, batch_norm)
            outp, gan_layer = res_block_8x(ps, scope, outp, fms // 2, fms // 2, "second", k, pixel_norm, batch_norm)
        else:
            a, _ = conv_layer(ps, scope + "g_cA%d" % upres, in_depool, fms, k, "lrelu", 1, batch_norm)   # :272
            if pixel_norm:
                a = ops.pixel_norm(a)
            outp, _ = conv_layer(ps, scope + "g_cB%d" % upres, a, fms, k, "lrelu", 1, batch_norm)        # :276
            if pixel_norm:
                outp = ops.pixel_norm(outp)
            gan_layer = outp
        x_g = outp
        if j == cur:
            dens, _ = conv_layer(ps, scope + "g_cdensOut%d" % upres, outp, 1, 1, None, 1, False, gain=1.0)  # :282
            if add_bicubic_upsample:
                if first_gen:
                    dens = (dens.astype(np.float64) + ops.avg_depool(x[..., 0:1], mode=2, scale=(2 ** j,))).astype(F32)  # :330
                else:
                    dens = (dens.astype(np.float64) + x[..., 0:1]).astype(F32)                                          # :332
    return dens


def gen2_input(y_prev, x_low, tile_high):
    """``x_in_2`` of multipassGAN-out.py:357: concat(previous pass density
    [N,H,W,1], nearest-resized low-res slice [N,H,W,C])."""
    up = ops.resize_nearest_tf1(x_low, tile_high, tile_high)
    return np.concatenate([y_prev, up], axis=3)


# ----------------------------------------------------------------------------
# 4x discriminators (multipassGAN-4x.py:572-662)
# ----------------------------------------------------------------------------
def disc_binclass(ps, in_low_density, in_high, up_res=4, upsampling_mode=2, batch_norm=True,
                  scope="discriminator/", pre="d"):
    """``disc_binclass`` (multipassGAN-4x.py:572-620).  in_low_density:
    [N,h,w,1]; in_high: [N,H,W,1].  Returns (logit, d1, d2, d3, d4)."""
    if upsampling_mode == 2:
        low = ops.max_depool(in_low_density, up_res, up_res)
    elif upsampling_mode == 0:
        low = ops.max_depool(in_low_density, 1, up_res)
    else:
        low = in_low_density
    x = np.concatenate([low, in_high], axis=-1)
    return _disc4_body(ps, x, batch_norm, scope, pre)


def disc_binclass_cond_tempo(ps, in_high3, batch_norm=True, scope="discriminatorTempo/", pre="t"):
    """``disc_binclass_cond_tempo`` (multipassGAN-4x.py:623-662); in: [N,H,W,3]."""
    return _disc4_body(ps, in_high3, batch_norm, scope, pre)[0]


def _disc4_body(ps, x, batch_norm, scope, pre):
    d1, _ = conv_layer(ps, scope + pre + "_c1", x, 32, 4, "lrelu", 2, False)
    d2, _ = conv_layer(ps, scope + pre + "_c2", d1, 64, 4, "lrelu", 2, batch_norm)
    d3, _ = conv_layer(ps, scope + pre + "_c3", d2, 128, 4, "lrelu", 2, batch_norm)
    d4, _ = conv_layer(ps, scope + pre + "_c4", d3, 256, 4, "lrelu", 1, batch_norm)
    flat = d4.reshape(d4.shape[0], -1)
    shape = (flat.shape[1], 1)
    w = ps.get(scope + pre + "_l5/weight", shape, "weight")
    b = ps.get(scope + pre + "_l5/bias", (1,), "bias")
    w_eff = (w.astype(np.float64) * np.float64(ops.wscale(shape, SQRT2))).astype(F32)
    return ops.fully_connected(flat, w_eff, b), d1, d2, d3, d4
