"""Network architectures as DATA, plus one small interpreter that emits them through the ``GAN`` builder.

The reference defines its networks as Python functions over TensorFlow (gen_resnet / disc_binclass*:
GAN/multipassGAN-4x.py:505-662; growing_gen / growBlockGen: GAN/multipassGAN-out.py:220-338 and
GAN/multipassGAN-8x.py:606-744; growing_disc*: GAN/multipassGAN-8x.py:752-923).  What the boundary
fixes is their RESULT: layer names (= checkpoint keys), widths, filter sizes, activation / norm
placement and the order variables are created in.  Here that result is written down as tables
(``*_table`` functions returning plain tuples) and ``emit_*`` walks a table.
``tests/test_abi_and_graph.py`` checks names, shapes and order against the oracle's parameter set.

Units of a table:
  ("res",  tag, a, b)      residual unit: cA<tag> kxk relu [pn] -> cB<tag> kxk linear, s<tag> 1x1 linear on the
                           unit's input, relu(B + s) [pn]                  (resBlock, -4x.py:505-526, -out.py:220-237)
  ("pair", tag, a, b)      two kxk lrelu convs cA<tag>, cB<tag>, each followed by [pn]        (-out.py:272-279,311-316)
"""
import math

from . import graph as tf          # the slice of the TF surface the emitters use
from .GAN import GAN, lrelu

RES, PAIR = "res", "pair"
ORDINALS = ("first", "second", "third", "fourth", "fifth")


def log2_int(v):
    return int(round(math.log(v, 2)))


# =====================================================================================================
# tables
# =====================================================================================================
def gen_resnet_table(c):
    """4x generator, k = 5 (-4x.py:560-564): (unit, follows the batchNorm flag)"""
    return ((RES, "0", 2 * c, 8 * c), True), ((RES, "1", 128, 128), True), ((RES, "2", 32, 8), True), \
           ((RES, "3", 2, 1), False)


# name suffix, width, stride, batch norm allowed (-4x.py:608-617, 648-657); then FC "<p>_l5"
DISC4_TABLE = (("c1", 32, 2, False), ("c2", 64, 2, True), ("c3", 128, 2, True), ("c4", 256, 1, True))


def level_fms(start_fms, max_fms, j):
    """feature maps of growing level j (-out.py:320, -8x.py:697,824)"""
    return min(int(start_fms / (2 ** j)), max_fms)


def growing_gen_table(start_fms, max_fms, levels, first_nn_arch, use_res_net):
    """(stem units, [(upres, units), ...]) of the 8x generator."""
    stem = ()
    if not first_nn_arch:
        if use_res_net:
            half = min(max_fms, start_fms // 2)
            stem = ((RES, "_1", 16, half // 8), (RES, "_2", half // 4, half // 2))           # -out.py:308-309
        else:
            stem = ((PAIR, "1", 32, min(start_fms // 2, max_fms)),)                            # -out.py:311-316
    blocks = []
    for j in range(1, levels + 1):
        f, up = level_fms(start_fms, max_fms, j), 2 ** j
        if first_nn_arch:                                                                      # -out.py:252-264
            count = {2: 5, 4: 3, 8: 2}.get(up, 2)
            wide = f if up == 2 else 2 * f
            units = tuple((RES, "_" + ORDINALS[i], wide if i == 0 else f, f) for i in range(count))
        elif use_res_net:                                                                      # -out.py:266-269
            units = ((RES, "_first", f, f), (RES, "_second", f // 2, f // 2))
        else:                                                                                  # -out.py:272-279
            units = ((PAIR, "%d" % up, f, f),)
        blocks.append((up, units))
    return stem, tuple(blocks)


def growing_disc_table(start_fms, max_fms, up_res, levels, first_nn_arch, filter_size, tall_filter=False):
    """per level j = levels..1 of the 8x critics (-8x.py:752-780,824-846):
    (upres, filter, A width, B width = width of the 1x1 skip from the pooled input)"""
    rows = []
    for j in range(levels, 0, -1):
        f, up = level_fms(start_fms, max_fms, j), 2 ** j
        b = min(min(2 * f, max_fms), start_fms // 2)
        if tall_filter:
            k = (filter_size + 2, filter_size)
        elif first_nn_arch:
            k = (4, 4)
        else:
            k = (filter_size, filter_size)
        a = (3 * f if up == 2 else 2 * f) if first_nn_arch else f
        rows.append((up, k, f, a, b))
    return int(start_fms / up_res), tuple(rows)


# =====================================================================================================
# interpreter
# =====================================================================================================
class _Emit(object):
    """walks units on one GAN object; `p` is the name prefix ("g")"""

    def __init__(self, gan, k, pn, reuse, train, p="g"):
        self.gan, self.k, self.pn, self.reuse, self.train, self.p = gan, [k, k], pn, reuse, train, p

    def conv(self, name, x, width, k, act, bn, **kw):
        return self.gan.convolutional_layer(width, k, act, stride=[1], name=name, in_layer=x, reuse=self.reuse,
                                            batch_norm=bn, train=self.train, **kw)

    def unit(self, x, unit, bn):
        kind, tag, a, b = unit
        gan, p = self.gan, self.p
        if kind == RES:
            h, _ = self.conv("%s_cA%s" % (p, tag), x, a, self.k, tf.relu, bn)
            if self.pn:
                h = gan.pixel_norm(h)
            main, _ = self.conv("%s_cB%s" % (p, tag), None, b, self.k, None, bn)       # reads gan.layer, like the reference
            skip, _ = self.conv("%s_s%s" % (p, tag), x, b, [1, 1], None, bn)
            out = tf.relu(tf.add(main, skip))
            return gan.pixel_norm(out) if self.pn else out      # without pn gan.layer stays the skip conv (a reference quirk)
        if kind == PAIR:
            h, _ = self.conv("%s_cA%s" % (p, tag), x, a, self.k, lrelu, bn)
            if self.pn:
                h = gan.pixel_norm(h)
            h, _ = self.conv("%s_cB%s" % (p, tag), h, b, self.k, lrelu, bn)
            return gan.pixel_norm(h) if self.pn else h
        raise ValueError("unknown unit kind %r" % (kind,))

    def units(self, x, units, bn):
        for u in units:
            x = self.unit(x, u, bn)
        return x


# ---------------------------------------------------------------------------------------------- 4x
def gen_resnet(_in, tileSizeLow, upRes, n_inputChannels, upsampling_mode=2, reuse=False, use_batch_norm=False,
               train=False):
    """-4x.py:528-569.  _in: flat placeholder [None, n_input]; returns [None, tileSizeHigh^2]."""
    high = tileSizeLow * upRes
    with tf.variable_scope("generator", reuse=reuse):
        rows = {2: tileSizeLow, 1: high, 3: high, 0: high}[upsampling_mode]
        cols = tileSizeLow if upsampling_mode in (2, 0) else high
        x = tf.reshape(_in, shape=[-1, rows, cols, n_inputChannels])
        gan = GAN(x)
        if upsampling_mode == 2:
            x = gan.max_depool(height_factor=upRes, width_factor=upRes)                         # :554
        elif upsampling_mode == 0:
            x = gan.max_depool(height_factor=1, width_factor=upRes)                             # :558
        em = _Emit(gan, 5, False, reuse, train)
        for unit, bn in gen_resnet_table(n_inputChannels):
            x = em.unit(x, unit, use_batch_norm and bn)
        return tf.reshape(x, shape=[-1, high * high])


def _disc4(x, p, reuse, use_batch_norm, train, bn_decay):
    gan = GAN(x, bn_decay=bn_decay)
    feats = []
    for name, width, stride, bn in DISC4_TABLE:
        f, _ = gan.convolutional_layer(width, [4, 4], lrelu, stride=[stride], name="%s_%s" % (p, name), reuse=reuse,
                                       batch_norm=use_batch_norm and bn, train=train)
        feats.append(f)
    gan.flatten()
    gan.fully_connected_layer(1, None, name="%s_l5" % p)
    return gan.y(), feats


def disc_binclass(in_low, in_high, tileSizeLow, upRes, n_input, n_inputChannels, upsampling_mode=2, reuse=False,
                  use_batch_norm=False, train=False, bn_decay=0.999):
    """-4x.py:572-620 (2D branch).  in_low: [None, n_input] generator input; in_high: [None, H*W].
    Returns (logit, d1, d2, d3, d4)."""
    high = tileSizeLow * upRes
    with tf.variable_scope("discriminator", reuse=reuse):
        # tf.slice(in_low, [0,0], [N, n_input/C]) (:583) keeps the FIRST n_input/C entries of the flat,
        # channel-interleaved row -- reproduced as written
        low = tf.slice_flat(in_low, n_input // n_inputChannels)
        if upsampling_mode == 2:
            low = GAN(tf.reshape(low, shape=[-1, tileSizeLow, tileSizeLow, 1])).max_depool(height_factor=upRes,
                                                                                           width_factor=upRes)
        elif upsampling_mode == 0:
            low = GAN(tf.reshape(low, shape=[-1, high, tileSizeLow, 1])).max_depool(height_factor=1, width_factor=upRes)
        else:
            low = tf.reshape(low, shape=[-1, high, high, 1])
        x = tf.concat([low, tf.reshape(in_high, shape=[-1, high, high, 1])], axis=-1)
        logit, feats = _disc4(x, "d", reuse, use_batch_norm, train, bn_decay)
        return (logit,) + tuple(feats)


def disc_binclass_cond_tempo(in_high, tileSizeLow, upRes, n_t_channels=3, reuse=False, use_batch_norm=False,
                             train=False, bn_decay=0.999):
    """-4x.py:622-659 (2D branch): n_t_channels advected frames packed as channels -> logit"""
    high = tileSizeLow * upRes
    with tf.variable_scope("discriminatorTempo", reuse=reuse):
        x = tf.reshape(in_high, shape=[-1, high, high, n_t_channels])
        return _disc4(x, "t", reuse, use_batch_norm, train, bn_decay)[0]


# ---------------------------------------------------------------------------------------------- 8x
class Cfg8x(object):
    """the module-level flags of multipassGAN-8x.py / -out.py that the model functions read"""

    def __init__(self, tileSizeLow=16, upRes=8, n_inputChannels=4, upsampling_mode=2, upsampleMode=1, filterSize=3,
                 start_fms=256, max_fms=256, first_nn_arch=True, use_res_net=True, pixel_norm=True,
                 addBicubicUpsample=True, use_mb_stddev=False, useVelInTDisc=False, bn_decay=0.999):
        self.tileSizeLow, self.upRes = tileSizeLow, upRes
        self.tileSizeHigh = tileSizeLow * upRes
        self.n_inputChannels = n_inputChannels
        self.upsampling_mode, self.upsampleMode = upsampling_mode, upsampleMode
        self.filterSize, self.start_fms, self.max_fms = filterSize, start_fms, max_fms
        self.first_nn_arch, self.use_res_net, self.pixel_norm = first_nn_arch, use_res_net, pixel_norm
        self.addBicubicUpsample, self.use_mb_stddev, self.useVelInTDisc = addBicubicUpsample, use_mb_stddev, useVelInTDisc
        self.bn_decay = bn_decay
        if upsampling_mode not in (1, 2, 3):
            raise NotImplementedError("upsampling_mode %d (only 1, 2, 3 are used by the example runs)" % upsampling_mode)
        self.n_input = tileSizeLow ** 2 * n_inputChannels            # -8x.py:402-416 (modes 1, 2, 3)
        self.n_output = self.tileSizeHigh ** 2

    @property
    def first_gen(self):
        return self.upsampling_mode == 2


def second_gen_input(x, y, tileSizeLow, tileSizeHigh, n_inputChannels):
    """x_in_2 of -out.py:357: concat(previous pass slice, nearest-resized low-res slice)."""
    up = tf.resize_images(tf.reshape(x, shape=[-1, tileSizeLow, tileSizeLow, n_inputChannels]),
                          [tileSizeHigh, tileSizeHigh], method=1)
    return tf.concat((tf.reshape(y, shape=[-1, tileSizeHigh, tileSizeHigh, 1]), up), axis=3)


def later_network_input(x, y2, cfg):
    """x_in of the second / third network in training (-8x.py:1041-1044): channel 1 of the two-channel `y`
    (the previous pass's output; channel 0 is the target) next to the nearest-resized low-res input"""
    c = cfg
    y4 = tf.reshape(y2, shape=[-1, c.tileSizeHigh, c.tileSizeHigh, 2])
    up = tf.resize_images(tf.reshape(x, shape=[-1, c.tileSizeLow, c.tileSizeLow, c.n_inputChannels]),
                          [c.tileSizeHigh, c.tileSizeHigh], method=1)
    x_in = tf.concat((tf.slice_channels(y4, 1, 1), up), axis=3)
    y_in = tf.reshape(tf.slice_channels(y4, 0, 1), shape=[-1, c.tileSizeHigh * c.tileSizeHigh])
    return x_in, y_in


def _density_head(x, upres, reuse, train, bn_decay):
    """g_cdensOut<upres>: 1x1 -> 1 channel, gain 1, on its own GAN object (-out.py:282, -8x.py:665,700)"""
    head, _ = GAN(x, bn_decay=bn_decay).convolutional_layer(1, [1, 1], None, stride=[1], name="g_cdensOut%d" % upres,
                                                            in_layer=x, reuse=reuse, batch_norm=False, train=train, gain=1)
    return head


def growing_gen(_in, cfg, percentage=None, reuse=False, use_batch_norm=False, train=False, currentUpres=None,
                output=None):
    """The 8x generator.  output=True (inference, -out.py:286-338): only the last level's density head.
    output=False (training, -8x.py:677-744): a head per level, faded in with lerp(old, new, percentage - (j-1)).
    _in: [N, h*w*C] flat or the 4D tensor second_gen_input / later_network_input built."""
    c = cfg
    output = (percentage is None) if output is None else output
    levels = log2_int(c.upRes) if currentUpres is None else currentUpres
    stem, blocks = growing_gen_table(c.start_fms, c.max_fms, levels, c.first_nn_arch, c.use_res_net)
    with tf.variable_scope("generator", reuse=reuse):
        if c.first_gen:
            x_in = tf.reshape(_in, shape=[-1, c.tileSizeLow, c.tileSizeLow, c.n_inputChannels])
        else:
            x_in = tf.reshape(_in, shape=[-1, c.tileSizeHigh, c.tileSizeHigh, c.n_inputChannels + 1])
        gan = GAN(x_in, bn_decay=c.bn_decay if not output else 0.0)
        em = _Emit(gan, c.filterSize, c.pixel_norm, reuse, train)
        # the stem's residual units never use batch norm (-out.py:308-309), its conv pair does (:311-314)
        x = x_in
        for u in stem:
            x = em.unit(x, u, use_batch_norm and u[0] == PAIR)
        old = None if output else _density_head(x, 1, reuse, train, c.bn_decay)
        dens = None
        for j, (up, units) in enumerate(blocks, start=1):
            last = j == len(blocks)
            with tf.variable_scope("genBlock%d" % up, reuse=reuse):
                if c.first_gen:
                    x = gan.avg_depool(mode=c.upsampleMode)      # acts on gan.layer (-out.py:243)
                x = em.units(x, units, use_batch_norm)
                if not output or last:
                    dens = _density_head(x, up, reuse, train, 0.0)
            if c.addBicubicUpsample and (not output or last):    # residual learning (-out.py:327-332, -8x.py:718-725)
                base = tf.slice_channels(x_in, 0, 1)
                dens = dens + (GAN(base).avg_depool(mode=2, scale=[up]) if c.first_gen else base)
            if not output:
                size = c.tileSizeLow * up if c.first_gen else c.tileSizeHigh
                if c.first_gen:
                    old = GAN(old).avg_depool(mode=1)
                old = tf.reshape(tf.lerp(old, dens, percentage - (j - 1)), shape=[-1, size, size, 1])
            elif last:
                old = dens
        size = int(old.get_shape()[1])
        return tf.reshape(old, shape=[-1, size * size])


def _critic(x_in, p, percentage, cfg, reuse, use_batch_norm, train, levels, mb_group):
    """shared body of growing_disc / growing_disc_tempo (-8x.py:814-863, 887-923): returns (score, features)"""
    c = cfg
    first_w, rows = growing_disc_table(c.start_fms, c.max_fms, c.upRes, levels, c.first_nn_arch, c.filterSize,
                                       tall_filter=(p == "t" and c.useVelInTDisc))
    gan = GAN(x_in, bn_decay=c.bn_decay)
    x, _ = gan.convolutional_layer(first_w, [1, 1], activation_function=None, in_layer=x_in, stride=[1],
                                   name="%s_cfromDensity%d" % (p, c.upRes), reuse=reuse, batch_norm=False, train=train)
    feats = [tf.lerp(None, x, percentage - (levels - 1))]
    raw = x_in
    skip_gan = GAN(raw, bn_decay=c.bn_decay)
    for j, (up, k, f, a, b) in zip(range(levels, 0, -1), rows):
        if c.first_gen:
            raw = GAN(raw).avg_pool()
        with tf.variable_scope("%sBlock%d" % (p, up), reuse=reuse):                            # growBlockDisc, :752-780
            kw = {} if c.first_nn_arch else {"in_channels": f}
            x1, _ = gan.convolutional_layer(a, list(k), lrelu, stride=[1], name="%s_cA%d" % (p, up), in_layer=x,
                                            reuse=reuse, batch_norm=False, train=train, in_channels=f)
            x2, _ = gan.convolutional_layer(b, list(k), lrelu, stride=[1], name="%s_cB%d" % (p, up), in_layer=x1,
                                            reuse=reuse, batch_norm=False, train=train, **kw)
            x = gan.avg_pool() if c.first_gen else x2
        skip, _ = skip_gan.convolutional_layer(b, [1, 1], None, stride=[1], name="%s_cfromDensity%d" % (p, up // 2),
                                               in_layer=raw, reuse=reuse, batch_norm=False, train=train)
        size = c.tileSizeLow * (up // 2) if c.first_gen else c.tileSizeHigh
        x = tf.reshape(tf.lerp(skip, x, percentage - (j - 1)), shape=[-1, size, size, b])
        feats.append(tf.lerp(None, x1, percentage - (j - 1)))
        feats.append(tf.lerp(None, x2, percentage - (j - 1)))
    if c.use_mb_stddev:
        x = gan.minibatch_stddev_layer(x, mb_group) if mb_group else gan.minibatch_stddev_layer(x)
    if not c.first_nn_arch:                                                                    # :852-854
        k = [c.filterSize, c.filterSize]
        x1, _ = gan.convolutional_layer(32, k, lrelu, stride=[1], name="%s_cA1" % p, in_layer=x, reuse=reuse,
                                        batch_norm=use_batch_norm, train=train)
        gan.convolutional_layer(4, k, None, stride=[1], name="%s_cB1" % p, in_layer=x1, reuse=reuse,
                                batch_norm=use_batch_norm, train=train)
    else:
        x1 = x
    feats.append(tf.lerp(None, x1, percentage))
    # the head reads gan.layer: with first_nn_arch that is the POOLED x2 of the last block, not the blended
    # tensor (which only feeds the feature list there); otherwise the <p>_cB1 output (:860-863)
    gan.flatten()
    gan.fully_connected_layer(1, None, name="%s_l61" % p, gain=1)
    return gan.y(), feats


def growing_disc(in_high_, in_low_, percentage, cfg, reuse=False, use_batch_norm=False, train=None, currentUpres=3):
    """-8x.py:783-863.  Returns (score [N,1], feature_layers)."""
    c = cfg
    with tf.variable_scope("spatial-disc", reuse=reuse):
        high = tf.reshape(in_high_, shape=[-1, c.tileSizeHigh, c.tileSizeHigh, 1])
        # every mode slices channel 0 of a [tileSizeLow, tileSizeLow, C] view and resizes it by upRes (:797-809)
        low = tf.slice_channels(tf.reshape(in_low_, shape=[-1, c.tileSizeLow, c.tileSizeLow, c.n_inputChannels]), 0, 1)
        low = GAN(tf.reshape(low, shape=[-1, c.tileSizeLow, c.tileSizeLow, 1])).avg_depool(scale=[c.upRes],
                                                                                           mode=c.upsampleMode)
        return _critic(tf.concat([low, high], axis=3), "d", percentage, c, reuse, use_batch_norm, train, currentUpres, 0)


def growing_disc_tempo(in_high_, percentage, cfg, n_t_channels=3, reuse=True, use_batch_norm=False, train=None,
                       currentUpres=3):
    """-8x.py:866-923 (useVelInTDisc 0): [N, H, W, 3] frame triples -> score"""
    c = cfg
    with tf.variable_scope("tempo-disc", reuse=reuse):
        x = tf.reshape(in_high_, shape=[-1, c.tileSizeHigh, c.tileSizeHigh, 12 if c.useVelInTDisc else 3])
        return _critic(x, "t", percentage, c, reuse, use_batch_norm, train, currentUpres, 1)[0]
