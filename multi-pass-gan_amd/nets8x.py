"""Training-mode wiring of the 8x progressive-growing networks (multipassGAN-8x.py:598-923), written as
the reference writes it against the ``GAN`` builder: the generator with its per-level density heads and
lerp fade-in, ``growBlockDisc``, ``growing_disc`` and ``growing_disc_tempo``.

``percentage`` is a ``graph.Scalar`` fed per iteration (tf.placeholder in the reference, :1018), so the
blend factors change without rebuilding the graph.  upsampling_mode 2 (first network: low-res input)
and 1 / 3 (later networks: high-res input with one extra channel) are built; mode 0 is not used by
the example runs.
"""
from . import graph as tf
from .GAN import GAN, lrelu
from .nets import growBlockGen, resBlock8x


def lerp(x, y, t):
    return tf.lerp(x, y, t)


class Cfg8x(object):
    """the module-level flags of multipassGAN-8x.py that the model functions read"""

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
        self.n_input = tileSizeLow ** 2 * n_inputChannels            # multipassGAN-8x.py:402-416 (modes 1, 2, 3)
        self.n_output = self.tileSizeHigh ** 2


def later_network_input(x, y2, cfg):
    """x_in of the second / third network (multipassGAN-8x.py:1041-1044): channel 1 of the two-channel `y`
    (the previous pass's output; channel 0 is the target) next to the nearest-resized low-res input"""
    c = cfg
    y4 = tf.reshape(y2, shape=[-1, c.tileSizeHigh, c.tileSizeHigh, 2])
    x_up = tf.resize_images(tf.reshape(x, shape=[-1, c.tileSizeLow, c.tileSizeLow, c.n_inputChannels]),
                            [c.tileSizeHigh, c.tileSizeHigh], method=1)
    x_in = tf.concat((tf.slice_channels(y4, 1, 1), x_up), axis=3)
    y_in = tf.reshape(tf.slice_channels(y4, 0, 1), shape=[-1, c.tileSizeHigh * c.tileSizeHigh])
    return x_in, y_in


def growing_gen(_in, percentage, cfg, reuse=False, use_batch_norm=False, train=None, currentUpres=3, output=False):
    """multipassGAN-8x.py:677-744"""
    c = cfg
    with tf.variable_scope("generator", reuse=reuse):
        if c.upsampling_mode == 2:
            _in = tf.reshape(_in, shape=[-1, c.tileSizeLow, c.tileSizeLow, c.n_inputChannels])
        else:
            _in = tf.reshape(_in, shape=[-1, c.tileSizeHigh, c.tileSizeHigh, c.n_inputChannels + 1])
        gan = GAN(_in, bn_decay=c.bn_decay)
        filter = [c.filterSize, c.filterSize]
        if c.first_nn_arch:
            x_g = _in
        elif c.use_res_net:
            half = min(c.max_fms, c.start_fms // 2)
            x_g = resBlock8x(gan, _in, 16, half // 8, reuse, False, "1", filter[0], c.pixel_norm, train)
            x_g = resBlock8x(gan, x_g, half // 4, half // 2, reuse, False, "2", filter[0], c.pixel_norm, train)
        else:
            x_g, _ = gan.convolutional_layer(32, filter, lrelu, stride=[1], name="g_cA%d" % (1), in_layer=_in,
                                             reuse=reuse, batch_norm=use_batch_norm, train=train)
            if c.pixel_norm:
                x_g = gan.pixel_norm(x_g)
            x_g, _ = gan.convolutional_layer(min(c.start_fms // 2, c.max_fms), filter, lrelu, stride=[1],
                                             name="g_cB%d" % (1), in_layer=x_g, reuse=reuse, batch_norm=use_batch_norm,
                                             train=train)
            if c.pixel_norm:
                x_g = gan.pixel_norm(x_g)
        _oldDens = None
        if not output:
            _oldDens, _ = GAN(x_g, bn_decay=c.bn_decay).convolutional_layer(
                1, [1, 1], None, stride=[1], name="g_cdensOut%d" % (1), in_layer=x_g, reuse=reuse, batch_norm=False,
                train=train, gain=1)
        _dens = None
        for j in range(1, currentUpres + 1):
            num_fms = min(int(c.start_fms / (2 ** j)), c.max_fms)
            firstGen = c.upsampling_mode == 2
            if not output or j == currentUpres:
                x_g, _dens = growBlockGen(gan, x_g, int(2 ** j), num_fms, use_batch_norm, train, reuse, False, firstGen,
                                          c.filterSize, c.first_nn_arch, c.use_res_net, c.pixel_norm, c.upsampleMode)
            else:
                x_g = growBlockGen(gan, x_g, int(2 ** j), num_fms, use_batch_norm, train, reuse, output, firstGen,
                                   c.filterSize, c.first_nn_arch, c.use_res_net, c.pixel_norm, c.upsampleMode)
            if c.addBicubicUpsample and (not output or j == currentUpres):      # residual learning (:718-725)
                if c.upsampling_mode == 2:
                    _dens = _dens + GAN(tf.slice_channels(_in, 0, 1)).avg_depool(mode=2, scale=[int(2 ** j)])
                else:
                    _dens = _dens + tf.slice_channels(_in, 0, 1)
            with tf.variable_scope("growingPart%i" % j, reuse=reuse):
                if not output:
                    if c.upsampling_mode == 2:
                        _oldDens = GAN(_oldDens).avg_depool(mode=1)
                        size = c.tileSizeLow * (2 ** j)
                    else:
                        size = c.tileSizeHigh
                    _oldDens = tf.reshape(lerp(_oldDens, _dens, percentage - (j - 1)), shape=[-1, size, size, 1])
                elif j == currentUpres:
                    _oldDens = _dens
        size = int(_oldDens.get_shape()[1])
        return tf.reshape(_oldDens, shape=[-1, size * size])


def growBlockDisc(gan, inp, upres, fms, use_batch_norm, train, reuse, name, cfg):
    """multipassGAN-8x.py:752-780 (gaussian drop-out layers are the identity: use_gdrop 0)"""
    c = cfg
    with tf.variable_scope(name + ("Block%d" % (upres)), reuse=reuse):
        if name == "t" and c.useVelInTDisc:
            filter = [c.filterSize + 2, c.filterSize]
        elif c.first_nn_arch:
            filter = [4, 4]
        else:
            filter = [c.filterSize, c.filterSize]
        fmsB = min(min(fms * 2, c.max_fms), c.start_fms // 2)
        if c.first_nn_arch:
            fmsA = fms * 3 if upres == 2 else fms * 2
            x1, _ = gan.convolutional_layer(fmsA, filter, lrelu, stride=[1], name=str(name) + "_cA%d" % (upres),
                                            in_layer=inp, reuse=reuse, batch_norm=use_batch_norm, train=train,
                                            in_channels=fms)
            x2, _ = gan.convolutional_layer(fmsB, filter, lrelu, stride=[1], name=str(name) + "_cB%d" % (upres),
                                            in_layer=x1, reuse=reuse, batch_norm=use_batch_norm, train=train)
        else:
            x1, _ = gan.convolutional_layer(fms, filter, lrelu, stride=[1], name=str(name) + "_cA%d" % (upres),
                                            in_layer=inp, reuse=reuse, batch_norm=use_batch_norm, train=train,
                                            in_channels=fms)
            x2, _ = gan.convolutional_layer(fmsB, filter, lrelu, stride=[1], name=str(name) + "_cB%d" % (upres),
                                            in_layer=x1, reuse=reuse, batch_norm=use_batch_norm, train=train,
                                            in_channels=fms)
        if c.upsampling_mode == 2:
            outp = gan.avg_pool()
        else:
            outp = x2
        return outp, x1, x2


def _disc_low_input(in_low_, cfg):
    c = cfg
    if c.upsampling_mode == 2:
        low = tf.slice_channels(tf.reshape(in_low_, shape=[-1, c.tileSizeLow, c.tileSizeLow, c.n_inputChannels]), 0, 1)
        return GAN(tf.reshape(low, shape=[-1, c.tileSizeLow, c.tileSizeLow, 1])).avg_depool(scale=[c.upRes],
                                                                                             mode=c.upsampleMode)
    # modes 1 / 3 (:805-809) slice a [tileSizeLow, tileSizeLow, C] view and resize it like mode 2
    low = tf.slice_channels(tf.reshape(in_low_, shape=[-1, c.tileSizeLow, c.tileSizeLow, c.n_inputChannels]), 0, 1)
    return GAN(tf.reshape(low, shape=[-1, c.tileSizeLow, c.tileSizeLow, 1])).avg_depool(scale=[c.upRes],
                                                                                         mode=c.upsampleMode)


def growing_disc(in_high_, in_low_, percentage, cfg, reuse=False, use_batch_norm=False, train=None, currentUpres=3):
    """multipassGAN-8x.py:783-863.  Returns (score [N,1], feature_layers)."""
    c = cfg
    with tf.variable_scope("spatial-disc", reuse=reuse):
        in_high_ = tf.reshape(in_high_, shape=[-1, c.tileSizeHigh, c.tileSizeHigh, 1])
        in_low_ = _disc_low_input(in_low_, c)
        in_high_ = tf.concat([in_low_, in_high_], axis=3)
        feature_layers = []
        gan = GAN(in_high_, bn_decay=c.bn_decay)
        x_, _ = gan.convolutional_layer(int(c.start_fms / c.upRes), [1, 1], activation_function=None, in_layer=in_high_,
                                        stride=[1], name="d_cfromDensity%d" % (c.upRes), reuse=reuse, batch_norm=False,
                                        train=train)
        feature_layers.append(lerp(None, x_, percentage - (currentUpres - 1)))
        inHigh = in_high_
        gan2 = GAN(inHigh, bn_decay=c.bn_decay)
        for j in range(currentUpres, 0, -1):
            num_fms = int(min(c.start_fms / (2 ** j), c.max_fms))
            if c.upsampling_mode == 2:
                inHigh = GAN(inHigh).avg_pool()
            x_, x1, x2 = growBlockDisc(gan, x_, int(2 ** j), int(num_fms), False, train, reuse, "d", c)
            fromDensFms = min(min(num_fms * 2, c.max_fms), c.start_fms // 2)
            _oldDens, _ = gan2.convolutional_layer(fromDensFms, [1, 1], None, stride=[1],
                                                   name="d_cfromDensity%d" % (2 ** (j - 1)), in_layer=inHigh,
                                                   reuse=reuse, batch_norm=False, train=train)
            with tf.variable_scope("blend%i" % j, reuse=reuse):
                size = c.tileSizeLow * (2 ** (j - 1)) if c.upsampling_mode == 2 else c.tileSizeHigh
                x_ = tf.reshape(lerp(_oldDens, x_, percentage - (j - 1)), shape=[-1, size, size, fromDensFms])
            feature_layers.append(lerp(None, x1, percentage - (j - 1)))
            feature_layers.append(lerp(None, x2, percentage - (j - 1)))
        if c.use_mb_stddev:
            x_ = gan.minibatch_stddev_layer(x_)
        filter = [c.filterSize, c.filterSize]
        if not c.first_nn_arch:
            x1, _ = gan.convolutional_layer(32, filter, lrelu, stride=[1], name="d_cA%d" % (1), in_layer=x_,
                                            reuse=reuse, batch_norm=use_batch_norm, train=train)
            x2, _ = gan.convolutional_layer(4, filter, None, stride=[1], name="d_cB%d" % (1), in_layer=x1,
                                            reuse=reuse, batch_norm=use_batch_norm, train=train)
        else:
            x1 = x_
        feature_layers.append(lerp(None, x1, percentage))
        # the head reads gan.layer: the pooled x2 of the last block when first_nn_arch (the blended x_ only
        # feeds the feature list there), the d_cB1 output otherwise (:860-863)
        gan.flatten()
        gan.fully_connected_layer(1, None, name="d_l6%d" % 1, gain=1)
        return gan.y(), feature_layers


def growing_disc_tempo(in_high_, percentage, cfg, n_t_channels=3, reuse=True, use_batch_norm=False, train=None,
                       currentUpres=3):
    """multipassGAN-8x.py:866-923 (useVelInTDisc 0): [N, H, W, 3] frame triples -> score"""
    c = cfg
    with tf.variable_scope("tempo-disc", reuse=reuse):
        in_high_ = tf.reshape(in_high_, shape=[-1, c.tileSizeHigh, c.tileSizeHigh, 12 if c.useVelInTDisc else 3])
        gan = GAN(in_high_, bn_decay=c.bn_decay)
        x, _ = gan.convolutional_layer(int(c.start_fms / c.upRes), [1, 1], activation_function=None, in_layer=in_high_,
                                       stride=[1], name="t_cfromDensity%d" % (c.upRes), reuse=reuse, batch_norm=False,
                                       train=train)
        inHigh = in_high_
        gan2 = GAN(inHigh, bn_decay=c.bn_decay)
        for j in range(currentUpres, 0, -1):
            num_fms = int(min(c.start_fms / (2 ** j), c.max_fms))
            if c.upsampling_mode == 2:
                inHigh = GAN(inHigh).avg_pool()
            x, x1, x2 = growBlockDisc(gan, x, int(2 ** j), int(num_fms), False, train, reuse, "t", c)
            fromDensFms = min(min(num_fms * 2, c.max_fms), c.start_fms // 2)
            _oldDens, _ = gan2.convolutional_layer(fromDensFms, [1, 1], None, stride=[1],
                                                   name="t_cfromDensity%d" % (2 ** (j - 1)), in_layer=inHigh,
                                                   reuse=reuse, batch_norm=False, train=train)
            with tf.variable_scope("blend%i" % j, reuse=reuse):
                size = c.tileSizeLow * (2 ** (j - 1)) if c.upsampling_mode == 2 else c.tileSizeHigh
                x = tf.reshape(lerp(_oldDens, x, percentage - (j - 1)), shape=[-1, size, size, fromDensFms])
        if c.use_mb_stddev:
            x = gan.minibatch_stddev_layer(x, 1)
        filter = [c.filterSize, c.filterSize]
        if not c.first_nn_arch:
            x1, _ = gan.convolutional_layer(32, filter, lrelu, stride=[1], name="t_cA%d" % (1), in_layer=x, reuse=reuse,
                                            batch_norm=use_batch_norm, train=train)
            gan.convolutional_layer(4, filter, None, stride=[1], name="t_cB%d" % (1), in_layer=x1, reuse=reuse,
                                    batch_norm=use_batch_norm, train=train)
        gan.flatten()
        gan.fully_connected_layer(1, None, name="t_l6%d" % 1, gain=1)
        return gan.y()
