"""The reference's model functions, written against the ``GAN`` builder exactly as
the reference drivers write them against ``tools_wscale/GAN.py`` -- only the
globals they read (tileSizeLow, upRes, ...) are explicit arguments here.

  gen_resnet                GAN/multipassGAN-4x.py:505-569
  growing_gen               GAN/multipassGAN-out.py:220-338 (output mode of multipassGAN-8x.py:606-744)
  disc_binclass(_cond_tempo) GAN/multipassGAN-4x.py:572-662
"""
import math

from . import graph as tf          # the slice of the TF surface the model code uses
from .GAN import GAN, lrelu


# ----------------------------------------------------------------------------
# 4x generator
# ----------------------------------------------------------------------------
def resBlock4x(gan, inp, s1, s2, reuse, use_batch_norm, rbId, filter_size=3, train=False):
    """multipassGAN-4x.py:505-526."""
    filter = [filter_size, filter_size]
    filter1 = [1, 1]
    gc1, _ = gan.convolutional_layer(s1, filter, tf.relu, stride=[1], name="g_cA%d" % rbId, in_layer=inp,
                                     reuse=reuse, batch_norm=use_batch_norm, train=train)
    gc2, _ = gan.convolutional_layer(s2, filter, None, stride=[1], name="g_cB%d" % rbId, reuse=reuse,
                                     batch_norm=use_batch_norm, train=train)
    gs1, _ = gan.convolutional_layer(s2, filter1, None, stride=[1], name="g_s%d" % rbId, in_layer=inp, reuse=reuse,
                                     batch_norm=use_batch_norm, train=train)
    return tf.relu(tf.add(gc2, gs1))


def gen_resnet(_in, tileSizeLow, upRes, n_inputChannels, upsampling_mode=2, reuse=False, use_batch_norm=False,
               train=False):
    """multipassGAN-4x.py:528-569.  _in: flat placeholder [None, n_input]; returns [None, n_output]."""
    tileSizeHigh = tileSizeLow * upRes
    with tf.variable_scope("generator", reuse=reuse):
        if upsampling_mode == 2:
            _in = tf.reshape(_in, shape=[-1, tileSizeLow, tileSizeLow, n_inputChannels])
        elif upsampling_mode == 1 or upsampling_mode == 3:
            _in = tf.reshape(_in, shape=[-1, tileSizeHigh, tileSizeHigh, n_inputChannels])
        elif upsampling_mode == 0:
            _in = tf.reshape(_in, shape=[-1, tileSizeHigh, tileSizeLow, n_inputChannels])
        filterSize = 5
        gan = GAN(_in)
        if upsampling_mode == 2:
            inp = gan.max_depool(height_factor=upRes, width_factor=upRes)
        elif upsampling_mode == 1 or upsampling_mode == 3:
            inp = _in
        elif upsampling_mode == 0:
            inp = gan.max_depool(height_factor=1, width_factor=upRes)
        ru1 = resBlock4x(gan, inp, n_inputChannels * 2, n_inputChannels * 8, reuse, use_batch_norm, 0, filterSize, train)
        ru2 = resBlock4x(gan, ru1, 128, 128, reuse, use_batch_norm, 1, filterSize, train)
        ru3 = resBlock4x(gan, ru2, 32, 8, reuse, use_batch_norm, 2, filterSize, train)
        ru4 = resBlock4x(gan, ru3, 2, 1, reuse, False, 3, filterSize, train)
        resF = tf.reshape(ru4, shape=[-1, tileSizeHigh * tileSizeHigh])
        return resF


# ----------------------------------------------------------------------------
# 8x growing generator (output mode)
# ----------------------------------------------------------------------------
def resBlock8x(gan, inp, s1, s2, reuse, use_batch_norm, name, filter_size=3, pixel_norm=True, train=False):
    """multipassGAN-out.py:220-237."""
    filter = [filter_size, filter_size]
    filter1 = [1, 1]
    gc1, _ = gan.convolutional_layer(s1, filter, tf.relu, stride=[1], name="g_cA_" + name, in_layer=inp, reuse=reuse,
                                     batch_norm=use_batch_norm, train=train)
    if pixel_norm:
        gc1 = gan.pixel_norm(gc1)
    gc2, _ = gan.convolutional_layer(s2, filter, None, stride=[1], name="g_cB_" + name, reuse=reuse,
                                     batch_norm=use_batch_norm, train=train)
    gs1, _ = gan.convolutional_layer(s2, filter1, None, stride=[1], name="g_s_" + name, in_layer=inp, reuse=reuse,
                                     batch_norm=use_batch_norm, train=train)
    resUnit1 = tf.relu(tf.add(gc2, gs1))
    if pixel_norm:
        resUnit1 = gan.pixel_norm(resUnit1)
    return resUnit1


def growBlockGen(gan, inp, upres, fms, use_batch_norm, train, reuse, output=False, firstGen=True, filterSize=3,
                 first_nn_arch=False, use_res_net=True, pixel_norm=True, upsampleMode=1):
    """multipassGAN-out.py:239-284."""
    with tf.variable_scope("genBlock%d" % (upres), reuse=reuse):
        if firstGen:
            inDepool = gan.avg_depool(mode=upsampleMode)      # acts on gan.layer (out.py:243)
        else:
            inDepool = inp
        filter = [filterSize, filterSize]
        if first_nn_arch:
            if upres == 2:
                outp = resBlock8x(gan, inDepool, fms, fms, reuse, use_batch_norm, "first", filter[0], pixel_norm, train)
                outp = resBlock8x(gan, outp, fms, fms, reuse, use_batch_norm, "second", filter[0], pixel_norm, train)
                outp = resBlock8x(gan, outp, fms, fms, reuse, use_batch_norm, "third", filter[0], pixel_norm, train)
                outp = resBlock8x(gan, outp, fms, fms, reuse, use_batch_norm, "fourth", filter[0], pixel_norm, train)
                outp = resBlock8x(gan, outp, fms, fms, reuse, use_batch_norm, "fifth", filter[0], pixel_norm, train)
            elif upres == 4:
                outp = resBlock8x(gan, inDepool, fms * 2, fms, reuse, use_batch_norm, "first", filter[0], pixel_norm, train)
                outp = resBlock8x(gan, outp, fms, fms, reuse, use_batch_norm, "second", filter[0], pixel_norm, train)
                outp = resBlock8x(gan, outp, fms, fms, reuse, use_batch_norm, "third", filter[0], pixel_norm, train)
            if upres == 8:
                outp = resBlock8x(gan, inDepool, fms * 2, fms, reuse, use_batch_norm, "first", filter[0], pixel_norm, train)
                outp = resBlock8x(gan, outp, fms, fms, reuse, use_batch_norm, "second", filter[0], pixel_norm, train)
        else:
            if use_res_net:
                outp = resBlock8x(gan, inDepool, fms, fms, reuse, use_batch_norm, "first", filter[0], pixel_norm, train)
                outp = resBlock8x(gan, outp, fms // 2, fms // 2, reuse, use_batch_norm, "second", filter[0], pixel_norm, train)
            else:
                inp, _ = gan.convolutional_layer(fms, filter, lrelu, stride=[1], name="g_cA%d" % (upres),
                                                 in_layer=inDepool, reuse=reuse, batch_norm=use_batch_norm, train=train)
                if pixel_norm:
                    inp = gan.pixel_norm(inp)
                outp, _ = gan.convolutional_layer(fms, filter, lrelu, stride=[1], name="g_cB%d" % (upres), in_layer=inp,
                                                  reuse=reuse, batch_norm=use_batch_norm, train=train)
                if pixel_norm:
                    outp = gan.pixel_norm(outp)
        if not output:
            outpDens, _ = GAN(outp, bn_decay=0.0).convolutional_layer(1, [1, 1], None, stride=[1],
                                                                      name="g_cdensOut%d" % (upres), in_layer=outp,
                                                                      reuse=reuse, batch_norm=False, train=train, gain=1)
            return outp, outpDens
        return outp


def growing_gen(_in, tileSizeLow, upRes, n_inputChannels, reuse=False, use_batch_norm=False, train=False,
                currentUpres=3, output=True, firstGen=True, filterSize=3, startFms=256, maxFms=256,
                add_adj_idcs=False, first_nn_arch=False, use_res_net=True, pixel_norm=True, upsampleMode=1,
                addBicubicUpsample=True):
    """multipassGAN-out.py:286-338."""
    tileSizeHigh = tileSizeLow * upRes
    with tf.variable_scope("generator", reuse=reuse):
        n_channels = n_inputChannels
        if add_adj_idcs:
            n_channels += 2
        if firstGen:
            _in = tf.reshape(_in, shape=[-1, tileSizeLow, tileSizeLow, n_channels])
        else:
            _in = tf.reshape(_in, shape=[-1, tileSizeHigh, tileSizeHigh, n_channels + 1])
        gan = GAN(_in, bn_decay=0.0)
        filter = [filterSize, filterSize]
        if first_nn_arch:
            x_g = _in
        else:
            if use_res_net:
                half = min(maxFms, startFms // 2)
                x_g = resBlock8x(gan, _in, 16, half // 8, reuse, False, "1", filter[0], pixel_norm, train)
                x_g = resBlock8x(gan, x_g, half // 4, half // 2, reuse, False, "2", filter[0], pixel_norm, train)
            else:
                x_g, _ = gan.convolutional_layer(32, filter, lrelu, stride=[1], name="g_cA%d" % (1), in_layer=_in,
                                                 reuse=reuse, batch_norm=use_batch_norm, train=train)
                if pixel_norm:
                    x_g = gan.pixel_norm(x_g)
                x_g, _ = gan.convolutional_layer(min(startFms // 2, maxFms), filter, lrelu, stride=[1],
                                                 name="g_cB%d" % (1), in_layer=x_g, reuse=reuse,
                                                 batch_norm=use_batch_norm, train=train)
                if pixel_norm:
                    x_g = gan.pixel_norm(x_g)
        _dens = None
        for j in range(1, currentUpres + 1):
            num_fms = min(int(startFms / (2 ** j)), maxFms)
            if not output or j == currentUpres:
                x_g, _dens = growBlockGen(gan, x_g, int(2 ** (j)), num_fms, use_batch_norm, train, reuse, False,
                                          firstGen, filterSize, first_nn_arch, use_res_net, pixel_norm, upsampleMode)
            else:
                x_g = growBlockGen(gan, x_g, int(2 ** (j)), num_fms, use_batch_norm, train, reuse, output, firstGen,
                                   filterSize, first_nn_arch, use_res_net, pixel_norm, upsampleMode)
            if addBicubicUpsample:
                if j == currentUpres:
                    if firstGen:
                        _dens = _dens + GAN(tf.slice_channels(_in, 0, 1)).avg_depool(mode=2, scale=[int(2 ** (j))])
                    else:
                        _dens = _dens + tf.slice_channels(_in, 0, 1)
        resF = tf.reshape(_dens, shape=[-1, tileSizeHigh * tileSizeHigh])
        return resF


def second_gen_input(x, y, tileSizeLow, tileSizeHigh, n_inputChannels):
    """x_in_2 of multipassGAN-out.py:357: concat(previous pass slice, nearest-resized low-res slice)."""
    return tf.concat((tf.reshape(y, shape=[-1, tileSizeHigh, tileSizeHigh, 1]),
                      tf.resize_images(tf.reshape(x, shape=[-1, tileSizeLow, tileSizeLow, n_inputChannels]),
                                       [tileSizeHigh, tileSizeHigh], method=1)), axis=3)


# ----------------------------------------------------------------------------
# 4x discriminators (forward)
# ----------------------------------------------------------------------------
def disc_binclass(in_low, in_high, tileSizeLow, upRes, n_input, n_inputChannels, upsampling_mode=2, reuse=False,
                  use_batch_norm=False, train=False, bn_decay=0.999):
    """multipassGAN-4x.py:572-620 (2D branch).  in_low: [None, n_input] generator input; in_high: [None, H*W]."""
    tileSizeHigh = tileSizeLow * upRes
    with tf.variable_scope("discriminator", reuse=reuse):
        # tf.slice(in_low, [0,0], [N, n_input/n_inputChannels]) (:583) keeps the FIRST n_input/C entries of
        # the flat (channel-interleaved) row -- reproduced as written.
        in_low = tf.slice_flat(in_low, n_input // n_inputChannels)
        if upsampling_mode == 2:
            in_low_img = GAN(tf.reshape(in_low, shape=[-1, tileSizeLow, tileSizeLow, 1])).max_depool(
                height_factor=upRes, width_factor=upRes)
        elif upsampling_mode == 0:
            in_low_img = GAN(tf.reshape(in_low, shape=[-1, tileSizeHigh, tileSizeLow, 1])).max_depool(
                height_factor=1, width_factor=upRes)
        else:
            in_low_img = tf.reshape(in_low, shape=[-1, tileSizeHigh, tileSizeHigh, 1])
        in_high = tf.reshape(in_high, shape=[-1, tileSizeHigh, tileSizeHigh, 1])
        filter = [4, 4]
        gan = GAN(tf.concat([in_low_img, in_high], axis=-1), bn_decay=bn_decay)
        d1, _ = gan.convolutional_layer(32, filter, lrelu, stride=[2], name="d_c1", reuse=reuse)
        d2, _ = gan.convolutional_layer(64, filter, lrelu, stride=[2], name="d_c2", reuse=reuse,
                                        batch_norm=use_batch_norm, train=train)
        d3, _ = gan.convolutional_layer(128, filter, lrelu, stride=[2], name="d_c3", reuse=reuse,
                                        batch_norm=use_batch_norm, train=train)
        d4, _ = gan.convolutional_layer(256, filter, lrelu, stride=[1], name="d_c4", reuse=reuse,
                                        batch_norm=use_batch_norm, train=train)
        gan.flatten()
        gan.fully_connected_layer(1, None, name="d_l5")
        return gan.y(), d1, d2, d3, d4


def disc_binclass_cond_tempo(in_high, tileSizeLow, upRes, n_t_channels=3, reuse=False, use_batch_norm=False,
                             train=False, bn_decay=0.999):
    """multipassGAN-4x.py:622-659 (2D branch): n_t_channels advected frames packed as channels -> logit"""
    tileSizeHigh = tileSizeLow * upRes
    with tf.variable_scope("discriminatorTempo", reuse=reuse):
        in_high = tf.reshape(in_high, shape=[-1, tileSizeHigh, tileSizeHigh, n_t_channels])
        filter = [4, 4]
        gan = GAN(in_high, bn_decay=bn_decay)
        gan.convolutional_layer(32, filter, lrelu, stride=[2], name="t_c1", reuse=reuse)
        gan.convolutional_layer(64, filter, lrelu, stride=[2], name="t_c2", reuse=reuse, batch_norm=use_batch_norm,
                                train=train)
        gan.convolutional_layer(128, filter, lrelu, stride=[2], name="t_c3", reuse=reuse, batch_norm=use_batch_norm,
                                train=train)
        gan.convolutional_layer(256, filter, lrelu, stride=[1], name="t_c4", reuse=reuse, batch_norm=use_batch_norm,
                                train=train)
        gan.flatten()
        gan.fully_connected_layer(1, None, name="t_l5")
        return gan.y()


def log2_int(v):
    return int(round(math.log(v, 2)))
