"""Layer builder with the interface of the reference's ``tools_wscale/GAN.py``.

Same class name, method names, keyword arguments, return arity and the same
``self.layer`` side effects (every call reads and replaces the "current
tensor"), producing variables under the same hierarchical names
(``<scope>/<name>/{weight,bias,gamma,beta,moving_mean,moving_variance}``).
Instead of TensorFlow ops the methods record nodes of ``graph``; a
``session.Session`` then runs them as fused HIP kernels (C ABI in
include/mpgan.h).  Citations: tools_wscale/GAN.py in the reference tree.
"""
import math

import numpy as np

from . import graph as G

VERBOSE = False


def _say(msg):
    if VERBOSE:
        print(msg)


def lrelu(x, leak=0.2, name="lrelu"):
    """GAN.py:733-737."""
    return G.lrelu(x, leak, name)


lrelu.__name__ = "lrelu"


class GAN(object):
    # GAN.py:19-35
    def __init__(self, _image, bn_decay=0.999):
        self.layer = _image
        self.batch_size = _image.shape[0]
        self.DOFs = 0
        self.preFlatShapes = []
        self.weight_stack = []
        self.layer_num = 0
        self.layer_num_gen = 0
        self.layer_num_disc = 0
        self.bn_decay = bn_decay
        _say("Input: {}".format(self.layer.get_shape()))

    # GAN.py:80-119 -- conv(SAME) -> +bias -> [batch norm] -> activation; returns (activated, linear)
    def convolutional_layer(self, outChannels, _patchShape, activation_function=G.tanh, stride=[1], name="conv",
                            reuse=False, batch_norm=False, train=None, in_layer=None, in_channels=None,
                            gain=np.sqrt(2)):
        if in_layer is None:
            in_layer = self.layer
        with G.variable_scope(name, reuse=reuse):
            self.layer_num += 1
            if in_channels is not None:
                inChannels = int(in_channels)
            else:
                inChannels = int(in_layer.get_shape()[-1])
            if len(_patchShape) != 2:
                raise NotImplementedError("only 2D patches: the multi-pass path is slice-wise (GAN.py:96-99 is the 3D branch)")
            shape = [_patchShape[0], _patchShape[1], inChannels, outChannels]
            W, wscale = self.weight_variable(shape, name=name, gain=gain)
            self.layer = self.conv2d(in_layer, (W, wscale), stride)
            self.DOFs += _patchShape[0] * _patchShape[1] * inChannels * outChannels
            self.weight_stack.append(W)
            b = self.bias_variable([outChannels], name=name)
            self.layer = G.bias_add(self.layer, b)
            self.DOFs += outChannels
            if batch_norm:
                # tf.contrib.layers.batch_norm(decay, scale=True, scope=<conv scope>, is_training=train), GAN.py:110
                gamma = G.get_variable("gamma", [outChannels], "gamma")
                beta = G.get_variable("beta", [outChannels], "beta")
                mean = G.get_variable("moving_mean", [outChannels], "moving_mean")
                var = G.get_variable("moving_variance", [outChannels], "moving_variance")
                self.layer = G.batch_norm(self.layer, gamma, beta, mean, var, eps=1e-3, training=bool(train))
            layer_lin = self.layer
            act = G.activation_name(activation_function)
            if act:
                self.layer = activation_function(self.layer)
            _say("Convolutional Layer '{}' {} ({}) : {}, BN:{}".format(name, tuple(shape), act or "None",
                                                                      self.layer.get_shape(), batch_norm))
            return self.layer, layer_lin

    # GAN.py:126-147
    def residual_block(self, s1, s2, filter, activation_function=G.tanh, name="RB", reuse=False, batch_norm=False,
                       train=None, in_layer=None):
        if in_layer is None:
            in_layer = self.layer
        filter1 = [1] * len(filter)
        _say("Residual Block:")
        A, _ = self.convolutional_layer(s1, filter, activation_function, stride=[1], name=name + "_A",
                                        in_layer=in_layer, reuse=reuse, batch_norm=batch_norm, train=train)
        B, _ = self.convolutional_layer(s2, filter, None, stride=[1], name=name + "_B", reuse=reuse,
                                        batch_norm=batch_norm, train=train)
        s, _ = self.convolutional_layer(s2, filter1, None, stride=[1], name=name + "_s", in_layer=in_layer,
                                        reuse=reuse, batch_norm=batch_norm, train=train)
        self.layer = G.add(B, s)
        layer_lin = self.layer
        if activation_function:
            self.layer = activation_function(self.layer)
        return self.layer, layer_lin

    # GAN.py:152-159
    def max_pool(self, window_size=[2], window_stride=[2]):
        self.layer = G.max_pool(self.layer, window_size[0], window_stride[0])
        _say("Max Pool {}: {}".format(window_size, self.layer.get_shape()))
        return self.layer

    # GAN.py:162-169
    def avg_pool(self, window_size=[2], window_stride=[2]):
        self.layer = G.avg_pool(self.layer, window_size[0], window_stride[0])
        _say("Avg Pool {}: {}".format(window_size, self.layer.get_shape()))
        return self.layer

    # GAN.py:423-435
    def flatten(self):
        layerShape = self.layer.get_shape()
        self.preFlatShapes.append(layerShape)
        flatSize = int(layerShape[1]) * int(layerShape[2]) * int(layerShape[3])
        self.layer = G.flatten(self.layer)
        _say("Flatten: {}".format(self.layer.get_shape()))
        return flatSize

    # GAN.py:438-456
    def fully_connected_layer(self, _numHidden, _act, name="full", gain=np.sqrt(2)):
        with G.variable_scope(name):
            self.layer_num += 1
            numInput = int(self.layer.get_shape()[1])
            W, wscale = self.weight_variable([numInput, _numHidden], name=name, gain=gain)
            b = self.bias_variable([_numHidden], name=name)
            self.DOFs += numInput * _numHidden + _numHidden
            self.layer = G.bias_add(G.matmul(self.layer, W, wscale), b)
            if _act:
                self.layer = _act(self.layer)
            return self.layer

    # GAN.py:461-469
    def unflatten(self):
        unflatShape = self.preFlatShapes.pop()
        self.layer = G.reshape(self.layer, [-1] + [int(s) for s in unflatShape[1:]])
        return self.layer

    # GAN.py:472-474
    def pixel_norm(self, in_layer, epsilon=1e-8):
        self.layer = G.pixel_norm(in_layer, epsilon)
        return self.layer

    # GAN.py:476-488
    def minibatch_stddev_layer(self, x, group_size=4):
        self.layer = G.minibatch_stddev(x, group_size)
        return self.layer

    # GAN.py:501-523: kb.resize_images == nearest replication by integer factors
    def max_depool(self, in_layer=None, depth_factor=2, height_factor=2, width_factor=2):
        if in_layer is None:
            in_layer = self.layer
        # the reference resizes self.layer, not in_layer (GAN.py:517); they coincide at every call site
        s = self.layer.get_shape()
        self.layer = G.resize_images(self.layer, [int(s[1]) * height_factor, int(s[2]) * width_factor], 1)
        _say("Max Depool : {}".format(self.layer.get_shape()))
        return self.layer

    # GAN.py:528-552 (2D branch): tf.image.resize_images(method=mode), 0 bilinear / 1 nearest / 2 bicubic
    def avg_depool(self, window_size=[1, 1], window_stride=[2, 2], mode=0, scale=[2]):
        if isinstance(scale, int):
            scale = [scale]
        s = self.layer.get_shape()
        if len(scale) == 1:
            outWidth, outHeight = int(s[2]) * scale[0], int(s[1]) * scale[0]
        else:
            outWidth, outHeight = int(s[2]) * scale[1], int(s[1]) * scale[0]
        self.layer = G.resize_images(self.layer, [int(outHeight), int(outWidth)], mode)
        _say("Avg Depool {}: {}".format(window_size, self.layer.get_shape()))
        return self.layer

    # GAN.py:554-560: a [1,1] linear convolution to C * 4 channels (a fixed 4, whatever `upres` is; no batch norm:
    # variable g_cPS<stage>/weight [1,1,C,4C]), then tf.depth_to_space(upres).  As in the reference, upres != 2 only
    # works when 4C is a multiple of upres^2.
    def pixel_shuffle(self, input_layer=None, upres=2, stage="1"):
        if input_layer is None:
            input_layer = self.layer
        in_ch = int(input_layer.get_shape()[-1])
        lin, _ = self.convolutional_layer(in_ch * 4, [1, 1], None, stride=[1], name="g_cPS" + stage,
                                          in_layer=input_layer, batch_norm=False)
        self.layer = G.depth_to_space(lin, upres)
        return self.layer

    # GAN.py:566-619, deconv2d :703-708.  As written, the reference cannot execute this method: it passes `init_mean` to
    # weight_variable, which has no such parameter (GAN.py:584 vs :661) -- a TypeError before any graph is built, so
    # there is no reference behaviour to diverge from.  Here the call works: tf.nn.conv2d_transpose with
    # output_shape [N, H*stride, W*stride, outChannels], filter [kh,kw,outChannels,inChannels], SAME; `init_mean` is
    # accepted (1.0 tags the scope name like the reference does) and `strideOverride` replaces the op's strides.
    def deconvolutional_layer(self, outChannels, _patchShape, activation_function=G.tanh, stride=[1], name="deconv",
                              reuse=False, batch_norm=False, train=None, init_mean=0., strideOverride=None):
        if init_mean == 1.:
            name = name + "_EXCLUDE_ME_"
        if len(_patchShape) != 2:
            raise NotImplementedError("only 2D patches: the multi-pass path is slice-wise (GAN.py:591-598 is the 3D branch)")
        with G.variable_scope(name, reuse=reuse):
            self.layer_num += 1
            inChannels = int(self.layer.get_shape()[-1])
            st = list(stride) * 2 if len(stride) == 1 else list(stride)
            dc = st if strideOverride is None else (list(strideOverride) * 2 if len(strideOverride) == 1 else list(strideOverride))
            if dc != st:
                raise NotImplementedError("strideOverride different from stride: output_shape and strides would disagree")
            shape = [_patchShape[0], _patchShape[1], outChannels, inChannels]
            W, wscale = self.weight_variable(shape, name=name)
            self.layer = G.conv2d_transpose(self.layer, W, (st[0], st[1]), wscale)
            self.DOFs += _patchShape[0] * _patchShape[1] * outChannels * inChannels
            b = self.bias_variable([outChannels], name=name)
            self.layer = G.bias_add(self.layer, b)
            self.DOFs += outChannels
            if batch_norm:
                gamma = G.get_variable("gamma", [outChannels], "gamma")
                beta = G.get_variable("beta", [outChannels], "beta")
                mean = G.get_variable("moving_mean", [outChannels], "moving_mean")
                var = G.get_variable("moving_variance", [outChannels], "moving_variance")
                self.layer = G.batch_norm(self.layer, gamma, beta, mean, var, eps=1e-3, training=bool(train))
            layer_lin = self.layer
            if G.activation_name(activation_function):
                self.layer = activation_function(self.layer)
            return self.layer, layer_lin

    # GAN.py:347-418: semi-Lagrangian (order 1) / MacCormack (order 2) advection of `source` by the velocity tile
    def advect(self, source, vel, flags, dt, order, strength=0.0, name="Advection", startBz=15):
        return G.advect(source, vel, flags, dt, order, strength, startBz)

    # GAN.py:624-631
    def noise(self, channels=-1):
        # as many noise channels as the layer has, or `channels` of them, N(0, 0.04), appended on the channel axis
        # (the reference does not count this as a layer: layer_num stays; the generator's seed is the node's own id)
        nch = int(channels) if channels > 0 else int(self.layer.get_shape()[-1])
        noise = G.random_normal_like(self.layer, nch, 0.04, seed=None)
        self.layer = G.concat([self.layer, noise], axis=-1)
        _say("Noise {}: {}".format(noise.get_shape(), self.layer.get_shape()))
        return self.layer

    # GAN.py:635-638
    def concat(self, layer):
        self.layer = G.concat([self.layer, layer], axis=-1)
        return self.layer

    # GAN.py:641-644
    def apply(self, op):
        self.layer = op(self.layer)
        return self.layer

    # GAN.py:646-649: only identity dropout (keep_prob 1) is meaningful at inference
    def dropout(self, keep_prob):
        if keep_prob != 1.0:
            raise NotImplementedError("dropout with keep_prob != 1")
        return self.layer

    def y(self):
        return self.layer

    def getDOFs(self):
        return self.DOFs

    # GAN.py:661-678: the stored variable is N(0,1); the graph multiplies by gain/sqrt(fan_in)
    def weight_variable(self, shape, name="w", gain=np.sqrt(2), use_he=False, in_lay=None, use_wscale=True):
        if in_lay is None:
            in_lay = np.prod(shape[:-1])
        std = gain / np.sqrt(in_lay)
        v = G.get_variable("weight", shape, "weight")
        return v, float(np.float32(std))

    # GAN.py:682-683
    def bias_variable(self, shape, name="b"):
        return G.get_variable("bias", shape, "bias")

    # GAN.py:686-691
    def conv2d(self, x, W, stride=[1]):
        w, wscale = W
        if len(stride) == 1:
            strides = (stride[0], stride[0])
        else:
            strides = (stride[0], stride[1])
        return G.conv2d(x, w, strides, wscale)
