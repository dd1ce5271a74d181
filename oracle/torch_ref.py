"""Second, independent CPU implementation (PyTorch-CPU fp32, oneDNN convs).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Two uses:
* cross-check of the numpy restatement in ``oracle/ops.py`` (tests), and
* the timed "CPU restatement (PyTorch-CPU); TF 1.x unavailable" baseline of
  ``bench.py``.
It is not the reference; TF 1.x cannot be installed in this image.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .ops import same_pad


def conv2d_same(x_nchw, w_hwio, stride=1):
    """tf.nn.conv2d SAME (GAN.py:686-691) on an NCHW torch tensor."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    _, pt, pb = same_pad(x_nchw.shape[2], kh, stride)
    _, pl, pr = same_pad(x_nchw.shape[3], kw, stride)
    w = torch.as_tensor(np.ascontiguousarray(np.transpose(w_hwio, (3, 2, 0, 1))))
    return F.conv2d(F.pad(x_nchw, (pl, pr, pt, pb)), w, stride=stride)


def conv_layer(params, scope, x, cout, k, act=None, stride=1, batch_norm=False, gain=math.sqrt(2.0)):
    """GAN.convolutional_layer (GAN.py:80-119), inference."""
    w = params[scope + "/weight"]
    b = params[scope + "/bias"]
    ws = np.float32(gain / np.sqrt(np.prod(w.shape[:-1])))
    y = conv2d_same(x, w * ws, stride) + torch.as_tensor(b).view(1, -1, 1, 1)
    if batch_norm:
        g = torch.as_tensor(params[scope + "/gamma"]).view(1, -1, 1, 1)
        be = torch.as_tensor(params[scope + "/beta"]).view(1, -1, 1, 1)
        mu = torch.as_tensor(params[scope + "/moving_mean"]).view(1, -1, 1, 1)
        var = torch.as_tensor(params[scope + "/moving_variance"]).view(1, -1, 1, 1)
        y = (y - mu) * (g / torch.sqrt(var + 1e-3)) + be
    if act == "relu":
        return torch.relu(y), y
    if act == "lrelu":
        return 0.6 * y + 0.4 * y.abs(), y
    return y, y


def gen_resnet(params, x_nhwc, up_res=4, upsampling_mode=2, batch_norm=True):
    """gen_resnet (multipassGAN-4x.py:528-569) with PyTorch-CPU; x: numpy NHWC."""
    with torch.no_grad():
        x = torch.as_tensor(np.ascontiguousarray(np.transpose(x_nhwc, (0, 3, 1, 2))))
        c = x.shape[1]
        if upsampling_mode == 2:
            x = x.repeat_interleave(up_res, 2).repeat_interleave(up_res, 3)
        elif upsampling_mode == 0:
            x = x.repeat_interleave(up_res, 3)
        widths = [(c * 2, c * 8, batch_norm), (128, 128, batch_norm), (32, 8, batch_norm), (2, 1, False)]
        for i, (s1, s2, bn) in enumerate(widths):
            a, _ = conv_layer(params, "generator/g_cA%d" % i, x, s1, 5, "relu", 1, bn)
            _, b = conv_layer(params, "generator/g_cB%d" % i, a, s2, 5, None, 1, bn)
            _, s = conv_layer(params, "generator/g_s%d" % i, x, s2, 1, None, 1, bn)
            x = torch.relu(b + s)
        return x.permute(0, 2, 3, 1).contiguous().numpy()


class fast_convs(object):
    """Context manager: ``oracle.ops.conv2d_same`` (float64 numpy, tap by tap) is replaced by the
    PyTorch-CPU fp32 convolution above while the block runs, so that the numpy restatements of the
    8x generators (``oracle.nets.growing_gen``) finish in seconds at 512^2 -- the full-size parity tests
    need that.  Everything else (resize, pixel_norm, residual sums) stays the numpy code.  Cross-checked
    against the float64 convolution in tests/test_oracle.py."""

    def __enter__(self):
        from . import ops
        self._ops, self._saved = ops, ops.conv2d_same

        def conv(x, w, stride=(1, 1)):
            assert stride[0] == stride[1]
            with torch.no_grad():
                xt = torch.as_tensor(np.ascontiguousarray(np.transpose(np.asarray(x, dtype=np.float32), (0, 3, 1, 2))))
                y = conv2d_same(xt, np.asarray(w, dtype=np.float32), stride[0])
                return np.ascontiguousarray(y.permute(0, 2, 3, 1).numpy())

        ops.conv2d_same = conv
        return self

    def __exit__(self, *exc):
        self._ops.conv2d_same = self._saved
        return False
