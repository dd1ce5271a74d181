"""CPU restatement of the 4x GAN training graph with float64 PyTorch autograd.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``); parity unpinned (TF 1.x is not installable
here and the reference holds no gradient fixtures), cross-checked against finite differences in
tests/test_oracle.py.  Follows multipassGAN-4x.py: gen_resnet :528-569, disc_binclass :572-620,
losses :744-768, GAN.convolutional_layer (GAN.py:80-119) with tf.contrib batch_norm in training
mode (batch moments, biased variance, eps 1e-3), lrelu (GAN.py:733-737), Adam as
tf.train.AdamOptimizer.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .ops import same_pad

DT = torch.float64


def to_params(params_np):
    """numpy dict -> float64 leaf tensors (trainable: weight, bias, gamma, beta)"""
    out = {}
    for k, v in params_np.items():
        t = torch.tensor(np.asarray(v), dtype=DT)
        t.requires_grad_(k.rsplit("/", 1)[-1] in ("weight", "bias", "gamma", "beta"))
        out[k] = t
    return out


def conv2d_same(x, w_hwio, stride=1):
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    _, pt, pb = same_pad(x.shape[2], kh, stride)
    _, pl, pr = same_pad(x.shape[3], kw, stride)
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w_hwio.permute(3, 2, 0, 1).contiguous(), stride=stride)


def lrelu(x, leak=0.2):
    return 0.5 * (1 + leak) * x + 0.5 * (1 - leak) * x.abs()


def conv_layer(p, scope, x, act=None, stride=1, batch_norm=False, gain=math.sqrt(2.0), stats=None):
    """x NCHW float64 -> (activated, linear); batch statistics when batch_norm"""
    w = p[scope + "/weight"]
    ws = float(np.float32(gain / np.sqrt(np.prod(w.shape[:-1]))))
    y = conv2d_same(x, w * ws, stride) + p[scope + "/bias"].view(1, -1, 1, 1)
    if batch_norm:
        mean = y.mean(dim=(0, 2, 3), keepdim=True)
        var = ((y - mean) ** 2).mean(dim=(0, 2, 3), keepdim=True)
        if stats is not None:
            stats[scope] = (mean.detach().flatten(), var.detach().flatten())
        y = (y - mean) / torch.sqrt(var + 1e-3) * p[scope + "/gamma"].view(1, -1, 1, 1) + p[scope + "/beta"].view(1, -1, 1, 1)
    if act == "relu":
        return torch.relu(y), y
    if act == "lrelu":
        return lrelu(y), y
    return y, y


def gen_resnet(p, x_nchw, up_res=4, upsampling_mode=2, batch_norm=True, stats=None):
    x = x_nchw
    c = x.shape[1]
    if upsampling_mode == 2:
        x = x.repeat_interleave(up_res, 2).repeat_interleave(up_res, 3)
    widths = [(c * 2, c * 8, batch_norm), (128, 128, batch_norm), (32, 8, batch_norm), (2, 1, False)]
    for i, (_, _, bn) in enumerate(widths):
        a, _ = conv_layer(p, "generator/g_cA%d" % i, x, "relu", 1, bn, stats=stats)
        _, b = conv_layer(p, "generator/g_cB%d" % i, a, None, 1, bn, stats=stats)
        _, s = conv_layer(p, "generator/g_s%d" % i, x, None, 1, bn, stats=stats)
        x = torch.relu(b + s)
    return x            # [N,1,H,W]


def disc_binclass(p, low_density_nchw, high_nchw, up_res=4, batch_norm=True, stats=None):
    low = low_density_nchw.repeat_interleave(up_res, 2).repeat_interleave(up_res, 3)
    x = torch.cat([low, high_nchw], dim=1)
    sc = "discriminator/"
    d1, _ = conv_layer(p, sc + "d_c1", x, "lrelu", 2, False, stats=stats)
    d2, _ = conv_layer(p, sc + "d_c2", d1, "lrelu", 2, batch_norm, stats=stats)
    d3, _ = conv_layer(p, sc + "d_c3", d2, "lrelu", 2, batch_norm, stats=stats)
    d4, _ = conv_layer(p, sc + "d_c4", d3, "lrelu", 1, batch_norm, stats=stats)
    flat = d4.permute(0, 2, 3, 1).reshape(d4.shape[0], -1)       # NHWC flatten (GAN.py:423-435)
    w = p[sc + "d_l5/weight"]
    ws = float(np.float32(math.sqrt(2.0) / np.sqrt(w.shape[0])))
    logit = flat @ (w * ws) + p[sc + "d_l5/bias"]
    return logit, d1, d2, d3, d4


def sigmoid_ce(logits, label):
    return (torch.clamp(logits, min=0) - logits * label + torch.log1p(torch.exp(-logits.abs()))).mean()


def losses_4x(p, batch_xs, batch_ys, tile_low, up_res, channels, batch_norm=True, lambda_l1=1.0, lambda2=0.0,
              lambda2_l=(1.0, 1.0, 1.0, 1.0), weight_dld=1.0, stats=None):
    """batch_xs [B, tile_low^2 * C], batch_ys [B, (tile_low*up)^2] (numpy) -> dict of loss tensors"""
    th = tile_low * up_res
    x_nhwc = torch.tensor(np.asarray(batch_xs), dtype=DT).reshape(-1, tile_low, tile_low, channels)
    x = x_nhwc.permute(0, 3, 1, 2)
    y = torch.tensor(np.asarray(batch_ys), dtype=DT).reshape(-1, 1, th, th)
    gen_part = gen_resnet(p, x, up_res, 2, batch_norm, stats)
    # tf.slice(in_low, [0,0], [N, n_input/C]) keeps the first n_input/C entries of the interleaved row (:583)
    flat = x_nhwc.reshape(x_nhwc.shape[0], -1)[:, :tile_low * tile_low]
    low = flat.reshape(-1, 1, tile_low, tile_low)
    s_real, s_fake = ({}, {}) if stats is not None else (None, None)
    disc, dy1, dy2, dy3, dy4 = disc_binclass(p, low, y, up_res, batch_norm, s_real)
    gen, gy1, gy2, gy3, gy4 = disc_binclass(p, low, gen_part, up_res, batch_norm, s_fake)
    if stats is not None:
        stats["disc_real"], stats["disc_fake"] = s_real, s_fake
    L = {}
    L["disc_loss_disc"] = sigmoid_ce(disc, torch.ones_like(disc))
    L["disc_loss_gen"] = sigmoid_ce(gen, torch.zeros_like(gen))
    layer = 0.0
    for kf, a, b in zip(lambda2_l, (dy1, dy2, dy3, dy4), (gy1, gy2, gy3, gy4)):
        layer = layer + kf * 0.5 * ((a - b) ** 2).sum()
    L["disc_loss_layer"] = layer
    L["disc_loss"] = L["disc_loss_disc"] * weight_dld + L["disc_loss_gen"]
    L["gen_loss"] = sigmoid_ce(gen, torch.ones_like(gen))
    L["gen_l2_loss"] = 0.5 * ((y - gen_part) ** 2).sum()
    L["gen_l1_loss"] = (y - gen_part).abs().mean()
    L["gen_loss_complete"] = L["gen_loss"] + L["gen_l1_loss"] * lambda_l1 + L["disc_loss_layer"] * lambda2
    L["gen_part"] = gen_part
    return L


def tensor_resample(value_nhwc, pos, clamp=True):
    """tensorResample (multipassGAN-4x.py:398-441), 2D: value [N,H,W,C] tensor, pos [N,H,W,2] = (y, x)"""
    n, h, w, c = value_nhwc.shape
    pos = torch.as_tensor(np.asarray(pos), dtype=DT).reshape(n, h, w, 2)
    f = pos - 0.5
    fl = torch.floor(f).long()
    out = 0.0
    bidx = torch.arange(n).view(n, 1, 1).expand(n, h, w)
    lim = torch.tensor([h - 1, w - 1])
    for ay in (0, 1):
        for ax in (0, 1):
            idx = fl + torch.tensor([ay, ax])
            inside = ((idx >= 0) & (idx <= lim)).all(dim=-1)
            if clamp:
                idx = torch.minimum(torch.clamp(idx, min=0), lim)
                inside = torch.ones_like(inside)
            wgt = (1.0 - (f - idx.to(DT)).abs()).prod(dim=-1, keepdim=True)
            safe = torch.minimum(torch.clamp(idx, min=0), lim)
            g = value_nhwc[bidx, safe[..., 0], safe[..., 1]]
            out = out + g * wgt * inside.unsqueeze(-1).to(DT)
    return out


def disc_tempo(p, x_nchw, batch_norm=True):
    """disc_binclass_cond_tempo (multipassGAN-4x.py:622-659): [N,3,H,W] -> logit"""
    sc = "discriminatorTempo/"
    t1, _ = conv_layer(p, sc + "t_c1", x_nchw, "lrelu", 2, False)
    t2, _ = conv_layer(p, sc + "t_c2", t1, "lrelu", 2, batch_norm)
    t3, _ = conv_layer(p, sc + "t_c3", t2, "lrelu", 2, batch_norm)
    t4, _ = conv_layer(p, sc + "t_c4", t3, "lrelu", 1, batch_norm)
    flat = t4.permute(0, 2, 3, 1).reshape(t4.shape[0], -1)
    w = p[sc + "t_l5/weight"]
    ws = float(np.float32(math.sqrt(2.0) / np.sqrt(w.shape[0])))
    return flat @ (w * ws) + p[sc + "t_l5/bias"]


def tempo_losses_4x(p, batch_xts, batch_yts, batch_y_pos, tile_low, up_res, channels, batch_norm=True, weight_dld=1.0,
                    adv=True, clamp=True):
    """t_disc_loss / t_gen_loss of multipassGAN-4x.py:790-866 for [3B, .] coherent frame rows"""
    th = tile_low * up_res
    x = torch.tensor(np.asarray(batch_xts), dtype=DT).reshape(-1, tile_low, tile_low, channels).permute(0, 3, 1, 2)
    gen_t = gen_resnet(p, x, up_res, 2, batch_norm)                       # [3B,1,H,W]

    def pack(frames_nhwc):
        v = tensor_resample(frames_nhwc, batch_y_pos, clamp) if adv else frames_nhwc
        v = v.reshape(-1, 3, th * th).permute(0, 2, 1)                    # batch, n_output, frames
        return v.reshape(-1, th, th, 3).permute(0, 3, 1, 2)

    fake = pack(gen_t.permute(0, 2, 3, 1))
    real = pack(torch.tensor(np.asarray(batch_yts), dtype=DT).reshape(-1, th, th, 1))
    g_t, d_t = disc_tempo(p, fake, batch_norm), disc_tempo(p, real, batch_norm)
    L = {}
    L["t_disc_loss"] = sigmoid_ce(d_t, torch.ones_like(d_t)) * weight_dld + sigmoid_ce(g_t, torch.zeros_like(g_t))
    L["t_gen_loss"] = sigmoid_ce(g_t, torch.ones_like(g_t))
    # useTempoL2 (multipassGAN-4x.py:815-826): sum over consecutive frame pairs of mean((frame_i - frame_i+1)^2)
    fr = fake.reshape(-1, 3, th * th)
    L["tl_gen_loss"] = sum(torch.mean((fr[:, i] - fr[:, i + 1]) ** 2) for i in range(2))
    return L


def grads(loss, p, tag):
    names = sorted(n for n, t in p.items() if t.requires_grad and tag in n)
    gs = torch.autograd.grad(loss, [p[n] for n in names], allow_unused=True, retain_graph=True)
    return {n: (g.numpy() if g is not None else np.zeros(tuple(p[n].shape))) for n, g in zip(names, gs)}


def adam_tf(param, grad, m, v, t, lr=2e-4, b1=0.5, b2=0.999, eps=1e-8):
    """one tf.train.AdamOptimizer update (numpy float64); returns (param, m, v)"""
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    # the ApplyAdam functor forms T(1) - beta in the tensor type (float32)
    omb1 = float(np.float32(1) - np.float32(b1))
    omb2 = float(np.float32(1) - np.float32(b2))
    m = m + (grad - m) * omb1
    v = v + (grad * grad - v) * omb2
    return param - lr_t * m / (np.sqrt(v) + eps), m, v
