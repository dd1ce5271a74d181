"""CPU restatement of the 8x progressive-growing training graph with float64 PyTorch autograd.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``); parity unpinned (no TF 1.x, no reference
fixtures for gradients), cross-checked against the numpy generator of ``oracle/nets.py`` at
percentage = 3 and against finite differences (tests/test_oracle.py).  Follows multipassGAN-8x.py:
lerp :598, resBlock :606-623, growBlockGen :625-675, growing_gen :677-744 (training wiring with the
density heads), growBlockDisc :752-780, growing_disc :783-863, losses :1082-1142 (WGAN-GP).
upsampling_mode 2, firstNNArch 1 or 0, pixelNorm on, batch norm off — the configuration of
example_run_training.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops as O
from .train_ref import conv2d_same, lrelu

DT = torch.float64


def _ws(w, gain):
    return float(np.float32(gain / np.sqrt(np.prod(w.shape[:-1]))))


def conv(p, scope, x, act=None, gain=math.sqrt(2.0)):
    w = p[scope + "/weight"]
    y = conv2d_same(x, w * _ws(w, gain), 1) + p[scope + "/bias"].view(1, -1, 1, 1)
    if act == "relu":
        return torch.relu(y)
    if act == "lrelu":
        return lrelu(y)
    return y


def pixel_norm(x, eps=1e-8):
    return x * torch.rsqrt((x * x).mean(dim=1, keepdim=True) + eps)


def lerp(x, y, t):
    t = min(max(float(t), 0.0), 1.0)
    return x + (y - x) * t


def up2(x):
    return x.repeat_interleave(2, 2).repeat_interleave(2, 3)


def res_block(p, scope, x, name, k_unused=None, pn=True):
    a = conv(p, scope + "g_cA_" + name, x, "relu")
    if pn:
        a = pixel_norm(a)
    b = conv(p, scope + "g_cB_" + name, a)
    s = conv(p, scope + "g_s_" + name, x)
    r = torch.relu(b + s)
    return pixel_norm(r) if pn else r


def growing_gen(p, x_nhwc_np, percentage, first_nn_arch=True, current_upres=3, pn=True):
    """x: numpy [N,h,w,C] -> [N,1,H,W] float64 tensor (training wiring: density heads + fade-in)"""
    x_in = torch.tensor(np.asarray(x_nhwc_np), dtype=DT).permute(0, 3, 1, 2)
    g = "generator/"
    x_g = x_in
    if not first_nn_arch:
        x_g = res_block(p, g, x_g, "1", pn=pn)
        x_g = res_block(p, g, x_g, "2", pn=pn)
    old = conv(p, g + "g_cdensOut1", x_g, None, gain=1)
    names = ["first", "second", "third", "fourth", "fifth"]
    for j in range(1, current_upres + 1):
        up = 2 ** j
        sc = g + "genBlock%d/" % up
        h = up2(x_g)
        n_res = {2: 5, 4: 3, 8: 2}[up] if first_nn_arch else 2
        for i in range(n_res):
            h = res_block(p, sc, h, names[i], pn=pn)
        x_g = h
        dens = conv(p, sc + "g_cdensOut%d" % up, x_g, None, gain=1)
        n, hh = x_nhwc_np.shape[0], x_nhwc_np.shape[1]
        bic = O.resize_bicubic_tf1(np.asarray(x_nhwc_np[..., :1], np.float32), hh * up, hh * up)
        dens = dens + torch.tensor(bic, dtype=DT).permute(0, 3, 1, 2)
        old = lerp(up2(old), dens, percentage - (j - 1))
    return old


def grow_block_disc(p, sc, name, x, upres):
    x1 = conv(p, sc + "%s_cA%d" % (name, upres), x, "lrelu")
    x2 = conv(p, sc + "%s_cB%d" % (name, upres), x1, "lrelu")
    return F.avg_pool2d(x2, 2), x1, x2


def growing_disc(p, high_nchw, low_density_np, percentage, up_res=8, current_upres=3):
    """high: [N,1,H,W] tensor; low_density: numpy [N,h,w,1] (channel 0 of the generator input).
    first_nn_arch wiring: the head reads the pooled x2 of the last block (GAN.layer), the blended x_ only
    feeds the feature list.  Returns (score [N,1], features)."""
    d = "spatial-disc/"
    low = torch.tensor(O.resize_nearest_tf1(np.asarray(low_density_np, np.float32), low_density_np.shape[1] * up_res,
                                            low_density_np.shape[2] * up_res), dtype=DT).permute(0, 3, 1, 2)
    inp = torch.cat([low, high_nchw], dim=1)
    x = conv(p, d + "d_cfromDensity%d" % up_res, inp)
    feats = [lerp(torch.zeros_like(x), x, percentage - (current_upres - 1))]
    in_high = inp
    pooled = None
    for j in range(current_upres, 0, -1):
        in_high = F.avg_pool2d(in_high, 2)
        pooled, x1, x2 = grow_block_disc(p, d + "dBlock%d/" % (2 ** j), "d", x, 2 ** j)
        old = conv(p, d + "d_cfromDensity%d" % (2 ** (j - 1)), in_high)
        x = lerp(old, pooled, percentage - (j - 1))
        feats.append(lerp(torch.zeros_like(x1), x1, percentage - (j - 1)))
        feats.append(lerp(torch.zeros_like(x2), x2, percentage - (j - 1)))
    feats.append(lerp(torch.zeros_like(x), x, percentage))
    flat = pooled.permute(0, 2, 3, 1).reshape(pooled.shape[0], -1)
    w = p[d + "d_l61/weight"]
    score = flat @ (w * _ws(w, 1.0)) + p[d + "d_l61/bias"]
    return score, feats


def growing_disc_tempo(p, x_nchw, percentage, up_res=8, current_upres=3):
    """growing_disc_tempo (multipassGAN-8x.py:866-923), firstNNArch: [N,3,H,W] -> score"""
    t = "tempo-disc/"
    x = conv(p, t + "t_cfromDensity%d" % up_res, x_nchw)
    in_high = x_nchw
    pooled = None
    for j in range(current_upres, 0, -1):
        in_high = F.avg_pool2d(in_high, 2)
        pooled, _, _ = grow_block_disc(p, t + "tBlock%d/" % (2 ** j), "t", x, 2 ** j)
        old = conv(p, t + "t_cfromDensity%d" % (2 ** (j - 1)), in_high)
        x = lerp(old, pooled, percentage - (j - 1))
    flat = pooled.permute(0, 2, 3, 1).reshape(pooled.shape[0], -1)
    w = p[t + "t_l61/weight"]
    return flat @ (w * _ws(w, 1.0)) + p[t + "t_l61/bias"]


def tempo_losses_8x(p, batch_xts, batch_yts, batch_y_pos, tile_low, channels, percentage=3.0, lerp_factor=None,
                    wgan_lambda=10.0, wgan_target=1.0, wgan_epsilon=1e-3, weight_dld=1.0, first_nn_arch=True, adv_mode=0):
    """t_disc_loss / g_loss_t (WGAN-GP) of multipassGAN-8x.py:1216-1300 at the final stage; adv_mode 0 looks the frames
    up at the fed positions (tensorResample), 1 / 2 advects them with GAN.advect on the velocity channels of x_t
    (:1193-1199,1221-1225)"""
    from .train_ref import tensor_resample
    from . import advect as ADV
    th = tile_low * 8
    xs = np.asarray(batch_xts, np.float32).reshape(-1, tile_low, tile_low, channels)
    gen_ts = growing_gen(p, xs, percentage, first_nn_arch)                 # [3B,1,H,W]

    def pack(frames_nhwc):
        if adv_mode:
            n = frames_nhwc.shape[0]
            v = ADV.advect_torch(frames_nhwc, xs[..., 1:4], np.zeros((n, th, th, 1), np.float32), 0.5, adv_mode, 1.0, n)
        else:
            v = tensor_resample(frames_nhwc, batch_y_pos, True)
        return v.reshape(-1, 3, th * th).permute(0, 2, 1)                  # [B, n_output, 3]

    fake = pack(gen_ts.permute(0, 2, 3, 1))
    real = pack(torch.tensor(np.asarray(batch_yts), dtype=DT).reshape(-1, th, th, 1))
    to_img = lambda v: v.reshape(-1, th, th, 3).permute(0, 3, 1, 2)       # noqa: E731
    gen_s = growing_disc_tempo(p, to_img(fake), percentage)
    disc_s = growing_disc_tempo(p, to_img(real), percentage)
    L = {}
    t_disc_loss = (-disc_s).mean() * weight_dld + gen_s.mean()
    if lerp_factor is not None:
        lf = torch.tensor(np.asarray(lerp_factor), dtype=DT).reshape(-1, 1, 1)
        y_gp = (lf * real + (1 - lf) * fake.detach()).requires_grad_(True)
        t_out = growing_disc_tempo(p, to_img(y_gp), percentage)
        (g,) = torch.autograd.grad(t_out.mean(), y_gp, create_graph=True)
        norm = torch.sqrt(((g + 1e-4) ** 2).sum(dim=1))
        t_disc_loss = t_disc_loss + (disc_s ** 2).mean() * wgan_epsilon + (wgan_lambda * (norm - wgan_target) ** 2).mean()
    L["t_disc_loss"] = t_disc_loss
    L["g_loss_t"] = (-gen_s).mean()
    return L


def later_gen(p, batch_xs, batch_ys2, tile_low, channels, percentage):
    """generator of the second / third network on (x, y = (target, previous pass)) rows (multipassGAN-8x.py
    :1041-1047, growing_gen :598-700 with use_res_net and not firstNNArch) -> (gen_y, target), NCHW"""
    th = tile_low * 8
    xs = np.asarray(batch_xs, np.float32).reshape(-1, tile_low, tile_low, channels)
    y2 = torch.tensor(np.asarray(batch_ys2), dtype=DT).reshape(-1, th, th, 2)
    target = y2[..., 0:1].permute(0, 3, 1, 2)
    prev = y2[..., 1:2].permute(0, 3, 1, 2)
    x_up = torch.tensor(O.resize_nearest_tf1(xs, th, th), dtype=DT).permute(0, 3, 1, 2)
    x_in = torch.cat([prev, x_up], dim=1)
    g = "generator/"
    x_g = res_block(p, g, x_in, "1")
    x_g = res_block(p, g, x_g, "2")
    old = conv(p, g + "g_cdensOut1", x_g, None, gain=1)
    for j in range(1, 4):
        up = 2 ** j
        sc = g + "genBlock%d/" % up
        x_g = res_block(p, sc, x_g, "first")
        x_g = res_block(p, sc, x_g, "second")
        dens = conv(p, sc + "g_cdensOut%d" % up, x_g, None, gain=1) + x_in[:, 0:1]
        old = lerp(old, dens, percentage - (j - 1))
    return old, target


def later_critic(p, scope, c, inp, percentage):
    """critic of the second / third network (no pooling: every block at tileSizeHigh; growing_disc :786-866 and
    growing_disc_tempo :869-923 with upsampling_mode 1 / 3, not firstNNArch); scope "spatial-disc/" with
    c = "d", or "tempo-disc/" with c = "t" """
    x = conv(p, scope + "%s_cfromDensity8" % c, inp)
    for j in range(3, 0, -1):
        blk = scope + "%sBlock%d/" % (c, 2 ** j)
        x1 = conv(p, blk + "%s_cA%d" % (c, 2 ** j), x, "lrelu")
        x2 = conv(p, blk + "%s_cB%d" % (c, 2 ** j), x1, "lrelu")
        oldd = conv(p, scope + "%s_cfromDensity%d" % (c, 2 ** (j - 1)), inp)
        x = lerp(oldd, x2, percentage - (j - 1))
    x1 = conv(p, scope + "%s_cA1" % c, x, "lrelu")
    x2 = conv(p, scope + "%s_cB1" % c, x1)
    flat = x2.permute(0, 2, 3, 1).reshape(x2.shape[0], -1)
    w = p[scope + "%s_l61/weight" % c]
    return flat @ (w * _ws(w, 1.0)) + p[scope + "%s_l61/bias" % c]


def tempo_later_nets_losses_8x(p, batch_xts, batch_yts2, batch_y_pos, tile_low, channels, percentage=3.0,
                               lerp_factor=None, wgan_lambda=10.0, wgan_target=1.0, wgan_epsilon=1e-3, weight_dld=1.0):
    """temporal critic of the second / third network (multipassGAN-8x.py:1167-1300 with upsampling_mode 1 / 3):
    the generator runs on the three coherent frames (previous pass = channel 1 of y_t), the real triple is
    channel 0 of y_t, both advected by the tensorResample look-up at tileSizeHigh"""
    from .train_ref import tensor_resample
    th = tile_low * 8
    gen_ts, target = later_gen(p, batch_xts, batch_yts2, tile_low, channels, percentage)

    def pack(frames_nhwc):
        v = tensor_resample(frames_nhwc, batch_y_pos, True)
        return v.reshape(-1, 3, th * th).permute(0, 2, 1)                  # [B, n_output, 3]

    fake = pack(gen_ts.permute(0, 2, 3, 1))
    real = pack(target.permute(0, 2, 3, 1))
    to_img = lambda v: v.reshape(-1, th, th, 3).permute(0, 3, 1, 2)       # noqa: E731
    gen_s = later_critic(p, "tempo-disc/", "t", to_img(fake), percentage)
    disc_s = later_critic(p, "tempo-disc/", "t", to_img(real), percentage)
    t_disc_loss = (-disc_s).mean() * weight_dld + gen_s.mean()
    if lerp_factor is not None:
        lf = torch.tensor(np.asarray(lerp_factor), dtype=DT).reshape(-1, 1, 1)
        y_gp = (lf * real + (1 - lf) * fake.detach()).requires_grad_(True)
        t_out = later_critic(p, "tempo-disc/", "t", to_img(y_gp), percentage)
        (g,) = torch.autograd.grad(t_out.mean(), y_gp, create_graph=True)
        norm = torch.sqrt(((g + 1e-4) ** 2).sum(dim=1))
        t_disc_loss = t_disc_loss + (disc_s ** 2).mean() * wgan_epsilon + (wgan_lambda * (norm - wgan_target) ** 2).mean()
    return {"t_disc_loss": t_disc_loss, "g_loss_t": (-gen_s).mean()}


def later_nets_losses_8x(p, batch_xs, batch_ys2, tile_low, channels, percentage=3.0, lerp_factor=None, filter_pn=True,
                         wgan_lambda=10.0, wgan_target=1.0, wgan_epsilon=1e-3, lambda_l1=1.0):
    """second / third network (upsampling_mode 1 / 3, use_res_net, not firstNNArch; multipassGAN-8x.py
    :1041-1060): y holds (target, previous pass); no resolution change inside generator or critic"""
    th = tile_low * 8
    xs = np.asarray(batch_xs, np.float32).reshape(-1, tile_low, tile_low, channels)
    gen_y, target = later_gen(p, batch_xs, batch_ys2, tile_low, channels, percentage)

    def critic(high):
        low = torch.tensor(O.resize_nearest_tf1(xs[..., :1], th, th), dtype=DT).permute(0, 3, 1, 2)
        return later_critic(p, "spatial-disc/", "d", torch.cat([low, high], dim=1), percentage)

    disc, gen = critic(target), critic(gen_y)
    L = {"gen_y": gen_y}
    disc_loss = (-disc).mean() + gen.mean()
    if lerp_factor is not None:
        lf = torch.tensor(np.asarray(lerp_factor), dtype=DT).reshape(-1, 1, 1, 1)
        y_gp = (lf * target + (1 - lf) * gen_y.detach()).requires_grad_(True)
        (gr,) = torch.autograd.grad(critic(y_gp).mean(), y_gp, create_graph=True)
        norm = torch.sqrt(((gr.reshape(gr.shape[0], -1) + 1e-4) ** 2).sum(dim=1))
        disc_loss = disc_loss + (disc ** 2).mean() * wgan_epsilon + (wgan_lambda * (norm - wgan_target) ** 2).mean()
    L["disc_loss"] = disc_loss
    L["l1_loss"] = (target - gen_y).abs().mean()
    L["gen_loss_complete"] = (-gen).mean() + L["l1_loss"] * lambda_l1
    return L


def losses_8x(p, batch_xs, batch_ys, tile_low, channels, percentage=3.0, lerp_factor=None, lambda_l1=1.0, lambda2=0.0,
              wgan_lambda=10.0, wgan_target=1.0, wgan_epsilon=1e-3, weight_dld=1.0, first_nn_arch=True):
    """WGAN-GP losses of multipassGAN-8x.py:1082-1142 at the final growing stage's tile size"""
    up = 8
    th = tile_low * up
    xs = np.asarray(batch_xs, np.float32).reshape(-1, tile_low, tile_low, channels)
    y = torch.tensor(np.asarray(batch_ys), dtype=DT).reshape(-1, 1, th, th)
    gen_y = growing_gen(p, xs, percentage, first_nn_arch)
    low = xs[..., :1]
    disc, f_y = growing_disc(p, y, low, percentage)
    gen, f_g = growing_disc(p, gen_y, low, percentage)
    L = {"gen_y": gen_y}
    layer = 0.0
    for a, b in zip(f_y, f_g):
        layer = layer + 0.5 * ((a - b) ** 2).sum()
    L["disc_loss_layer"] = layer
    L["d_loss_y"], L["d_loss_g"] = (-disc).mean(), gen.mean()
    disc_loss = L["d_loss_y"] * weight_dld + L["d_loss_g"]
    L["l1_loss"] = (y - gen_y).abs().mean()
    L["g_loss_d"] = (-gen).mean()
    if lerp_factor is not None:
        lf = torch.tensor(np.asarray(lerp_factor), dtype=DT).reshape(-1, 1, 1, 1)
        y_gp = (lf * y + (1 - lf) * gen_y.detach()).requires_grad_(True)
        d_out, _ = growing_disc(p, y_gp, low, percentage)
        (g,) = torch.autograd.grad(d_out.mean(), y_gp, create_graph=True)
        norm = torch.sqrt(((g.reshape(g.shape[0], -1) + 1e-4) ** 2).sum(dim=1))
        L["grad_penalty_d"] = (wgan_lambda * (norm - wgan_target) ** 2).mean()
        L["epsilon_penalty_d"] = (disc ** 2).mean()
        disc_loss = disc_loss + L["epsilon_penalty_d"] * wgan_epsilon + L["grad_penalty_d"]
    L["disc_loss"] = disc_loss
    L["gen_loss_complete"] = L["g_loss_d"] + L["l1_loss"] * lambda_l1 + L["disc_loss_layer"] * lambda2
    return L


# ----------------------------------------------------------------------------------------------
# the optimiser call (multipassGAN-8x.py:490-541, 1305-1362), float64 numpy
# ----------------------------------------------------------------------------------------------
def stage_variables(names, stage, levels):
    """:1316-1321: every variable at the last stage; before that, names containing "1", "2", .., "2^(stage+1)" """
    if stage >= levels - 1:
        return list(names)
    keys = ["%i" % (2 ** i) for i in range(stage + 2)]
    return [n for n in names if any(k in n for k in keys)]


class StagedAdamRef(object):
    """per stage an Adam (own m, v, t) over its variable subset; optional dynamic loss scaling with ONE ls_var;
    optional MovingAverageOptimizer shadows per stage.  grads passed to step() are d(loss * 2^ls_var)."""

    def __init__(self, params, levels, lr=1e-4, beta1=0.0, beta2=0.99, eps=1e-8, loss_scaling=False, ema_decay=None):
        import numpy as np
        self.np = np
        self.p = {k: np.array(v, dtype=np.float64) for k, v in params.items()}
        self.names = sorted(self.p)
        self.levels, self.lr, self.b1, self.b2, self.eps = levels, lr, beta1, beta2, eps
        self.loss_scaling, self.ema_decay = loss_scaling, ema_decay
        self.m = [{k: np.zeros_like(v) for k, v in self.p.items()} for _ in range(levels)]
        self.v = [{k: np.zeros_like(v) for k, v in self.p.items()} for _ in range(levels)]
        self.t = [0] * levels
        self.ls_var = np.float32(64.0)
        self.shadow = None if ema_decay is None else [{k: v.copy() for k, v in self.p.items()} for _ in range(levels)]

    def loss_scale(self):
        np = self.np
        return float(np.exp(np.float32(self.ls_var) * np.float32(np.log(2.0)))) if self.loss_scaling else 1.0

    def step(self, grads, stage):
        """grads: name -> array (missing = zero).  Returns True if the update was applied."""
        np = self.np
        sel = stage_variables(self.names, stage, self.levels)
        coef = 1.0
        if self.loss_scaling:
            with np.errstate(all="ignore"):
                coef = np.float32(1.0 / len(sel)) * np.exp(-np.float32(self.ls_var) * np.float32(np.log(2.0)))
        scaled = {}
        ok = True
        with np.errstate(all="ignore"):
            for k in sel:
                g = np.asarray(grads.get(k, np.zeros_like(self.p[k])), dtype=np.float32) * np.float32(coef)
                ok = ok and bool(np.isfinite(g).all())
                scaled[k] = g.astype(np.float64)
        if self.loss_scaling and not ok:
            self.ls_var = np.float32(self.ls_var - 1.0)
            return False
        self.t[stage] += 1
        t = self.t[stage]
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        for k in sel:
            g = scaled[k]
            m, v = self.m[stage][k], self.v[stage][k]
            m += (g - m) * (1.0 - self.b1)
            v += (g * g - v) * (1.0 - self.b2)
            self.p[k] -= lr_t * m / (np.sqrt(v) + self.eps)
            if self.shadow is not None:
                self.shadow[stage][k] -= (1.0 - self.ema_decay) * (self.shadow[stage][k] - self.p[k])
        if self.loss_scaling:
            self.ls_var = np.float32(self.ls_var + np.float32(0.0005))
        return True
