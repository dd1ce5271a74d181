"""diagnostic: the gradient parity check of tests/test_train_gpu.py::test_gan4x_losses_and_gradients[4-True] repeated in
one process; prints every repetition whose generator gradients are off, with per-tensor errors and the best-fit scale"""
import math
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch
import test_train_gpu as T
TR = T.TR
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
import os
from mpgan_amd import train_ops as _to, train as _tr
if os.environ.get("ZERO_AMAX"):            # emulate a lost abs-max (scale 1) for the batch-norm backward of layers with that many channels
    _cs = [int(v) for v in os.environ["ZERO_AMAX"].split(",")]
    _bn = _to.bn_train_bwd
    def _bn_zero(dy, x, mean, var, gamma, eps=1e-3, want_amax=False):
        r = _bn(dy, x, mean, var, gamma, eps, want_amax)
        if want_amax and x.shape[-1] in _cs and x.shape[1] == int(os.environ.get("ZERO_AMAX_H", x.shape[1])):
            r[3].zero_()
        return r
    _to.bn_train_bwd = _bn_zero
TRACE = []
if os.environ.get("TRACE"):                # checksums of every data-gradient convolution's operands and result, per repetition
    from mpgan_amd import ops as _ops
    _mc = _tr._mfma_conv
    def _cs(t):
        return float(t.double().abs().sum().item())
    def _traced(x, w, wscale, prec, bias=None, act=None, leak=0.2, pad_hi=0, rescale=False, amax=None, keep=None):
        out = _mc(x, w, wscale, prec, bias, act, leak, pad_hi, rescale, amax, keep)
        if rescale:
            rec = {"shape": tuple(w.shape), "w": _cs(w), "out": _cs(out), "amax": float(amax.item()) if amax is not None else None}
            if not isinstance(x, torch.Tensor):
                rec["x_hi"] = _cs(x.buf[:, :, 0].float()); rec["x_lo"] = _cs(x.buf[:, :, 1].float())
            else:
                rec["x"] = _cs(x)
            TRACE.append(rec)
        return out
    _tr._mfma_conv = _traced
if os.environ.get("DGRAD_VALU"):           # the data gradient of layers with this many output AND input channels on the fp32 vector-ALU kernel
    _sel = int(os.environ["DGRAD_VALU"])
    _mc2 = _tr._mfma_conv
    def _route(x, w, wscale, prec, bias=None, act=None, leak=0.2, pad_hi=0, rescale=False, amax=None, keep=None):
        if rescale and isinstance(x, torch.Tensor) and w.shape[2] == _sel and w.shape[3] == _sel:
            w0 = w.permute(0, 1, 3, 2).flip(0, 1).contiguous()          # back to [kh, kw, cin, cout] of the layer
            return _to.conv2d_dgrad(x.contiguous(), w0, (x.shape[1], x.shape[2]), (1, 1), wscale)
        return _mc2(x, w, wscale, prec, bias, act, leak, pad_hi, rescale, amax, keep)
    _tr._mfma_conv = _route
if os.environ.get("REF_ACT"):              # activation backward in the tensor library
    def _act_ref(dy, y, act, leak=0.2, want_amax=False):
        if act == "relu":
            d = dy * (y > 0)
        elif act == "lrelu":
            d = dy * torch.where(y > 0, torch.ones_like(y), torch.where(y < 0, torch.full_like(y, leak), torch.full_like(y, 0.5 * (1 + leak))))
        else:
            d = dy * (1 - y * y)
        return (d, d.abs().max()) if want_amax else d
    _to.act_bwd = _act_ref
if os.environ.get("REF_BN"):               # batch-norm backward in the tensor library
    def _bn_ref(dy, x, mean, var, gamma, eps=1e-3, want_amax=False):
        c = x.shape[-1]
        n = x.numel() // c
        isd = torch.rsqrt(var + eps)
        xh = (x - mean) * isd
        dbeta = dy.reshape(-1, c).sum(0)
        dgamma = (dy * xh).reshape(-1, c).sum(0)
        dx = gamma * isd * (dy - dbeta / n - xh * dgamma / n)
        return (dx, dgamma, dbeta, dx.abs().max()) if want_amax else (dx, dgamma, dbeta)
    _to.bn_train_bwd = _bn_ref
CALLS = []
if os.environ.get("WATCH2"):               # every activation / batch-norm backward call: inputs and outputs, by call order
    _a0, _b0 = _to.act_bwd, _to.bn_train_bwd
    def _wa(dy, y, act, leak=0.2, want_amax=False):
        r = _a0(dy, y, act, leak, want_amax)
        CALLS.append(("act %s %s" % (act, tuple(dy.shape)), [dy.detach().clone(), y.detach().clone()], [(r[0] if want_amax else r).detach().clone()]))
        return r
    def _wb(dy, x, mean, var, gamma, eps=1e-3, want_amax=False):
        r = _b0(dy, x, mean, var, gamma, eps, want_amax)
        CALLS.append(("bn %s" % (tuple(dy.shape),), [t.detach().clone() for t in (dy, x, mean, var, gamma)], [t.detach().clone() for t in r[:3]]))
        return r
    _to.act_bwd, _to.bn_train_bwd = _wa, _wb
LAYER = []
if os.environ.get("WATCH"):                # checksums of what the backward of the 8 -> 128 5x5 generator layer receives
    _bw = _tr.ConvLayerFn.backward
    def _watch(ctx, dy):
        t = ctx.saved_tensors
        if tuple(t[1].shape) == (5, 5, 8, 128) and ctx.bn:
            names = ("x", "w", "lin", "mean", "var", "gamma", "y")
            rec = {"dy": dy.detach().clone()}
            for nme, v in zip(names, t):
                rec[nme] = v.detach().clone()
            outs = _bw(ctx, dy)
            for nme, v in zip(("dx", "dw", "db", "dgamma", "dbeta"), outs):
                if v is not None:
                    rec["out_" + nme] = v.detach().clone()
            LAYER.append(rec)
            return outs
        return _bw(ctx, dy)
    _tr.ConvLayerFn.backward = staticmethod(_watch)
if os.environ.get("NO_WGRAD_MM"):          # fp32 vector-ALU weight gradients everywhere, no shared G8 of d
    _to.wgrad_mfma_ok = lambda *a, **k: False
if os.environ.get("NO_SHARE"):             # matrix weight gradients through the fp32 entry (own conversions), no shared G8
    _orig = _tr._mfma_conv
    def _no_keep(*a, **k):
        k.pop("keep", None)
        return _orig(*a, **k)
    _tr._mfma_conv = _no_keep
tr, p, xs, ys = T._trainer_and_oracle(8, 4, 4, True)
Lr = TR.losses_4x(p, xs, ys, 8, 4, 4, batch_norm=True)
rd = TR.grads(Lr["disc_loss"], p, "d_")
rg = TR.grads(Lr["gen_loss_complete"], p, "g_")
bad = 0
first = None
poison = len(sys.argv) > 2
for it in range(reps):
    if poison:       # hand the allocator blocks full of NaN (or a large finite value): whoever reads memory it did not write shows
        val = float("nan") if sys.argv[2] == "nan" else float(sys.argv[2])
        junk = [torch.full((n,), val, device="cuda:0") for n in [2 ** k for k in range(6, 25)] * 2]
        del junk
    L = tr.losses(xs, ys)
    gd = torch.autograd.grad(L["disc_loss"], tr.opt_d.params, allow_unused=True, retain_graph=True)
    gg = torch.autograd.grad(L["gen_loss_complete"], tr.opt_g.params, allow_unused=True)
    calls_now = list(CALLS); del CALLS[:]
    if it == 0:
        calls0 = calls_now
        if calls0:
            print("watch2: %d activation / batch-norm backward calls per repetition" % len(calls0), flush=True)
    elif calls_now:
        def _rel(u, v):
            return float((u - v).norm() / (u.norm() + 1e-30))
        for k, (c0, c1) in enumerate(zip(calls0, calls_now)):
            ein = max(_rel(u, v) for u, v in zip(c0[1], c1[1]))
            eout = max(_rel(u, v) for u, v in zip(c0[2], c1[2]))
            if ein > 1e-5 or eout > 1e-5:
                print("rep %d: call %d (%s): inputs differ %.1e, outputs differ %.1e  <- first call that differs from repetition 0" % (it, k, c0[0], ein, eout), flush=True)
                break
    lay_now = list(LAYER); del LAYER[:]
    if it == 0:
        lay0 = lay_now
        if os.environ.get("WATCH"):
            print("watch: %d backward calls of the 8->128 layer recorded per repetition; keys %s" % (len(lay0), sorted(lay0[0]) if lay0 else None), flush=True)
    elif lay_now:
        for k, (a0, b0) in enumerate(zip(lay0, lay_now)):
            dd = {}
            for kk in a0:
                e = float((a0[kk] - b0[kk]).norm() / (a0[kk].norm() + 1e-30))
                if e > 1e-5:
                    dd[kk] = float("%.2e" % e)
            if dd:
                print("rep %d: backward %d of the 8->128 layer sees other inputs (relative L2 vs repetition 0): %s" % (it, k, dd), flush=True)
    trace_now = list(TRACE); del TRACE[:]
    if it == 0:
        trace0 = trace_now
    elif trace_now:
        def far(u, v):
            return u is not None and abs(u - v) > 3e-5 * max(abs(u), abs(v), 1e-30)
        for k, (a, b) in enumerate(zip(trace0, trace_now)):
            d = {kk: (a[kk], b[kk]) for kk in a if kk != "shape" and far(a[kk], b[kk])}
            if d:
                print("rep %d: data-gradient convolution %d %s differs beyond 3e-5: %s" % (it, k, a["shape"], d), flush=True)
    cur = {"loss_d": L["disc_loss"].detach().clone(), "loss_g": L["gen_loss_complete"].detach().clone()}
    for nme, g in list(zip(tr.opt_d.names, gd)) + list(zip(tr.opt_g.names, gg)):
        cur[nme] = g.detach().clone()
    if first is None:
        first = cur
    else:
        diffs = []
        for kk, v in cur.items():
            r0 = first[kk]
            e = float((v - r0).norm() / (r0.norm() + 1e-30))
            if e > 2e-5:
                diffs.append((kk.replace("generator/", "g:").replace("discriminator/", "d:"), float("%.1e" % e)))
        if diffs:
            print("rep %d vs rep 0: %s" % (it, diffs), flush=True)
    for grp, names, got, want in (("d", tr.opt_d.names, gd, rd), ("g", tr.opt_g.names, gg, rg)):
        num = den = dot = gg2 = 0.0
        worst = []
        for nme, g in zip(names, got):
            if nme in T.BN_BIASES:
                continue
            w = want[nme]
            gnp = g.cpu().numpy().astype(np.float64)
            num += float(((gnp - w) ** 2).sum()); den += float((w ** 2).sum())
            dot += float((gnp * w).sum()); gg2 += float((gnp ** 2).sum())
            r = T.rel(gnp, w)
            if r > 1e-4:
                worst.append((nme.replace("generator/", "").replace("discriminator/", ""), float("%.2g" % r)))
        err = math.sqrt(num / den)
        if not (err <= 5e-5):
            bad += 1
            scale = dot / den
            print("rep %d group %s: total %.2e, best-fit scale %.6f, cosine %.8f, %d tensors off: %s" % (
                it, grp, err, scale, dot / math.sqrt(den * gg2), len(worst), worst[:40]), flush=True)
print("repetitions %d, deviating %d" % (reps, bad))
