"""diagnostic: every torch.empty / empty_like of the training step returns NaN-filled memory (call sites listed in
MPG_CLEAN come back zeroed instead): finds the buffer that is read before it is written"""
import math
import os
import sys
import traceback
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch
import test_train_gpu as T
TR = T.TR
clean = set(filter(None, os.environ.get("MPG_CLEAN", "").split(",")))
sites = {}
_empty, _empty_like = torch.empty, torch.empty_like


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "mpgan_amd" in fr.filename or "multi-pass-gan_amd" in fr.filename:
            return "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
    return "other"


def fill(t):
    s = site()
    sites[s] = sites.get(s, 0) + 1
    if t.is_cuda and t.numel():
        if s in clean:
            t.zero_()
        elif t.dtype in (torch.float32, torch.float16):
            t.fill_(float("nan"))
        elif t.dtype == torch.uint8:
            t.fill_(0xFF)
    return t


torch.empty = lambda *a, **k: fill(_empty(*a, **k))
torch.empty_like = lambda *a, **k: fill(_empty_like(*a, **k))
tr, p, xs, ys = T._trainer_and_oracle(8, 4, 4, True)
Lr = TR.losses_4x(p, xs, ys, 8, 4, 4, batch_norm=True)
rg = TR.grads(Lr["gen_loss_complete"], p, "g_")
for it in range(2):
    L = tr.losses(xs, ys)
    gd = torch.autograd.grad(L["disc_loss"], tr.opt_d.params, allow_unused=True, retain_graph=True)
    gg = torch.autograd.grad(L["gen_loss_complete"], tr.opt_g.params, allow_unused=True)
    num = den = 0.0
    off = []
    for nme, g in zip(tr.opt_g.names, gg):
        if nme in T.BN_BIASES:
            continue
        w = rg[nme]
        gnp = g.cpu().numpy().astype(np.float64)
        num += float(((gnp - w) ** 2).sum()); den += float((w ** 2).sum())
        r = T.rel(gnp, w)
        if not (r <= 1e-4):
            off.append(nme.replace("generator/", ""))
    print("rep %d: generator gradient error %.3e; tensors off: %s" % (it, math.sqrt(num / den), off), flush=True)
print("allocation sites:", sorted(sites.items()))
