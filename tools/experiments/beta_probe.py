import sys, math
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_train_gpu as T
from mpgan_amd import train_ops
TR = T.TR
for trial in range(3):
    tr, p, xs, ys = T._trainer_and_oracle(8, 4, 4, True)
    L = tr.losses(xs, ys)
    Lr = TR.losses_4x(p, xs, ys, 8, 4, 4, batch_norm=True)
    gg = torch.autograd.grad(L["gen_loss_complete"], tr.opt_g.params, allow_unused=True)
    rg = TR.grads(Lr["gen_loss_complete"], p, "g_")
    out = []
    for nme, g in zip(tr.opt_g.names, gg):
        w = rg[nme]
        gnp = g.cpu().numpy().astype(np.float64)
        r = T.rel(gnp, w)
        if r > 2e-4 and nme not in T.BN_BIASES:
            out.append("%s %.2e |w| %.2e" % (nme.split("/")[-2] + "/" + nme.split("/")[-1], r, float(np.sqrt((w ** 2).sum()))))
    print("trial", trial, "; ".join(out))
