#!/usr/bin/env python3
"""Secondary benchmark: the 4x GAN training iteration (BASELINE.json configs[2], "C3"):
G + D forward/backward + both Adam updates on a batch of 16 density+velocity tiles
(tileSize 16 -> 64^2 high-res tiles, the script default multipassGAN-4x.py:40; `--tile 64` is the
256^2 variant that loads the matrix cores).  Same JSON contract as bench.py; the driver's headline
run stays bench.py.

  python bench_train.py [--tile 16] [--batch 16] [--steps 20] [--warmup 3] [--eager]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

DENSE_F16_MFMA_PEAK_TFLOPS = 2500.0


def fwd_flops_per_tile(tile_high, c):
    hw = tile_high * tile_high
    g = 0
    for cin, s1, s2 in [(c, 2 * c, 8 * c), (8 * c, 128, 128), (128, 32, 8), (8, 2, 1)]:
        g += 2 * 25 * cin * s1 * hw + 2 * 25 * s1 * s2 * hw + 2 * cin * s2 * hw
    d, h = 0, tile_high
    for cin, cout, s in [(2, 32, 2), (32, 64, 2), (64, 128, 2), (128, 256, 1)]:
        h = h // s
        d += 2 * 16 * cin * cout * h * h
    d += 2 * h * h * 256
    return g, d


def wgrad_roofline(device, tile_high, batch, iters=10):
    """the matrix-core weight gradient of resBlock 1's 5x5 128->128 conv on one batch"""
    from mpgan_amd import train_ops
    g = torch.Generator(device=device).manual_seed(1)
    x = torch.randn((batch, tile_high, tile_high, 128), device=device, generator=g).relu_()
    dy = torch.randn((batch, tile_high, tile_high, 128), device=device, generator=g) * 1e-4
    for _ in range(2):
        train_ops.conv2d_wgrad_mfma(x, dy, 5, 5, 0.025, 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    e0.record()
    for _ in range(iters):
        train_ops.conv2d_wgrad_mfma(x, dy, 5, 5, 0.025, 3)
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * 25 * 128 * 128 * tile_high * tile_high * batch
    ach = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "mpg_conv2d_wgrad_mfma 5x5 128->128 (absmax + P16 rewrite + wgrad_mfma_kernel<5,2,3> x2), "
                                       "%d tiles of %d^2" % (batch, tile_high),
            "achieved": round(ach, 2), "peak": DENSE_F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / DENSE_F16_MFMA_PEAK_TFLOPS, 4), "traffic": None, "launch_ms": round(ms, 4),
            "algorithmic_gflop_per_launch": round(flops / 1e9, 2), "mfma_products_per_mac": "3 fp16"}


def cpu_baseline(tile, batch, c, seed):
    """one iteration of the float64 autograd restatement (oracle/train_ref.py) on the host cores, on a
    bounded sample: min(batch, 4) tiles, scaled linearly to the batch"""
    from oracle import train_ref as TR
    from oracle.nets import ParamSource
    from mpgan_amd.train import Trainer4x
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, 16)))
    tr = Trainer4x(tileSizeLow=tile, upRes=4, n_inputChannels=c, batch_norm=True, device="cpu")
    ps = ParamSource(seed=seed)
    p = TR.to_params({n: ps.get(n, s.shape, s.kind) for n, s in tr.graph.variables.items()})
    nb = min(batch, 4)
    rng = np.random.default_rng(0)
    xs = rng.random((nb, tile * tile * c)).astype(np.float32)
    ys = rng.random((nb, (tile * 4) ** 2)).astype(np.float32)
    t0 = time.time()
    L = TR.losses_4x(p, xs, ys, tile, 4, c)
    TR.grads(L["disc_loss"], p, "d_")
    L = TR.losses_4x(p, xs, ys, tile, 4, c)
    TR.grads(L["gen_loss_complete"], p, "g_")
    dt = (time.time() - t0) * batch / nb
    return {"value": round(1.0 / dt, 5), "unit": "iterations/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "D-step + G-step forward/backward of oracle/train_ref.py (PyTorch-CPU float64 autograd; TF 1.x "
                      "unavailable) on %d of %d tiles, scaled linearly; no Adam" % (nb, batch)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--tile", type=int, default=16, help="low-res tile edge (high-res = 4x)")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--channels", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--eager", action="store_true", help="launch kernel by kernel instead of replaying the captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.gpus != 1:
        raise SystemExit("bench_train.py measures one GPU (data-parallel training is a later row)")
    from mpgan_amd import _lib
    from mpgan_amd.train import Trainer4x
    _lib.load()
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: the training step has no CPU fallback")
    dev = torch.device("cuda:0")
    tr = Trainer4x(tileSizeLow=args.tile, upRes=4, n_inputChannels=args.channels, batch_norm=True, device="cuda:0")
    rng = np.random.default_rng(0)
    xs = torch.as_tensor(rng.random((args.batch, args.tile ** 2 * args.channels)).astype(np.float32), device=dev)
    ys = torch.as_tensor(rng.random((args.batch, (args.tile * 4) ** 2)).astype(np.float32), device=dev)
    step = tr.train_step if args.eager else tr.train_step_graphed
    for _ in range(max(args.warmup, 1)):
        step(xs, ys)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        d, g = step(xs, ys)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / args.steps
    th = args.tile * 4
    gf, df = fwd_flops_per_tile(th, args.channels)
    # D-step: G fwd, D fwd x2, D bwd x2 (dgrad + wgrad = 2x fwd); G-step: G fwd, D fwd x2, D(fake) dgrad, G bwd (2x)
    flops = args.batch * ((gf + 2 * df + 2 * 2 * df) + (gf + 2 * df + df + 2 * gf))
    out = {
        "metric": "training iterations/s, 4x GAN step (G+D fwd/bwd + Adam), %d tiles of %d^2, %d channels" % (args.batch, th, args.channels),
        "value": round(1.0 / dt, 3), "unit": "iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16x3 (fp16 hi/lo split, three MFMA products, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: 4x training step, tileSize %d -> %d^2, batch %d, density+velocity, "
                               "batchNorm on, spatial discriminator, discRuns=genRuns=1" % (args.tile, th, args.batch),
                   "launch": "eager" if args.eager else "hipGraph replay", "tiles_per_s": round(args.batch / dt, 1),
                   "algorithmic_tflop_per_iteration": round(flops / 1e12, 3),
                   "algorithmic_tflops": round(flops / dt / 1e12, 1),
                   "disc_loss": float(d), "gen_loss_complete": float(g)},
        "roofline": wgrad_roofline(dev, th, args.batch),
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.tile, args.batch, args.channels, 5)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
