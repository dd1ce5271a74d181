#!/usr/bin/env python3
"""Secondary benchmark: the 4x GAN training iteration (BASELINE.json configs[2], "C3"):
G + D forward/backward + both Adam updates on a batch of 16 density+velocity tiles
(tileSize 16 -> 64^2 high-res tiles, the script default multipassGAN-4x.py:40; `--tile 64` is the
256^2 variant that loads the matrix cores).  Same JSON contract as bench.py; the driver's headline
run stays bench.py.

  python bench_train.py [--workload c3|c5] [--tile 16] [--batch 16] [--steps 20] [--warmup 3] [--eager]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench_train.py --gpus N ...

`--workload c5` is one stage-3 iteration of the 8x progressive-growing training (BASELINE configs[4]):
growing_gen (firstNNArch, startFms 256, 6 input channels) + growing_disc, WGAN-GP, tileSize 16 -> 128^2,
batch 16 per GPU.  With N > 1 every rank trains on its own batch and the flat gradient bucket of each
optimiser is averaged with one RCCL all-reduce before the Adam kernel (weak scaling).
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


DTYPES = {3: "f16x3 (fp16 hi/lo split, three MFMA products, fp32 accumulate)",
          2: "forward / data gradient: f16+2xbf6 (MPG_PREC_F16F6) where the kernels cover the shape, else f16x3; weight gradient f16x3; "
             "fp32 accumulate"}


def wgrad_roofline(device, tile_high, batch, iters=10, k=5, c=128):
    """the matrix-core weight gradient of the widest conv of the step on one batch (4x: resBlock 1's 5x5
    128->128; 8x net1: a 3x3 64->64 conv of the 4x level)"""
    from mpgan_amd import train_ops
    g = torch.Generator(device=device).manual_seed(1)
    x = torch.randn((batch, tile_high, tile_high, c), device=device, generator=g).relu_()
    dy = torch.randn((batch, tile_high, tile_high, c), device=device, generator=g) * 1e-4
    # as the training step issues it (train.ConvLayerFn.backward): x is the G8 operand of the forward launch (kept), dy is
    # converted to G8 once for both gradient kernels, scaled by the max |dy| that the kernel producing dy returns; the
    # conversion is timed with the weight gradient although the data gradient shares it
    from mpgan_amd import ops
    dy_amax = ops.absmax(dy)
    xg = ops.to_g8(x)

    def call():
        return train_ops.conv2d_wgrad_g8(xg, ops.to_g8(dy, amax=dy_amax), k, k, 0.025, 3, None, dy_amax)
    for _ in range(2):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * k * k * c * c * tile_high * tile_high * batch
    ach = flops / (ms * 1e-3) / 1e12
    # HBM bytes per call from the committed PMC passes (profiles/r03/roofline_pmc_wgrad.json), which were taken
    # on exactly one shape; other shapes report null
    traffic = None
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03", "roofline_pmc_wgrad.json")
    if (k, c, batch, tile_high) == (5, 128, 16, 256) and os.path.exists(pmc):
        with open(pmc) as f:
            per = json.load(f)["per_kernel_per_call"]
        traffic = sum(v["hbm_read_bytes"] + v["hbm_write_bytes"] for kname, v in per.items() if "absmax" not in kname)
    return {"bound": "mfma", "kernel": "mpg_conv2d_wgrad_g8 %dx%d %d->%d (scaled G8 conversion of dy + wgrad_ring_kernel; x = the forward "
                                       "launch's G8 operand), %d tiles of %d^2" % (k, k, c, c, batch, tile_high),
            "achieved": round(ach, 2), "peak": DENSE_F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / DENSE_F16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "launch_ms": round(ms, 4),
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
    global torch                      # imported below, after the GPU-free launcher branch
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=("c3", "c5"))
    ap.add_argument("--tile", type=int, default=16, help="low-res tile edge (high-res = 4x / 8x)")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--channels", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prec", type=int, default=3, choices=(2, 3),
                    help="3: fp32-grade convolutions (default, what the parity tests hold the step to); 2: forward and "
                         "data-gradient convolutions at MPG_PREC_F16F6 where the kernels cover the shape (weight gradients stay at 3)")
    ap.add_argument("--eager", action="store_true", help="launch kernel by kernel instead of replaying the captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch-timeout", type=float, default=3000.0)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without torch.distributed.run: this process never touches the GPU; it starts one fresh rank per GPU
        # (mpgan_amd.launch), relays rank 0's JSON line and exits with the worst return code -- never a re-exec
        from mpgan_amd import launch
        rc, out = launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, timeout=args.launch_timeout)
        line = launch.last_json_line(out)
        if line is not None:
            print(line)
        else:
            sys.stderr.write(out)
            rc = rc or 1
        sys.stdout.flush()
        sys.exit(rc)
    import torch
    from mpgan_amd import _lib
    from mpgan_amd import dist as mdist
    from mpgan_amd.train import Trainer4x, Trainer8x
    _lib.load()
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: the training step has no CPU fallback")
    comm, dev = mdist.init_from_env()
    rank = comm.rank if comm else 0
    world = comm.world if comm else 1
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d (launch with torch.distributed.run)" % (args.gpus, world))
    rng = np.random.default_rng(rank)
    if args.workload == "c5":
        out = bench_c5(args, comm, dev, rank, world, rng)
        if rank == 0:
            print(json.dumps(out))
        return
    tr = Trainer4x(tileSizeLow=args.tile, upRes=4, n_inputChannels=args.channels, batch_norm=True, device=str(dev), comm=comm,
                   prec=args.prec)
    xs = torch.as_tensor(rng.random((args.batch, args.tile ** 2 * args.channels)).astype(np.float32), device=dev)
    ys = torch.as_tensor(rng.random((args.batch, (args.tile * 4) ** 2)).astype(np.float32), device=dev)
    step = tr.train_step if (args.eager or world > 1) else tr.train_step_graphed
    dt, (d, g) = timed(step, (xs, ys), args, comm, dev)
    th = args.tile * 4
    gf, df = fwd_flops_per_tile(th, args.channels)
    # D-step: G fwd, D fwd x2, D bwd x2 (dgrad + wgrad = 2x fwd); G-step: G fwd, D fwd x2, D(fake) dgrad, G bwd (2x)
    flops = args.batch * ((gf + 2 * df + 2 * 2 * df) + (gf + 2 * df + df + 2 * gf))
    out = {
        "metric": "training iterations/s, 4x GAN step (G+D fwd/bwd + Adam), %d tiles of %d^2 per GPU, %d channels" % (args.batch, th, args.channels),
        "value": round(1.0 / dt, 3), "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPES[args.prec], "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: 4x training step, tileSize %d -> %d^2, batch %d, density+velocity, "
                               "batchNorm on, spatial discriminator, discRuns=genRuns=1" % (args.tile, th, args.batch),
                   "launch": "eager" if (args.eager or world > 1) else "hipGraph replay",
                   "tiles_per_s": round(world * args.batch / dt, 1), "parallelism": "dp%d" % world,
                   "algorithmic_tflop_per_iteration": round(flops / 1e12, 3),
                   "algorithmic_tflops": round(flops / dt / 1e12, 1),
                   "disc_loss": float(d), "gen_loss_complete": float(g)},
        "roofline": wgrad_roofline(dev, th, args.batch),
    }
    if rank != 0:
        return
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.tile, args.batch, args.channels, 5)
    print(json.dumps(out))


def timed(step, batch, args, comm, dev):
    """W warm-up steps, then K steps bracketed by barrier + synchronize; the maximum over ranks"""
    for _ in range(max(args.warmup, 1)):
        res = step(*batch)
    torch.cuda.synchronize(dev)
    if comm:
        comm.barrier()
    t0 = time.time()
    for _ in range(args.steps):
        res = step(*batch)
    torch.cuda.synchronize(dev)
    if comm:
        comm.barrier()
    dt = (time.time() - t0) / args.steps
    if comm:
        dt = comm.max_float(dt, dev)
    return dt, res


def bench_c5(args, comm, dev, rank, world, rng):
    from mpgan_amd.arch import Cfg8x
    from mpgan_amd.train import Trainer8x
    cfg = Cfg8x(tileSizeLow=args.tile, upRes=8, n_inputChannels=6, start_fms=256, max_fms=256)
    tr = Trainer8x(cfg, device=str(dev), comm=comm, prec=args.prec)
    xs = torch.as_tensor(rng.random((args.batch, cfg.n_input)).astype(np.float32), device=dev)
    ys = torch.as_tensor(rng.random((args.batch, cfg.n_output)).astype(np.float32), device=dev)
    dt, (d, g) = timed(lambda a, b: tr.train_step(a, b, 3.0), (xs, ys), args, comm, dev)
    th = cfg.tileSizeHigh
    return {
        "metric": "training iterations/s, 8x progressive-growing stage 3 (growing_gen + growing_disc, WGAN-GP, Adam), "
                  "%d tiles of %d^2 per GPU" % (args.batch, th),
        "value": round(1.0 / dt, 3), "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPES[args.prec], "data": "synthetic",
        "config": {"workload": "BASELINE configs[4] per GPU: multipassGAN-8x.py final stage (percentage 3.0), tileSize %d -> %d^2, "
                               "batch %d, firstNNArch, startFms 256, 6 input channels, WGAN-GP" % (args.tile, th, args.batch),
                   "parallelism": "dp%d" % world, "tiles_per_s": round(world * args.batch / dt, 1),
                   "disc_loss": float(d), "gen_loss_complete": float(g)},
        "roofline": wgrad_roofline(dev, th // 2, args.batch, k=3, c=64) if rank == 0 else None,
    }


if __name__ == "__main__":
    main()
