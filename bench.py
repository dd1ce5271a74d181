#!/usr/bin/env python3
"""Headline benchmark: 4x two-pass generator inference, 64^3 -> 256^3 density-only
(BASELINE.json configs[1], "C2"): a batch of 8 volumes per GPU, device-resident in,
device-resident out.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = both passes of `volumes_per_gpu * N` synthetic volumes.  With N > 1 every
volume's slices are sharded over the ranks along the pass's slice axis and an RCCL
all-gather reassembles the volume between the passes (weak scaling: 8 volumes per GPU).
Prints ONE JSON line on rank 0.
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

SIM, UP = 64, 4
S = SIM * UP
SLICES_PER_VOLUME = 2 * S
# algorithmic work: sum over conv layers of 2*kh*kw*Cin*Cout*H*W at 256^2, C = 1
GFLOP_PER_SLICE = 71.692
DENSE_F16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense


def gen_resnet_flops_per_slice(c=1, hw=256 * 256):
    f = 0
    widths = [(c, 2 * c, 8 * c), (8 * c, 128, 128), (128, 32, 8), (8, 2, 1)]
    for cin, s1, s2 in widths:
        f += 2 * 25 * cin * s1 * hw + 2 * 25 * s1 * s2 * hw + 2 * cin * s2 * hw
    return f


def dominant_kernel_roofline(mpg, device, prec, iters=20):
    """conv_mfma_kernel of resBlock 1's B conv (5x5 128->128 plus its 1x1 8->128 shortcut as a
    second K-segment, multipassGAN-4x.py:561) on one batch of 8 slices, timed with events on
    the launch stream."""
    from mpgan_amd import ops
    n, h, w = 8, S, S
    g = torch.Generator(device=device).manual_seed(1)
    a = torch.randn((n, h, w, 128), device=device, generator=g).relu_()
    x = torch.randn((n, h, w, 8), device=device, generator=g).relu_()
    wb = torch.randn((5, 5, 128, 128), device=device, generator=g)
    ws = torch.randn((1, 1, 8, 128), device=device, generator=g)
    bias = torch.randn(128, device=device, generator=g)
    pkb = ops.pack_conv_weights(wb, wscale=float(np.sqrt(2.0 / 3200)), prec=prec)
    pks = ops.pack_conv_weights(ws, wscale=float(np.sqrt(2.0 / 8)), prec=prec)
    segs = [ops.Segment(a, pkb), ops.Segment(x, pks)]     # inputs converted to the G8 layout once, outside the loop

    def launch():
        return ops.conv2d_fused(segs, (h, w), bias=bias, act="relu", want_f32=False, want_g8=(prec != 2),
                                want_g8c=(prec == 2))

    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * (25 * 128 + 8) * 128 * h * w * n
    achieved = flops / (ms * 1e-3) / 1e12
    # HBM bytes per launch of this very launch from the committed rocprofv3 PMC passes (FETCH_SIZE x 2
    # per MI355X_MICROARCH.md + WRITE_SIZE; profiles/r01/roofline_pmc_b1convB.json); null for other modes
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01", "roofline_pmc_b1convB.json")
    if prec == 2 and os.path.exists(pmc):
        with open(pmc) as f:
            traffic = json.load(f)["hbm_bytes_per_launch"]
    return {
        "bound": "mfma",
        "kernel": "conv_mfma%s_kernel<NT=4> prec %d (resBlock1 convB 5x5 128->128 + 1x1 8->128 shortcut, 8 slices of 256^2)" % ("_f8" if prec == 2 else "", prec),
        "achieved": round(achieved, 2),
        "peak": DENSE_F16_MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / DENSE_F16_MFMA_PEAK_TFLOPS, 4),
        "traffic": traffic,
        "traffic_source": "profiles/r01/roofline_pmc_b1convB.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" if traffic else None,
        "launch_ms": round(ms, 4),
        "algorithmic_gflop_per_launch": round(flops / 1e9, 2),
        "mfma_products_per_mac": {3: "3 fp16", 2: "1 fp16 + 2 fp8 (MX, K=64)", 1: "1 fp16"}[prec],
    }


def cpu_baseline(p1, p2, low, slices=6):
    """the oracle's PyTorch-CPU twin (oracle/torch_ref.py) on a bounded sample of the same
    workload: `slices` slices of each pass of volume 0, extrapolated to 2 x 256 slices."""
    from oracle import multipass as OM
    from oracle import ops as O
    from oracle import torch_ref
    # the GPU box gives one GPU's share of the host (16 cores), not all of os.cpu_count()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, int(os.environ.get("MPG_CPU_THREADS", "16")))))
    xs = O.zoom_axis_linear(low, 0, UP)[S // 2 - slices // 2: S // 2 - slices // 2 + slices]
    torch_ref.gen_resnet(p1, xs[:1], UP, 2, True)      # warm up oneDNN
    t0 = time.time()
    r1 = torch_ref.gen_resnet(p1, xs, UP, 2, True)
    t1 = time.time()
    x2 = np.ascontiguousarray(np.broadcast_to(OM.cutoff(r1), (slices, S, S, 1)))
    t2 = time.time()
    torch_ref.gen_resnet(p2, x2, UP, 1, True)
    t3 = time.time()
    per_slice = ((t1 - t0) + (t3 - t2)) / (2 * slices)
    return {
        "value": round(1.0 / (per_slice * SLICES_PER_VOLUME), 6),
        "unit": "volumes/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": "%d slices of pass 1 + %d slices of pass 2 of one 64^3->256^3 volume with oracle/torch_ref.py "
                  "(PyTorch-CPU fp32; TF 1.x unavailable), extrapolated linearly to 512 slices; %.3f s/slice"
                  % (slices, slices, per_slice),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--volumes-per-gpu", type=int, default=8)
    ap.add_argument("--prec", type=int, default=2, choices=(1, 2, 3),
                    help="2 = MPG_PREC_F16F8 (default), 3 = MPG_PREC_F16X3, 1 = MPG_PREC_F16X1 (outside the 1e-3 tolerance)")
    ap.add_argument("--slice-batch", type=int, default=8, help="slices per generator launch (reference: 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="sharded", choices=("sharded", "replicas"),
                    help="N > 1: shard every volume's slices over the ranks with an all-gather between the passes "
                         "(default, north_star), or give every rank its own whole volumes and exchange nothing")
    args = ap.parse_args()

    import mpgan_amd
    from mpgan_amd import dist as mdist
    from mpgan_amd import multipass as MP
    from mpgan_amd.synthetic import synthetic_volume

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    comm, device = mdist.init_from_env()
    world = comm.world if comm is not None else 1
    rank = comm.rank if comm is not None else 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(device)

    n_vol = args.volumes_per_gpu * world
    cfg1 = dict(tile_low=SIM, up_res=UP, channels=1, upsampling_mode=2, batch_norm=True)
    cfg2 = dict(tile_low=SIM, up_res=UP, channels=1, upsampling_mode=1, batch_norm=True)
    g1 = MP.Generator("gen_resnet", cfg1, None, args.prec, device=device, seed=777)
    g2 = MP.Generator("gen_resnet", cfg2, None, args.prec, device=device, seed=778)
    replicas = args.mode == "replicas" and world > 1
    mine = range(rank * args.volumes_per_gpu, (rank + 1) * args.volumes_per_gpu) if replicas else range(n_vol)
    lows_np = [synthetic_volume(SIM, 1, i) for i in mine]
    lows = [torch.as_tensor(v).to(device) for v in lows_np]       # resident in HBM before the timed region

    def step():
        # the volumes are independent: their passes are pipelined by one volume so that the all-gather of
        # one volume's slabs overlaps the convolutions of its neighbours (N > 1); same arithmetic either way
        return MP.two_pass_4x_batch(g1, g2, lows, UP, batch=args.slice_batch, comm=None if replicas else comm)

    for _ in range(args.warmup):
        step()
    if comm is not None:
        comm.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = step()
    torch.cuda.synchronize(device)
    if comm is not None:
        comm.barrier()
    dt = time.perf_counter() - t0
    if comm is not None:
        dt = comm.max_float(dt, device)
    checksum = float(outs[0].double().sum().item())
    # parity of the timed arithmetic on the full-size volume 0: relative L2 against the F16X3 path
    # (itself held to 1e-4 of the oracle by tests/test_nets_gpu.py); north_star tolerance is 1e-3
    parity = None
    if rank == 0 and args.prec != 3:
        r1 = MP.Generator("gen_resnet", cfg1, g1.params(), 3, device=device)
        r2 = MP.Generator("gen_resnet", cfg2, g2.params(), 3, device=device)
        ref, _ = MP.two_pass_4x(r1, r2, lows[0], UP, batch=args.slice_batch)
        # (sharded runs all-gather the slabs, so volume 0 is whole on rank 0 in every mode)
        parity = float(((outs[0].double() - ref.double()).norm() / ref.double().norm()).item())
        del r1, r2, ref

    if rank != 0:
        return
    vol_per_s = n_vol * args.steps / dt
    result = {
        "metric": "volumes/sec (4x two-pass generator inference, 64^3->256^3 density-only)",
        "value": round(vol_per_s, 4),
        "unit": "volumes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {3: "f16x3->f32", 2: "f16+2xfp8->f32", 1: "f16->f32"}[args.prec],
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: 4x two-pass gen_resnet inference, 64^3->256^3 density-only, "
                        "%d volumes per GPU resident in HBM" % args.volumes_per_gpu,
            "volumes_per_step": n_vol,
            "slices_per_volume": SLICES_PER_VOLUME,
            "slice_batch": args.slice_batch,
            "parallelism": ("whole volumes per rank x%d, no exchange" % world if replicas else
                            "slice-axis sharding x%d + all-gather between passes, exchanges overlapped with the next volume" % world)
                           if world > 1 else "single GPU",
            "precision": {3: "MPG_PREC_F16X3 (fp16 hi/lo split, three fp16 MFMA products, fp32 accumulate)",
                          2: "MPG_PREC_F16F8 (fp16 product + two fp8 MX correction products, fp32 accumulate)",
                          1: "MPG_PREC_F16X1 (outside the 1e-3 tolerance)"}[args.prec],
        },
        "rel_l2_vs_f16x3_volume0": parity,
        "slices_per_s": round(vol_per_s * SLICES_PER_VOLUME, 2),
        "algorithmic_tflops": round(vol_per_s * SLICES_PER_VOLUME * gen_resnet_flops_per_slice() / 1e12, 2),
        "checksum_volume0": checksum,
    }
    result["roofline"] = dominant_kernel_roofline(mpgan_amd, device, args.prec)
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(g1.params(), g2.params(), lows_np[0])
    print(json.dumps(result))


if __name__ == "__main__":
    main()
