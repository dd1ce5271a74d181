#!/usr/bin/env python3
"""Headline benchmark: 4x two-pass generator inference, 64^3 -> 256^3 density-only
(BASELINE.json configs[1], "C2"): a batch of 8 volumes per GPU, device-resident in,
device-resident out.

  python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process (which never touches the GPU) starts N
ranks of itself -- one process per GPU, RCCL -- relays rank 0's JSON line and exits with the worst
return code (multi-pass-gan_amd/launch.py).  Under torch.distributed.run the ranks are used as given.

One step = both passes of `volumes_per_gpu * N` synthetic volumes.  With N > 1 every volume's slices
are sharded over the ranks along the pass's slice axis and an RCCL all-gather reassembles the volume
between the passes (weak scaling: 8 volumes per GPU); the exchange-free alternative (whole volumes
per rank) is timed in the same job and reported as `value_replicas`.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# must be in the environment before the HIP runtime starts (dmabuf IPC for RCCL on this driver)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIM, UP = 64, 4
S = SIM * UP
SLICES_PER_VOLUME = 2 * S
# algorithmic work: sum over conv layers of 2*kh*kw*Cin*Cout*H*W at 256^2, C = 1
GFLOP_PER_SLICE = 71.692
DENSE_F16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense
PARITY_TOL = 1e-3                        # BASELINE.json north_star: relative L2 on density fields
PREC_NAME = {3: "f16x3", 2: "f16f6", 1: "f16x1"}
DTYPE_NAME = {3: "f16x3->f32", 2: "f16+2xbf6->f32", 1: "f16->f32"}
ROUND = "r03"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--volumes-per-gpu", type=int, default=8)
    ap.add_argument("--prec", type=int, default=2, choices=(1, 2, 3),
                    help="arithmetic of `value`: 2 = MPG_PREC_F16F6 (product default), 3 = MPG_PREC_F16X3, "
                         "1 = MPG_PREC_F16X1 (outside the 1e-3 tolerance)")
    ap.add_argument("--slice-batch", type=int, default=8, help="slices per generator launch (reference: 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (and the oracle parity it carries)")
    ap.add_argument("--no-second-prec", action="store_true", help="skip the second timed figure (value_f16x3)")
    ap.add_argument("--no-train-block", action="store_true", help="skip the `train_c3` block (BASELINE configs[2], ~5 s)")
    ap.add_argument("--cpu-slices", type=int, default=12, help="slices per pass of the CPU leg (>= 8)")
    ap.add_argument("--lanes", type=int, default=2,
                    help="HIP streams the independent volumes are dealt to (multipass.two_pass_4x_batch lanes)")
    ap.add_argument("--mode", default="both", choices=("both", "sharded", "replicas"),
                    help="N > 1: `sharded` = slices of every volume over the ranks + all-gather between the passes "
                         "(north_star; this is `value`), `replicas` = whole volumes per rank, no exchange; both = time both")
    ap.add_argument("--exchange", default="all_gather", choices=("all_gather", "all_to_all"),
                    help="N > 1, sharded mode: how a pass's slabs reach the next pass.  all_gather (north_star): the whole "
                         "volume to every rank; all_to_all: only the blocks each rank's planes need (1/N of the bytes).  "
                         "With --mode both the other one is timed too and reported as value_<exchange>")
    ap.add_argument("--workload", default="c2", choices=("c2", "c4"),
                    help="c2 (default, the headline): BASELINE configs[1]; c4: BASELINE configs[3], one 64^3 x 4 -> 512^3 "
                         "volume through the three 8x generators, slices sharded over the ranks")
    ap.add_argument("--launch-timeout", type=float, default=3000.0)
    return ap.parse_args(argv)


def gen_resnet_flops_per_slice(c=1, hw=256 * 256):
    f = 0
    widths = [(c, 2 * c, 8 * c), (8 * c, 128, 128), (128, 32, 8), (8, 2, 1)]
    for cin, s1, s2 in widths:
        f += 2 * 25 * cin * s1 * hw + 2 * 25 * s1 * s2 * hw + 2 * cin * s2 * hw
    return f


def launch_self(args, argv):
    """parent of an N-rank run: no GPU call in this process.  A failed sharded run (the north_star configuration: slice-axis
    sharding + RCCL all-gather) is a FAILED benchmark: the exchange-free partition is still measured, by a fresh job, but
    only as `value_replicas` next to `value: null`, and the exit code is non-zero."""
    from mpgan_amd import launch
    rc, out = launch.spawn_ranks(os.path.abspath(__file__), argv, args.gpus, timeout=args.launch_timeout)
    line = launch.last_json_line(out)
    if (rc != 0 or line is None) and args.mode != "replicas":
        sys.stderr.write(out)
        sys.stderr.write("bench.py: the sharded run failed (rc %s); measuring whole volumes per rank for the record\n" % rc)
        rc2, out2 = launch.spawn_ranks(os.path.abspath(__file__), list(argv) + ["--mode", "replicas", "--no-cpu-baseline",
                                                                               "--no-second-prec"],
                                       args.gpus, timeout=args.launch_timeout)
        line2 = launch.last_json_line(out2)
        res = failed_line(args, "sharded run: rc %s" % rc)
        if rc2 == 0 and line2 is not None:
            rep = json.loads(line2)
            res["value_replicas"] = rep.get("value")
            res["ms_per_step_replicas"] = rep.get("ms_per_step")
            res["config"] = rep.get("config", res["config"])
        print(json.dumps(res))
        sys.stdout.flush()
        return rc or 1
    if line is not None:
        print(line)
    else:
        sys.stderr.write(out)
        rc = rc or 1
    sys.stdout.flush()
    return rc


def failed_line(args, err):
    """the JSON line of a run whose sharded pipeline failed: no `value`"""
    return {"metric": "volumes/sec (4x two-pass generator inference, 64^3->256^3 density-only)", "value": None,
            "unit": "volumes/s", "n_gpus": args.gpus, "rccl_ranks": 0, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_NAME[args.prec], "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 4x two-pass gen_resnet inference, 64^3->256^3 density-only, "
                                   "%d volumes per GPU resident in HBM" % args.volumes_per_gpu},
            "sharded_error": err}


def _time_events(fn, iters, device, torch):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) / iters


def dominant_kernel_roofline(gen2, x_batch, device, prec, iters=20, pipeline_pass=None):
    """The dominant launch: resBlock 1's B conv (5x5 128->128 plus its 1x1 8->128 shortcut as a second
    K-segment, multipassGAN-4x.py:561) on one batch of 8 slices of 256^2.  Timed twice with events on the
    launch stream: on the activations the pipeline itself produced for `x_batch` (tapped from the running
    session; this is `achieved`) and on dense random data (`launch_ms_random`)."""
    import numpy as np
    import torch
    from mpgan_amd import ops
    n, h, w = 8, S, S
    flops = 2.0 * (25 * 128 + 8) * 128 * h * w * n
    sess = gen2.sess
    sess.tap, sess.tapped = "g_cB1/", None
    gen2(x_batch)
    call = sess.tapped
    sess.tapped = None
    # in the pipeline: every launch of this layer during one more pass over a volume (single stream, so that no other
    # lane's kernels share the CUs), bracketed by HIP events on the launch stream
    ms_inpipe = n_inpipe = None
    if pipeline_pass is not None:
        sess.tap_events = []
        pipeline_pass()
        torch.cuda.synchronize(device)
        t = [a.elapsed_time(b) for a, b in sess.tap_events]
        sess.tap_events = None
        if t:
            ms_inpipe, n_inpipe = sum(t) / len(t), len(t)
    sess.tap = None
    ms_pipe = None
    if call is not None:
        for _ in range(3):
            ops.conv2d_fused(**call)
        ms_pipe = _time_events(lambda: ops.conv2d_fused(**call), iters, device, torch)
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
        return ops.conv2d_fused(segs, (h, w), bias=bias, act="relu", want_f32=False, want_g8=True)

    for _ in range(3):
        launch()
    ms_rand = _time_events(launch, iters, device, torch)
    ms = ms_inpipe if ms_inpipe is not None else (ms_pipe if ms_pipe is not None else ms_rand)
    achieved = flops / (ms * 1e-3) / 1e12
    board = board_under_load(lambda: ops.conv2d_fused(**call) if call is not None else launch(), device, torch)
    # HBM bytes per launch of this very launch from the committed rocprofv3 PMC passes (FETCH_SIZE x 2
    # per MI355X_MICROARCH.md + WRITE_SIZE); null for other modes
    traffic = src = None
    for rnd in (ROUND, "r02", "r01"):
        pmc = os.path.join(ROOT, "profiles", rnd, "roofline_pmc_b1convB.json")
        if prec == 2 and os.path.exists(pmc):
            with open(pmc) as f:
                traffic = json.load(f)["hbm_bytes_per_launch"]
            src = "profiles/%s/roofline_pmc_b1convB.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % rnd
            break
    return {
        "bound": "mfma",
        "kernel": "conv_mfma%s_kernel<NT=4> prec %d (resBlock1 convB 5x5 128->128 + 1x1 8->128 shortcut, 8 slices of 256^2)" % ("_f6" if prec == 2 else "", prec),
        "achieved": round(achieved, 2),
        "peak": DENSE_F16_MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / DENSE_F16_MFMA_PEAK_TFLOPS, 4),
        "traffic": traffic,
        "traffic_source": src,
        "launch_ms": round(ms, 4),
        "launch_ms_data": ("mean over the %d launches of this layer in pass 2 of volume 0, HIP events around each launch on its "
                           "stream inside the running pipeline (one lane)" % n_inpipe) if ms_inpipe is not None else
                          ("replay of one launch on activations taken from the pipeline" if ms_pipe is not None else "dense random"),
        "launch_ms_replay": round(ms_pipe, 4) if ms_pipe is not None else None,
        "launch_ms_random": round(ms_rand, 4),
        "algorithmic_gflop_per_launch": round(flops / 1e9, 2),
        "board_under_this_launch": board,
        "mfma_products_per_mac": {3: "3 fp16", 2: "1 fp16 + 2 bf6 (MX e3m2, K=64 in the cycles of one fp16 K=16): 1.5 fp16-equivalent units", 1: "1 fp16"}[prec],
    }


def board_under_load(fn, device, torch, secs=1.5):
    """package power and shader clock (rocm-smi, polled from a thread) while `fn` is launched back to back for `secs`: the
    layer kernels run at the board's power limit, and the clock is what the limit leaves (profiles/r03/power_clocks.txt);
    None where rocm-smi is not there or prints something else"""
    import re
    import subprocess
    import threading
    rows, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "-d", "0", "-P", "-c", "--csv"], capture_output=True, text=True, timeout=5).stdout
                rows.append(out.strip().splitlines()[-1])
            except Exception:                      # noqa: BLE001 -- a probe, never a reason to fail the bench
                return
            time.sleep(0.2)
    try:
        th = threading.Thread(target=poll, daemon=True)
        th.start()
        t0 = time.time()
        while time.time() - t0 < secs:
            for _ in range(20):
                fn()
            torch.cuda.synchronize(device)
        stop.set()
        th.join(timeout=6)
        clk, pw = [], []
        for r in rows[1:]:                          # the first sample may predate the load
            f = r.split(",")
            m = [int(v) for v in re.findall(r"\((\d+)Mhz\)", r)]
            if len(m) >= 3 and f[-1].replace(".", "", 1).isdigit():
                clk.append(m[2])
                pw.append(float(f[-1]))
        if not clk:
            return None
        return {"sclk_mhz": int(sum(clk) / len(clk)), "package_power_w": round(sum(pw) / len(pw), 1), "samples": len(clk),
                "source": "rocm-smi -P -c polled while the launch replays back to back for %.1f s" % secs}
    except Exception:                              # noqa: BLE001
        return None


def train_c3_block(device, steps=20, warmup=3):
    """BASELINE configs[2] ("C3") next to the headline, outside its timed region: one 4x training iteration (G + D
    forward / backward + both Adam updates; 16 tiles of 16 -> 64^2, density + velocity, batch norm, spatial discriminator,
    replayed from the captured hipGraph) and the matrix-core weight gradient of its widest layer.  Same code as
    `python bench_train.py`, which also has the CPU port beside it and the 256^2 / 8x variants."""
    import numpy as np
    import torch
    import bench_train as BT
    from mpgan_amd.train import Trainer4x
    BT.torch = torch
    tile, batch, ch = 16, 16, 4
    rng = np.random.default_rng(0)
    tr = Trainer4x(tileSizeLow=tile, upRes=4, n_inputChannels=ch, batch_norm=True, device=str(device))
    xs = torch.as_tensor(rng.random((batch, tile * tile * ch)).astype(np.float32), device=device)
    ys = torch.as_tensor(rng.random((batch, (tile * 4) ** 2)).astype(np.float32), device=device)
    for _ in range(warmup):
        tr.train_step_graphed(xs, ys)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        d, g = tr.train_step_graphed(xs, ys)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    gf, df = BT.fwd_flops_per_tile(tile * 4, ch)
    flops = batch * ((gf + 2 * df + 2 * 2 * df) + (gf + 2 * df + df + 2 * gf))
    roof = BT.wgrad_roofline(device, tile * 4, batch)
    return {"workload": "BASELINE configs[2]: 4x training step, tileSize 16 -> 64^2, batch 16, density+velocity, batchNorm, "
                        "spatial discriminator, hipGraph replay", "iterations_per_s": round(1.0 / dt, 2),
            "ms_per_iteration": round(dt * 1e3, 3), "steps": steps, "tiles_per_s": round(batch / dt, 1),
            "algorithmic_tflops": round(flops / dt / 1e12, 1), "dtype": "f16x3->f32",
            "disc_loss": float(d), "gen_loss_complete": float(g), "wgrad_roofline": roof}


def _rel_l2(a, b):
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def cpu_baseline_and_parity(p1, p2, low, gpu_runs, slices=12):
    """The oracle's PyTorch-CPU twin (oracle/torch_ref.py) on a bounded sample of the same workload --
    `slices` slices of each pass of volume 0 (first / middle / last thirds of the slice axis), timed and
    extrapolated to 2 x 256 slices -- and, with those very slices, the parity of the GPU volumes:
    pass 1 directly, pass 2 evaluated by the oracle on the planes of the GPU's own pass-1 volume.
    gpu_runs: {name: (final [z,y,x], pass-1 volume [z,y,x])} as numpy arrays."""
    import numpy as np
    import torch
    from oracle import multipass as OM
    from oracle import ops as O
    from oracle import torch_ref
    # the GPU box gives one GPU's share of the host (16 cores), not all of os.cpu_count()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, int(os.environ.get("MPG_CPU_THREADS", "16")))))
    third = max(slices // 3, 1)
    idx = sorted(set(list(range(third)) + list(range(S // 2 - third // 2, S // 2 - third // 2 + third))
                     + list(range(S - (slices - 2 * third), S))))
    xs = O.zoom_axis_linear(low, 0, UP)[idx]
    torch_ref.gen_resnet(p1, xs[:1], UP, 2, True)      # warm up oneDNN
    t0 = time.time()
    r1 = torch_ref.gen_resnet(p1, xs, UP, 2, True)
    t1 = time.time()
    r1 = OM.cutoff(r1[..., 0])                                       # the pass-1 file carries the cutoff (4x.py:1156)
    cpu_s = t1 - t0
    parity = {}
    for name, (final, v1) in gpu_runs.items():
        x2 = np.ascontiguousarray(v1[:, :, idx].transpose(2, 0, 1))[..., None]   # [x][z][y] planes (4x.py:1113)
        t2 = time.time()
        r2 = torch_ref.gen_resnet(p2, x2, UP, 1, True)
        t3 = time.time()
        if name == next(iter(gpu_runs)):
            cpu_s += t3 - t2
        want = OM.cutoff(r2[..., 0])                                 # [x][z][y]
        got = final[:, :, idx].transpose(2, 0, 1)
        parity[name] = {"pass1": _rel_l2(v1[idx], r1), "pass2_given_pass1": _rel_l2(got, want)}
    per_slice = cpu_s / (2 * len(idx))
    base = {
        "value": round(1.0 / (per_slice * SLICES_PER_VOLUME), 6),
        "unit": "volumes/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": "%d slices of pass 1 + %d slices of pass 2 of one 64^3->256^3 volume (first/middle/last of each slice "
                  "axis) with oracle/torch_ref.py (PyTorch-CPU fp32; TF 1.x unavailable), extrapolated linearly to 512 "
                  "slices; %.3f s/slice" % (len(idx), len(idx), per_slice),
    }
    return base, parity, idx


C4_CFG = [dict(first_gen=True, filter_size=3, start_fms=256, max_fms=256, add_adj=True, first_nn_arch=True, use_res_net=True),
          dict(first_gen=False, filter_size=5, start_fms=192, max_fms=192, use_res_net=True),
          dict(first_gen=False, filter_size=5, start_fms=192, max_fms=96, use_res_net=False)]    # example_run_output.py:18-47
C4_GFLOP_PER_SLICE = [136.63, 405.48, 397.21]


def main_c4(args):
    """BASELINE configs[3]: 8x three-pass inference, 64^3 density + velocity -> 512^3, the slices of every pass sharded over
    the ranks (multipass.multipass_8x with `comm`), same JSON shape as the headline.  One step = one volume (strong
    scaling: the volume is what it is; with N ranks every rank evaluates 512 / N slices per pass)."""
    import torch
    import mpgan_amd  # noqa: F401
    from mpgan_amd import dist as mdist
    from mpgan_amd import multipass as MP
    from mpgan_amd.synthetic import synthetic_volume
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    comm, device = mdist.init_from_env()
    world = comm.world if comm is not None else 1
    rank = comm.rank if comm is not None else 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    torch.cuda.set_device(device)
    MP.set_pass_lanes(args.lanes)
    gens = [MP.Generator("growing_gen", dict(tile_low=SIM, up_res=8, channels=4, **c), None, args.prec, device=device, seed=100 + i)
            for i, c in enumerate(C4_CFG)]
    low = torch.as_tensor(synthetic_volume(SIM, 4, 0)).to(device)

    def timed(exchange):
        step = lambda: MP.multipass_8x(gens, low, 8, batches=(8, 2, 2), comm=comm, exchange=exchange)
        for _ in range(args.warmup):
            step()
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize(device)
        if comm is not None:
            comm.barrier()
        dt = time.perf_counter() - t0
        return (comm.max_float(dt, device) if comm is not None else dt), out
    try:
        dt, out = timed(args.exchange)
    except Exception as e:                       # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:160] if str(e) else "")
        sys.stderr.write("bench.py rank %d: the sharded run failed (%s)\n" % (rank, err))
        if rank == 0:
            print(json.dumps(failed_line(args, err)))
            sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3)
    extra = {}
    if world > 1 and args.mode == "both":
        other = "all_to_all" if args.exchange == "all_gather" else "all_gather"
        dto, _ = timed(other)
        extra["value_" + other] = round(args.steps / dto, 4)
    if rank != 0:
        return 0
    s8 = SIM * 8
    tflop = sum(C4_GFLOP_PER_SLICE) * s8 / 1e3
    res = {
        "metric": "volumes/sec (8x three-pass generator inference, 64^3->512^3 density+velocity)",
        "value": round(args.steps / dt, 4), "unit": "volumes/s", "n_gpus": world, "rccl_ranks": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": DTYPE_NAME[args.prec], "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: 8x three-pass growing_gen inference (example_run_output.py:18-47), one "
                               "64^3 x 4-channel volume -> 512^3 per step", "slices_per_volume": 3 * s8, "lanes": max(args.lanes, 1),
                   "parallelism": ("slice-axis sharding x%d + %s between passes" % (world, args.exchange)) if world > 1 else "single GPU"},
        "slices_per_s": round(3 * s8 * args.steps / dt, 2),
        "algorithmic_tflops": round(tflop * args.steps / dt, 2),
        "checksum_volume": float(out.double().sum().item()),
    }
    res.update(extra)
    print(json.dumps(res))
    sys.stdout.flush()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_self(args, argv))
    if args.workload == "c4":
        return main_c4(args)

    import torch
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
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    torch.cuda.set_device(device)
    MP.set_pass_lanes(args.lanes)

    n_vol = args.volumes_per_gpu * world
    cfg1 = dict(tile_low=SIM, up_res=UP, channels=1, upsampling_mode=2, batch_norm=True)
    cfg2 = dict(tile_low=SIM, up_res=UP, channels=1, upsampling_mode=1, batch_norm=True)
    g1 = MP.Generator("gen_resnet", cfg1, None, args.prec, device=device, seed=777)
    g2 = MP.Generator("gen_resnet", cfg2, None, args.prec, device=device, seed=778)
    lows_np = [synthetic_volume(SIM, 1, i) for i in range(n_vol)]
    lows = [torch.as_tensor(v).to(device) for v in lows_np]       # resident in HBM before the timed region
    mine = lows[rank * args.volumes_per_gpu:(rank + 1) * args.volumes_per_gpu]

    def timed(ga, gb, replicas, exchange=args.exchange):
        """W warm-up steps, then exactly K steps between barrier + synchronize; max over ranks"""
        lanes = [(ga.clone(), gb.clone()) for _ in range(max(args.lanes, 1) - 1)]

        def step():
            # the volumes are independent: their passes are pipelined by one volume so that the all-gather of
            # one volume's slabs overlaps the convolutions of its neighbours (N > 1); same arithmetic either way
            if replicas:
                return MP.two_pass_4x_batch(ga, gb, mine, UP, batch=args.slice_batch, comm=None, lanes=lanes)
            return MP.two_pass_4x_batch(ga, gb, lows, UP, batch=args.slice_batch, comm=comm, lanes=lanes, exchange=exchange)
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
        return dt, outs

    sharded_first = args.mode in ("both", "sharded") or world == 1
    extra = {}
    if world > 1 and sharded_first:
        # the sharded pipeline is the only part that needs RCCL.  A rank on which it raises does not go on computing on a
        # runtime that has just thrown: it reports and exits non-zero (the launcher -- launch.spawn_ranks or
        # torch.distributed.run -- then ends the other ranks); a self-launched job is followed by a fresh replicas job
        try:
            dt, outs = timed(g1, g2, replicas=False)
        except Exception as e:                       # noqa: BLE001 -- anything RCCL / the runtime throws
            err = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:160] if str(e) else "")
            sys.stderr.write("bench.py rank %d: the sharded run failed (%s)\n" % (rank, err))
            if rank == 0:
                print(json.dumps(failed_line(args, err)))
                sys.stdout.flush()
            sys.stderr.flush()
            os._exit(3)
    else:
        dt, outs = timed(g1, g2, replicas=not sharded_first)
    checksum = float(outs[0].double().sum().item())
    del outs
    if world > 1 and args.mode == "both":
        dt_r, o = timed(g1, g2, replicas=True)
        del o
        extra["value_replicas"] = round(n_vol * args.steps / dt_r, 4)
        extra["ms_per_step_replicas"] = round(dt_r / args.steps * 1e3, 3)
        other = "all_to_all" if args.exchange == "all_gather" else "all_gather"
        dt_o, o = timed(g1, g2, replicas=False, exchange=other)
        del o
        extra["value_" + other] = round(n_vol * args.steps / dt_o, 4)
    second = None
    if not args.no_second_prec and args.prec != 3:
        r1 = MP.Generator("gen_resnet", cfg1, g1.params(), 3, device=device)
        r2 = MP.Generator("gen_resnet", cfg2, g2.params(), 3, device=device)
        dt3, o = timed(r1, r2, replicas=not sharded_first)
        del o
        extra["value_f16x3"] = round(n_vol * args.steps / dt3, 4)
        extra["ms_per_step_f16x3"] = round(dt3 / args.steps * 1e3, 3)
        second = (r1, r2)

    if rank != 0:
        return 0
    vol_per_s = n_vol * args.steps / dt
    replicas_only = world > 1 and args.mode == "replicas"
    result = {
        "metric": "volumes/sec (4x two-pass generator inference, 64^3->256^3 density-only)",
        "value": round(vol_per_s, 4),
        "unit": "volumes/s",
        "n_gpus": world,
        "rccl_ranks": 0 if (world > 1 and args.mode == "replicas") else world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE_NAME[args.prec],
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: 4x two-pass gen_resnet inference, 64^3->256^3 density-only, "
                        "%d volumes per GPU resident in HBM" % args.volumes_per_gpu,
            "volumes_per_step": n_vol,
            "slices_per_volume": SLICES_PER_VOLUME,
            "slice_batch": args.slice_batch,
            "lanes": max(args.lanes, 1),
            "parallelism": ("whole volumes per rank x%d, no exchange" % world if replicas_only else
                            "slice-axis sharding x%d + %s between passes, exchanges overlapped with the next volume" % (world, args.exchange))
                           if world > 1 else "single GPU",
            "precision": {3: "MPG_PREC_F16X3 (fp16 hi/lo split, three fp16 MFMA products, fp32 accumulate)",
                          2: "MPG_PREC_F16F6 (fp16 product + two bf6 MX block-scaled correction products, fp32 accumulate; the "
                             "default of the inference drivers, `prec` parameter)",
                          1: "MPG_PREC_F16X1 (outside the 1e-3 tolerance)"}[args.prec],
        },
        "slices_per_s": round(vol_per_s * SLICES_PER_VOLUME, 2),
        "algorithmic_tflops": round(vol_per_s * SLICES_PER_VOLUME * gen_resnet_flops_per_slice() / 1e12, 2),
        "checksum_volume0": checksum,
    }
    result.update(extra)
    # one un-timed run of volume 0 per arithmetic: the volumes the parity figures are taken on
    gpu_runs = {}
    f0, v0 = MP.two_pass_4x(g1, g2, lows[0], UP, batch=args.slice_batch)
    gpu_runs[PREC_NAME[args.prec]] = (f0.cpu().numpy(), v0.cpu().numpy())
    if second is not None:
        f3, v3 = MP.two_pass_4x(second[0], second[1], lows[0], UP, batch=args.slice_batch)
        result["rel_l2_vs_f16x3_volume0"] = float(((f0.double() - f3.double()).norm() / f3.double().norm()).item())
        gpu_runs["f16x3"] = (f3.cpu().numpy(), v3.cpu().numpy())
        del f3, v3
    x_batch = MP.ops.volume_transpose(v0, (2, 0, 1)).reshape(S, S, S, 1)[120:128].contiguous()
    def one_more_pass():
        MP.set_pass_lanes(1)
        try:
            MP.two_pass_4x(g1, g2, lows[0], UP, batch=args.slice_batch)
        finally:
            MP.set_pass_lanes(args.lanes)

    result["roofline"] = dominant_kernel_roofline(g2, x_batch, device, args.prec, pipeline_pass=one_more_pass)
    ok = True
    if world == 1 and not args.no_cpu_baseline:
        base, parity, idx = cpu_baseline_and_parity(g1.params(), g2.params(), lows_np[0], gpu_runs,
                                                    max(args.cpu_slices, 8))
        result["cpu_baseline"] = base
        result["rel_l2_vs_oracle"] = parity
        result["rel_l2_vs_oracle_slices"] = "volume 0, slice indices %s of each pass; tolerance %.0e" % (idx, PARITY_TOL)
        worst = max(max(v.values()) for v in parity.values())
        ok = worst <= PARITY_TOL
        result["parity_ok"] = bool(ok)
    if world == 1 and not args.no_train_block:
        try:
            result["train_c3"] = train_c3_block(device)
        except Exception as e:               # noqa: BLE001 -- the headline line must not die with the side block
            result["train_c3"] = {"error": "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")}
    print(json.dumps(result))
    sys.stdout.flush()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
