"""Launches on several HIP streams must reproduce the one-stream results bit for bit.

Guards the finding of profiles/r03/packed_fp32_followup.md on the hardware itself: a small-layer kernel built with
packed-fp32 FMAs (v_pk_fma_f32 .. op_sel:[0,1,0]) returned wrong sums in lanes 48-63 when it shared a CU with MFMA waves of
another stream's convolution (the library is therefore built with -fno-slp-vectorize -fno-vectorize, and
tests/test_abi_and_graph.py::test_no_packed_fp32_valu checks the code object)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _layer(ops, cin, cout, k, extra, seed, n=8, h=256):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn((n, h, h, cin), device=DEV, generator=g).relu_()
    w = torch.randn((k, k, cin, cout), device=DEV, generator=g)
    segs = [ops.Segment(x, ops.pack_conv_weights(w, wscale=0.05, prec=2))]
    if extra:
        x2 = torch.randn((n, h, h, extra), device=DEV, generator=g).relu_()
        w2 = torch.randn((1, 1, extra, cout), device=DEV, generator=g)
        segs.append(ops.Segment(x2, ops.pack_conv_weights(w2, wscale=0.05, prec=2)))
    return segs, (h, h)


def test_single_launches_next_to_mfma_launches_of_other_streams(mpg):
    from mpgan_amd import ops
    # the small two-segment layer that showed the fault, next to one-cout-tile MFMA layers (they can share a SIMD with it)
    jobs = [_layer(ops, 2, 1, 5, 8, 0), _layer(ops, 128, 32, 5, None, 1), _layer(ops, 32, 8, 5, 128, 2),
            _layer(ops, 8, 2, 5, None, 3), _layer(ops, 128, 32, 5, None, 4)]
    run = lambda j: ops.conv2d_fused(j[0], j[1], act="relu", want_f32=True)
    ref = []
    for j in jobs:
        ref.append(run(j).clone())
        torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in jobs]
    for rep in range(8):
        outs = []
        for st, j in zip(streams, jobs):
            with torch.cuda.stream(st):
                outs.append([run(j) for _ in range(3)])
        torch.cuda.synchronize()
        for i, os_ in enumerate(outs):
            for o in os_:
                assert torch.equal(o, ref[i]), "launch %d differs under concurrency (rep %d)" % (i, rep)


@pytest.mark.parametrize("prec", [2, 3])
def test_generator_calls_on_four_streams(mpg, prec):
    from mpgan_amd import multipass as MP
    cfg = dict(tile_low=64, up_res=4, channels=1, upsampling_mode=1, batch_norm=True)
    g = MP.Generator("gen_resnet", cfg, None, prec, device=DEV, seed=778)
    gens = [g] + [g.clone() for _ in range(3)]
    xs = [torch.rand((8, 256, 256, 1), device=DEV, generator=torch.Generator(device=DEV).manual_seed(i)) for i in range(4)]
    ref = []
    for gg, x in zip(gens, xs):
        ref.append(gg(x).clone())
        torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in gens]
    for rep in range(6):
        outs = []
        for _ in range(3):
            for st, gg, x in zip(streams, gens, xs):
                with torch.cuda.stream(st):
                    outs.append(gg(x))
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            assert torch.equal(o, ref[i % 4]), "call %d differs under concurrency (rep %d)" % (i, rep)


def test_four_channel_volumes_on_lanes(mpg):
    """whole 4-channel volumes (density + velocities: zooms, transposes, the channel marshalling kernel and both generators)
    as whole-volume lanes against the same volumes one after the other.  Everything between the passes is a kernel of this
    library on the lane's own stream: no tensor-library elementwise kernel runs beside the convolutions of another lane."""
    from mpgan_amd import multipass as MP
    from mpgan_amd.synthetic import synthetic_volume
    mk = lambda mode, seed: MP.Generator("gen_resnet", dict(tile_low=16, up_res=4, channels=4, upsampling_mode=mode, batch_norm=True),
                                         None, 2, device=DEV, seed=seed)
    g1, g2 = mk(2, 5), mk(1, 6)
    lows = [torch.as_tensor(synthetic_volume(16, 4, 11 + i), device=DEV) for i in range(4)]
    ref = []
    for low in lows:
        ref.append(MP.two_pass_4x(g1, g2, low, 4, batch=8, vel_scale=0.5)[0].clone())
        torch.cuda.synchronize()
    lanes = [(g1.clone(), g2.clone()) for _ in range(3)]
    for rep in range(4):
        outs = MP.two_pass_4x_batch(g1, g2, lows, 4, batch=8, vel_scale=0.5, lanes=lanes)
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            assert torch.equal(o, ref[i]), "volume %d differs on lanes (rep %d)" % (i, rep)
