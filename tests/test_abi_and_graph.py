"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol
include/mpgan.h declares; the GAN builder creates the reference's variable names; the
session fuses the graphs into the expected launches (no GPU, no compute)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(mpg):
    from mpgan_amd import _lib
    text = open(os.path.join(ROOT, "include", "mpgan.h")).read()
    declared = set(re.findall(r"\b(mpg_[a-z0-9_]+)\s*\(", text))
    declared -= {"mpg_conv_seg", "mpg_conv_desc"}
    assert len(declared) >= 15
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(_lib.PROTOTYPES)
    assert lib.mpg_version().startswith(b"mpgan-hip")
    # host-side helper: no GPU needed
    # 16 chunks of one channel group x 13 one-k-step stages x (4 cout tiles x 1 KiB x hi/lo)
    assert lib.mpg_conv_pack_size(5, 5, 128, 128, 3) == 16 * 13 * 1 * 4 * 1024 * 2
    assert lib.mpg_conv_pack_size(5, 5, 128, 200, 3) == 0
    assert lib.mpg_g8_bytes(2, 16, 32, 11) == 2 * 2 * 2 * 16 * 32 * 16


def test_struct_layout_matches_header(mpg):
    import ctypes
    from mpgan_amd import _lib
    assert ctypes.sizeof(_lib.ConvSeg) == 48
    assert _lib.ConvDesc.seg.offset == 24 and ctypes.sizeof(_lib.ConvDesc) == 24 + 4 * 48 + 64 + 8
    assert _lib.ConvDesc.y_g8.offset == 264 and _lib.ConvDesc.prec.offset == 272 and _lib.ConvDesc.reserved.offset == 276
    assert _lib.ConvDesc.in_amax.offset == 280


def test_compute_refuses_without_gpu(mpg):
    import torch
    from mpgan_amd import _lib, ops
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(_lib.MpgError):
        ops.cutoff(torch.zeros(4))
    from mpgan_amd import multipass as MP
    g = MP.Generator("gen_resnet", dict(tile_low=8, up_res=4, channels=1), None)
    with pytest.raises(_lib.MpgError):
        g(torch.zeros(1, 8, 8, 1))


def test_builder_variable_names_match_reference_scopes(mpg):
    from mpgan_amd import multipass as MP
    from oracle import nets as ON
    ps = ON.ParamSource()
    ON.gen_resnet(ps, np.zeros((1, 4, 4, 4), np.float32), 4, 2, True)
    g = MP.Generator("gen_resnet", dict(tile_low=4, up_res=4, channels=4, upsampling_mode=2), None)
    assert list(g.graph.variables) == ps.order
    assert {k: v.shape for k, v in g.graph.variables.items()} == {k: v.shape for k, v in ps.params.items()}
    assert "generator/g_cA0/moving_variance" in g.graph.variables and "generator/g_s3/bias" in g.graph.variables
    cfgs = [dict(first_gen=True, filter_size=3, start_fms=256, max_fms=256, add_adj=True, first_nn_arch=True),
            dict(first_gen=False, filter_size=5, start_fms=192, max_fms=192),
            dict(first_gen=False, filter_size=5, start_fms=192, max_fms=96, use_res_net=False)]
    for c in cfgs:
        ps = ON.ParamSource()
        x = np.zeros((1, 4, 4, 6 if c["first_gen"] else 4), np.float32)
        if c["first_gen"]:
            ON.growing_gen(ps, x, 8, True, 3, 256, 256, True, True)
        else:
            ON.growing_gen(ps, ON.gen2_input(np.zeros((1, 32, 32, 1), np.float32), x, 32), 8, False, c["filter_size"],
                           c["start_fms"], c["max_fms"], False, c.get("use_res_net", True))
        g = MP.Generator("growing_gen", dict(tile_low=4, up_res=8, channels=4, **c), None)
        assert sorted(g.graph.variables) == sorted(ps.params)
    assert "generator/genBlock8/g_cdensOut8/weight" in g.graph.variables


def test_gan_layer_side_effects(mpg):
    from mpgan_amd import graph as G
    from mpgan_amd.GAN import GAN, lrelu
    G.reset_default_graph()
    x = G.placeholder([None, 8, 8, 3])
    gan = GAN(x)
    a, lin = gan.convolutional_layer(16, [3, 3], lrelu, name="c1")
    assert gan.layer is a and lin.op == "bias_add" and a.attrs["act"] == "lrelu"
    b, _ = gan.convolutional_layer(4, [1, 1], None, name="c2")           # reads self.layer
    assert b.inputs[0].inputs[0] is a and gan.layer is b
    gan.max_depool(height_factor=2, width_factor=2)
    assert gan.layer.shape == (None, 16, 16, 4)
    gan.avg_depool(mode=2, scale=[2])
    assert gan.layer.shape == (None, 32, 32, 4) and gan.layer.attrs["method"] == 2
    gan.avg_pool()
    assert gan.layer.shape == (None, 16, 16, 4)
    n = gan.flatten()
    assert n == 16 * 16 * 4
    gan.fully_connected_layer(1, None, name="fc", gain=1)
    assert gan.y().shape == (None, 1)
    assert gan.getDOFs() == 3 * 3 * 3 * 16 + 16 + 16 * 4 + 4 + n + 1
    assert list(G.get_default_graph().variables) == ["c1/weight", "c1/bias", "c2/weight", "c2/bias", "fc/weight", "fc/bias"]
    gan2 = GAN(G.placeholder([None, 8, 8, 3]))
    d, dlin = gan2.deconvolutional_layer(5, [4, 4], lrelu, stride=[2], name="up")      # GAN.py:566-619 (TypeError in the reference)
    assert d.shape == (None, 16, 16, 5) and dlin.op == "bias_add" and dlin.inputs[0].op == "conv2d_transpose"
    assert G.get_default_graph().variables["up/weight"].shape == (4, 4, 5, 3)
    gan2.pixel_shuffle(upres=2, stage="1")                                                # GAN.py:554-560
    assert gan2.layer.shape == (None, 32, 32, 5) and gan2.layer.op == "depth_to_space"
    assert G.get_default_graph().variables["g_cPS1/weight"].shape == (1, 1, 5, 20)
    G.reset_default_graph()


def test_fusion_plan(mpg):
    from mpgan_amd import multipass as MP
    g = MP.Generator("gen_resnet", dict(tile_low=8, up_res=4, channels=1, upsampling_mode=2), None, prec=3)
    launches = [e for e in g.sess.plan_summary(g.sampler) if e["kind"] in ("conv2d_fused", "conv2d_small_pair")]
    # resBlock 0 (1 -> 2 -> 8) is one launch, the middle tensor stays in LDS; resBlock 3 (8 -> 2 -> 1) stays two launches: the
    # halo recomputation of an 8-channel first convolution costs more than the launch saved (ops.small_pair_ok)
    assert [(e["kind"], e["cout"]) for e in launches] == [("conv2d_small_pair", 8), ("conv2d_fused", 128), ("conv2d_fused", 128),
                                                         ("conv2d_fused", 32), ("conv2d_fused", 8), ("conv2d_fused", 2),
                                                         ("conv2d_fused", 1)]
    assert [e["cmid"] for e in launches if e["kind"] == "conv2d_small_pair"] == [2]
    # activations between fused launches travel as G8 only; the fetched tensor is fp32
    assert [e["emit"] for e in launches[:-1]] == [{"f32": False, "g8": True}] * 6
    assert launches[-1]["emit"] == {"f32": True, "g8": False}
    # F16F6: every launch of this net has 1 or 4 cout tiles, so all of it runs in that mode
    gf = MP.Generator("gen_resnet", dict(tile_low=8, up_res=4, channels=1, upsampling_mode=2), None)   # default: F16F6
    lf = [e for e in gf.sess.plan_summary(gf.sampler) if e["kind"] in ("conv2d_fused", "conv2d_small_pair")]
    # ... except the 8 -> 128 layer, whose short contraction (K = 200) runs on the three-product fp16 kernel
    assert [e["prec"] for e in lf] == [2, 3, 2, 2, 2, 2, 2] and all(e["emit"]["g8"] and not e["emit"]["f32"] for e in lf[:-1])
    # a per-launch precision map mixes modes; every mode reads the same G8 tensor
    g8x = MP.Generator("growing_gen", dict(tile_low=8, up_res=8, channels=4, first_gen=True, filter_size=3, start_fms=256,
                                           max_fms=256, add_adj=True, first_nn_arch=True), None, prec=2,
                       prec_map=[("genBlock4/g_cA_second", 3)])
    l8 = [e for e in g8x.sess.plan_summary(g8x.sampler) if e["kind"] == "conv2d_fused"]
    assert set(e["prec"] for e in l8) == {2, 3}
    assert [len(e["segments"]) for e in launches] == [2, 1, 2, 1, 2, 1, 2]
    assert launches[0]["segments"][0]["up_log2"] == 2 and launches[0]["segments"][1]["up_log2"] == 2
    assert all(e["act"] == "relu" for e in launches)
    other = [e["kind"] for e in g.sess.plan_summary(g.sampler) if e["kind"] not in ("conv2d_fused", "conv2d_small_pair")]
    assert set(other) <= {"reshape"}
    g2 = MP.Generator("growing_gen", dict(tile_low=8, up_res=8, channels=4, first_gen=False, filter_size=5,
                                          start_fms=192, max_fms=192), None)
    plan = [e for e in g2.sess.plan_summary(g2.sampler) if e["kind"] == "conv2d_fused"]
    # first conv reads concat(y, nearest_x8(x)) as two segments; its resblock partner adds the two 1x1 segments
    assert [(s["cin"], s["up_log2"]) for s in plan[0]["segments"]] == [(1, 0), (4, 3)]
    assert [(s["cin"], s["up_log2"], s["kernel"]) for s in plan[1]["segments"]] == [(16, 0, (5, 5)), (1, 0, (1, 1)), (4, 3, (1, 1))]
    assert plan[-1]["cout"] == 1 and plan[-1]["post_add"] is not None and plan[0]["pixel_norm"]


def test_trainer_builds_on_cpu_and_refuses_to_run_without_gpu():
    """graph construction and parameter grouping need no GPU; the step itself has no CPU fallback"""
    import torch
    from mpgan_amd import _lib
    from mpgan_amd.train import Trainer4x
    tr = Trainer4x(tileSizeLow=8, upRes=4, n_inputChannels=4, batch_norm=True)
    assert len(tr.opt_g.names) == 4 * 3 * 2 + 3 * 3 * 2          # weights+biases of 12 convs, gamma+beta of 9
    assert len(tr.opt_d.names) == 5 * 2 + 3 * 2
    assert all("g_" in n for n in tr.opt_g.names) and all("d_" in n for n in tr.opt_d.names)
    if not torch.cuda.is_available():
        with pytest.raises(_lib.MpgError):
            tr.losses(np.zeros((2, 8 * 8 * 4), np.float32), np.zeros((2, 32 * 32), np.float32))


def test_architecture_tables(mpg):
    """multi-pass-gan_amd/arch.py holds the networks as tables; the widths are those of SURVEY.md appendix A
    (example_run_output.py:18-47 for the three 8x generators, multipassGAN-8x.py:752-846 for the critics)"""
    from mpgan_amd import arch
    assert [u for u, _ in arch.gen_resnet_table(4)] == [("res", "0", 8, 32), ("res", "1", 128, 128), ("res", "2", 32, 8),
                                                       ("res", "3", 2, 1)]
    assert [bn for _, bn in arch.gen_resnet_table(1)] == [True, True, True, False]
    stem, blocks = arch.growing_gen_table(256, 256, 3, True, True)           # net 1
    assert stem == () and [(up, len(us)) for up, us in blocks] == [(2, 5), (4, 3), (8, 2)]
    assert blocks[0][1][0] == ("res", "_first", 128, 128) and blocks[1][1][0] == ("res", "_first", 128, 64)
    assert blocks[2][1] == (("res", "_first", 64, 32), ("res", "_second", 32, 32))
    stem, blocks = arch.growing_gen_table(192, 192, 3, False, True)          # net 2
    assert stem == (("res", "_1", 16, 12), ("res", "_2", 24, 48))
    assert [us for _, us in blocks] == [(("res", "_first", 96, 96), ("res", "_second", 48, 48)),
                                        (("res", "_first", 48, 48), ("res", "_second", 24, 24)),
                                        (("res", "_first", 24, 24), ("res", "_second", 12, 12))]
    stem, blocks = arch.growing_gen_table(192, 96, 3, False, False)          # net 3
    assert stem == (("pair", "1", 32, 96),)
    assert [us for _, us in blocks] == [(("pair", "2", 96, 96),), (("pair", "4", 48, 48),), (("pair", "8", 24, 24),)]
    first, rows = arch.growing_disc_table(256, 256, 8, 3, True, 3)           # net-1 critic: 2->32 | 32->64->64 | 64->128->128 | 128->384->128
    assert first == 32 and rows == ((8, (4, 4), 32, 64, 64), (4, (4, 4), 64, 128, 128), (2, (4, 4), 128, 384, 128))
    first, rows = arch.growing_disc_table(192, 192, 8, 3, False, 5)
    assert first == 24 and [r[1:] for r in rows] == [((5, 5), 24, 24, 48), ((5, 5), 48, 48, 96), ((5, 5), 96, 96, 96)]


def test_no_packed_fp32_valu(mpg):
    """The device code holds no packed-fp32 VALU instruction: one with op_sel bit 1 set (low half = HIGH word of source 1)
    next to MFMA waves of another stream was measured to return wrong sums in lanes 48-63 on MI355X
    (profiles/r03/packed_fp32_followup.md), so the library is built with -fno-slp-vectorize -fno-vectorize.  Disassembles
    the gfx950 code object of the built library."""
    import os
    import re
    import subprocess
    import tempfile
    from mpgan_amd import _lib
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    bundler, objdump = os.path.join(llvm, "clang-offload-bundler"), os.path.join(llvm, "llvm-objdump")
    asm = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", _lib.LIB_PATH, fat],
                       check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        assert starts, "no offload bundle in the library"
        for k, a in enumerate(starts):                          # one bundle per translation unit
            part = os.path.join(tmp, "part%d.bin" % k)
            with open(part, "wb") as f:
                f.write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, "dev%d.co" % k)
            subprocess.run([bundler, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + part, "--output=" + co], check=True)
            asm.append(subprocess.run([objdump, "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout)
    asm = "\n".join(asm)
    assert "v_mfma_f32_32x32x16_f16" in asm                     # it is the library's device code
    assert "v_mfma_scale_f32_32x32x64_f8f6f4" in asm and "v_cvt_scalef32_pk32_bf6_f16" in asm    # the F16F6 correction path
    packed = re.findall(r"v_pk_(?:fma|mul|add)_f32", asm)
    assert not packed, "%d packed-fp32 VALU instructions in libmpgan_hip.so" % len(packed)
    # the matrix-core convolution kernels neither spill nor use scratch: a scratch_ instruction inside one of them means
    # the register budget of a K loop was exceeded (round 2 shipped conv_mfma_f8_kernel<2> with 8 spilled registers)
    body, spills = None, {}
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            body = m.group(1)
        elif body and ("conv_mfma" in body or "wgrad_mfma" in body) and "scratch_" in line:
            # (in the weight-gradient kernel a scratch load in front of an LDS-DMA instruction waits for every copy in
            # flight: a two-entry pointer table indexed at run time cost 50 % there, profiles/r03/wgrad_variants.md)
            spills[body] = spills.get(body, 0) + 1
    assert not spills, "scratch traffic in the matrix-core kernels: %s" % spills
    assert "ds_read_b64_tr_b16" in asm                          # the weight gradient reads G8 rows through the transposing LDS read
    assert len(set(re.findall(r"<(\S*conv_mfma_f6_kernel\S*)>:", asm))) == 4
