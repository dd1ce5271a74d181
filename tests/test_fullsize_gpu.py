"""Parity of the benched arithmetic AT the benched sizes (BASELINE.json configs[1] "C2" and the slice shapes
of configs[3] "C4"), against the oracle evaluated on the host cores.

C2: one whole 64^3 -> 256^3 volume through both passes (multipassGAN-4x.py:1090-1169).  The oracle
(oracle/torch_ref.py, PyTorch-CPU fp32 twin of the numpy restatement) evaluates ALL 256 slices of pass 1 --
pass 2 needs whole (z,y) planes of the pass-1 volume -- and 24 slices of pass 2 (first, middle and last 8 of
the x axis).  MPG_PREC_F16F8 (bench / driver default) is held to 5e-4, MPG_PREC_F16X3 to 1e-4; north_star
asks 1e-3 relative L2 on density fields.

C4: one batch of 8 slices at 512^2 through each of the three growing generators with the widths of
example_run_output.py:18-47 (NT = 2 / 3 / 4 cout-tile F16F8 kernels); the oracle evaluates slices 0, 3, 7.
"""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import multipass as OM
from oracle import nets as ON
from oracle import ops as O
from oracle import torch_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {3: 1e-4, 2: 5e-4}
SIM, UP = 64, 4
S = SIM * UP
IDX = list(range(8)) + list(range(S // 2 - 4, S // 2 + 4)) + list(range(S - 8, S))


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=DEV)


@pytest.fixture(scope="module")
def MP(mpg):
    from mpgan_amd import multipass
    return multipass


@pytest.fixture(scope="module")
def c2_oracle(MP):
    """oracle pass 1 of the whole volume, computed once for both precisions"""
    from mpgan_amd.synthetic import synthetic_volume
    torch.set_num_threads(16)
    low = synthetic_volume(SIM, 1, 0)
    cfg1 = dict(tile_low=SIM, up_res=UP, channels=1, upsampling_mode=2, batch_norm=True)
    cfg2 = dict(tile_low=SIM, up_res=UP, channels=1, upsampling_mode=1, batch_norm=True)
    g1 = MP.Generator("gen_resnet", cfg1, None, 3, device=DEV, seed=777)
    g2 = MP.Generator("gen_resnet", cfg2, None, 3, device=DEV, seed=778)
    p1, p2 = g1.params(), g2.params()
    xs = O.zoom_axis_linear(low, 0, UP)
    r1 = np.concatenate([torch_ref.gen_resnet(p1, xs[i:i + 16], UP, 2, True) for i in range(0, S, 16)], axis=0)
    r1 = OM.cutoff(r1.reshape(S, S, S))                                     # [z,y,x], as written to density_low_2x2
    x2 = OM.pass2_input_4x(r1, low, UP)[IDX]                                # [x][z][y] planes of the ORACLE's pass 1
    r2 = torch_ref.gen_resnet(p2, x2, UP, 1, True)[..., 0]
    final = OM.cutoff(r2)                                                   # final[z,y,x] = r2[x][z][y]
    return dict(low=low, cfg1=cfg1, cfg2=cfg2, p1=p1, p2=p2, v1=r1, final_planes=final)


@pytest.mark.parametrize("prec", [2, 3])
def test_c2_two_pass_full_size_vs_oracle(MP, c2_oracle, prec):
    o = c2_oracle
    g1 = MP.Generator("gen_resnet", o["cfg1"], o["p1"], prec, device=DEV)
    g2 = MP.Generator("gen_resnet", o["cfg2"], o["p2"], prec, device=DEV)
    final, v1 = MP.two_pass_4x(g1, g2, _t(o["low"]), UP, batch=8)
    v1 = v1.cpu().numpy()
    final = final.cpu().numpy()
    assert final.shape == (S, S, S)
    e1 = rel_l2(v1, o["v1"])
    got = final[:, :, IDX].transpose(2, 0, 1)
    e2 = rel_l2(got, o["final_planes"])
    per_slice = max(rel_l2(got[i], o["final_planes"][i]) for i in range(len(IDX)))
    print("C2 prec %d: pass 1 (256 slices) %.3e, end to end (24 slices of pass 2) %.3e, worst slice %.3e" % (prec, e1, e2, per_slice))
    assert e1 < TOL[prec], e1
    assert e2 < TOL[prec], e2
    assert per_slice < 2 * TOL[prec], per_slice


NET_CFGS = {
    # example_run_output.py:18-47
    "net1": dict(first_gen=True, filter_size=3, start_fms=256, max_fms=256, add_adj=True, first_nn_arch=True, use_res_net=True),
    "net2": dict(first_gen=False, filter_size=5, start_fms=192, max_fms=192, use_res_net=True),
    "net3": dict(first_gen=False, filter_size=5, start_fms=192, max_fms=96, use_res_net=False),
}


@pytest.mark.parametrize("prec", [2, 3])
@pytest.mark.parametrize("name", ["net1", "net2", "net3"])
def test_c4_slice_batch_512_vs_oracle(MP, name, prec):
    """8 slices of 512^2 (64^2 low-res, 4 channels + neighbours) through one growing generator"""
    from mpgan_amd.synthetic import synthetic_volume
    torch.set_num_threads(16)
    cfg = NET_CFGS[name]
    low, up, nch = 64, 8, 4
    hi = low * up
    vol = synthetic_volume(low, nch, 3)                     # [z,y,x,4]
    ps = ON.ParamSource(seed=51)
    chk = [0, 3, 7]
    if cfg["first_gen"]:
        x = OM.add_adjacent(vol[20:28], nch)                # [8,64,64,6]
        with torch_ref.fast_convs():
            ref = ON.growing_gen(ps, x[chk], up, True, cfg["filter_size"], cfg["start_fms"], cfg["max_fms"], True, True)[..., 0]
        y_in = None
    else:
        x = vol[20:28]
        # previous-pass density: a smooth non-negative field at the high resolution
        yp = O.resize_bicubic_tf1(np.maximum(vol[30:38, :, :, :1], 0), hi, hi).astype(np.float32)
        with torch_ref.fast_convs():
            ref = ON.growing_gen(ps, ON.gen2_input(yp[chk], x[chk], hi), up, False, cfg["filter_size"], cfg["start_fms"],
                                 cfg["max_fms"], False, cfg["use_res_net"])[..., 0]
        y_in = _t(yp[..., 0])
    gen = MP.Generator("growing_gen", dict(tile_low=low, up_res=up, channels=nch, **cfg), params=ps.params, prec=prec)
    y = gen(_t(x), y_in).cpu().numpy()
    assert y.shape == (8, hi, hi)
    err = rel_l2(y[chk], ref)
    print("C4 %s prec %d: %.3e" % (name, prec, err))
    assert err < TOL[prec], err
