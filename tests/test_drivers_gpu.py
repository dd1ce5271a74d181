"""End-to-end drop-in check of the driver scripts: same command lines as
example_run_output.py (reduced sizes), .uni files in, .uni files out, compared with the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import rel_l2
from oracle import multipass as OM
from oracle import nets as ON

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, args, cwd):
    cmd = [sys.executable, os.path.join(ROOT, "GAN", script)] + [str(a) for a in args]
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def _make_sim(tmp, sim, frames, vel):
    import mpgan_amd  # noqa: F401
    from mpgan_amd import uniio
    from mpgan_amd.synthetic import synthetic_volume
    d = tmp / "data" / "sim_1005"
    d.mkdir(parents=True)
    vols = []
    for f in range(frames):
        v = synthetic_volume(sim, 4, f)
        vols.append(v)
        uniio.writeUni(str(d / ("density_low_%04d.uni" % f)), uniio.make_header(sim, sim, sim), v[..., 0:1])
        if vel:
            uniio.writeUni(str(d / ("velocity_low_%04d.uni" % f)), uniio.make_header(sim, sim, sim, vec3=True), v[..., 1:4])
    (tmp / "models" / "test_0004").mkdir(parents=True)
    (tmp / "models" / "test_0048").mkdir(parents=True)
    return vols


@pytest.mark.parametrize("vel", [0, 1])
def test_4x_two_invocations(tmp_path, vel):
    from mpgan_amd import uniio
    sim, up = 8, 4
    vols = _make_sim(tmp_path, sim, 2, vel)
    common = ["upRes", up, "out", 1, "tileSize", sim, "simSize", sim, "fromSim", 1005, "toSim", 1005, "dataDim", 2,
              "useVelocities", vel, "basePath", str(tmp_path / "models") + "/", "packedSimPath", str(tmp_path / "data") + "/",
              "frame_min", 0, "frame_max", 2, "genUni", 1, "velScale", 1.0, "synthWeights", 1, "genModel", "gen_resnet"]
    _run("multipassGAN-4x.py", common + ["randSeed", 101, "load_model_test", 4, "load_model_no", 1199,
                                         "upsamplingMode", 2, "upsampledData", 0], str(tmp_path))
    _run("multipassGAN-4x.py", common + ["randSeed", 102, "load_model_test", 48, "load_model_no", 799,
                                         "upsamplingMode", 1, "upsampledData", 1], str(tmp_path))
    for f in range(2):
        low = vols[f] if vel else vols[f][..., 0:1]
        ref, ref1 = OM.two_pass_4x(ON.ParamSource(seed=101), ON.ParamSource(seed=102), low, up, True)
        h1, v1 = uniio.readUni(str(tmp_path / "data" / "sim_1005" / ("density_low_2x2_%04d.uni" % f)))
        h2, v2 = uniio.readUni(str(tmp_path / "data" / "sim_1005" / ("density_low_1x1_%04d.uni" % f)))
        assert (h2["dimX"], h2["dimY"], h2["dimZ"]) == (32, 32, 32) and v2.shape == (32, 32, 32, 1)
        assert rel_l2(v1[..., 0], ref1) < 1e-4
        assert rel_l2(v2[..., 0], ref) < 1e-4


def test_8x_out_driver(tmp_path):
    from mpgan_amd import uniio
    sim, up = 4, 8
    vols = _make_sim(tmp_path, sim, 1, 1)
    (tmp_path / "models" / "test_0000").mkdir()
    args = ["randSeed", 200, "upRes", up, "pixelNorm", 1, "batchNorm", 0, "out", 1, "tileSize", sim, "simSize", sim,
            "fromSim", 1005, "useVelocities", 1, "useVorticities", 0, "useK_Eps_Turb", 0, "useFlags", 0,
            "genModel", "gen_resnet", "discModel", "disc_binclass", "basePath", str(tmp_path / "models") + "/",
            "packedSimPath", str(tmp_path / "data") + "/", "frame_max", 1, "frame_min", 0, "velScale", 1.0, "genUni", 1,
            "upsampleMode", 1, "usePixelShuffle", 0, "loadEmas", 0, "addBicubicUpsample", 1, "gpu", 0, "transposeAxis", 0,
            "firstNNArch", 1, "load_model_test_1", 0, "load_model_no_1", 299, "use_res_net1", 1, "add_adj_idcs1", 1,
            "startFms1", 256, "maxFms1", 256, "filterSize1", 3, "load_model_test_2", 4, "load_model_no_2", 585,
            "use_res_net2", 1, "add_adj_idcs2", 0, "startFms2", 192, "maxFms2", 192, "filterSize2", 5,
            "load_model_test_3", -1, "load_model_no_3", -1, "use_res_net3", 0, "add_adj_idcs3", 0, "startFms3", 192,
            "maxFms3", 96, "filterSize3", 5, "synthWeights", 1]
    out = _run("multipassGAN-out.py", args, str(tmp_path))
    assert "stored .uni file" in out
    cfgs = [dict(filter_size=3, start_fms=256, max_fms=256, add_adj=True, first_nn_arch=True, use_res_net=True),
            dict(filter_size=5, start_fms=192, max_fms=192, use_res_net=True)]
    ref = OM.multipass_8x([ON.ParamSource(seed=200), ON.ParamSource(seed=201)], cfgs, vols[0], up)
    h, v = uniio.readUni(str(tmp_path / "data" / "sim_1005" / "source_0000.uni"))
    assert v.shape == (32, 32, 32, 1) and h["dimX"] == 32
    assert rel_l2(v[..., 0], ref) < 1e-4
    # unknown parameters abort like the reference (paramhelpers.py:29-37)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "GAN", "multipassGAN-out.py"), "nonsense", "1"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "not used" in r.stdout
