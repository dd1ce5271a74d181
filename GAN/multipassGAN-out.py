#!/usr/bin/env python3
"""Multi-network 8x inference driver: same ``name value`` command line, input files and
output files as the reference's GAN/multipassGAN-out.py (params :28-99, loader :133-138,
generate3DUniForNewNetwork :390-618, frame loop :629-632), executed on one MI355X through the
HIP path (volumes stay in HBM between the passes).

Differences, all deliberate: weights come from ``model_%04d.ckpt.npz`` (see checkpoint.py);
``synthWeights 1`` (extra, optional) falls back to seeded random weights when a checkpoint is
missing; PNG previews are skipped (scipy.misc.imsave no longer exists); three loaded networks
run as three passes (the reference exits on an inverted check, :624-626); ``prec`` selects the
arithmetic (2 = MPG_PREC_F16F6, default; 3 = MPG_PREC_F16X3).  ``transposeAxis`` 0..3 pick the slicing
axes of the passes exactly as :397-547 do, including what is broken there: the third pass of
transposeAxis 2 raises the reference's IndexError (:542); ``add_adj_idcs2/3`` are ignored (the
reference's code for them writes past the array it allocates, :488-500).
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import mpgan_amd  # noqa: E402,F401
from mpgan_amd import checkpoint, multipass, uniio  # noqa: E402
from mpgan_amd import fluiddataloader as FDL  # noqa: E402
from mpgan_amd import paramhelpers as ph  # noqa: E402

outputOnly = int(ph.getParam("out", False)) > 0
basePath = ph.getParam("basePath", '../2ddata_gan/')
randSeed = int(ph.getParam("randSeed", 1))
simSizeLow = int(ph.getParam("simSize", 64))
tileSizeLow = int(ph.getParam("tileSize", 16))
upRes = int(ph.getParam("upRes", 4))
packedSimPath = ph.getParam("packedSimPath", '/data/share/GANdata/2ddata_sim/')
fromSim = int(ph.getParam("fromSim", 1000))
frame_min = int(ph.getParam("frame_min", 0))
frame_max = int(ph.getParam("frame_max", 200))
genModel = ph.getParam("genModel", 'gen_test')
discModel = ph.getParam("discModel", 'disc_test')
batch_norm = int(ph.getParam("batchNorm", False)) > 0
pixel_norm = int(ph.getParam("pixelNorm", True)) > 0
useVelocities = int(ph.getParam("useVelocities", 0))
useVorticities = int(ph.getParam("useVorticities", 0))
useFlags = int(ph.getParam("useFlags", 0))
useK_Eps_Turb = int(ph.getParam("useK_Eps_Turb", 0))
velScale = float(ph.getParam("velScale", 1.0))
generateUni = int(ph.getParam("genUni", False))
upsampleMode = int(ph.getParam("upsampleMode", 1))
usePixelShuffle = int(ph.getParam("usePixelShuffle", 0))
addBicubicUpsample = int(ph.getParam("addBicubicUpsample", 0))
load_emas = int(ph.getParam("loadEmas", 0))
firstNNArch = int(ph.getParam("firstNNArch", True))
transposeAxis = int(ph.getParam("transposeAxis", 0))
gpu = ph.getParam("gpu", "0")
synthWeights = int(ph.getParam("synthWeights", 0))          # extension: seeded random weights if no checkpoint
prec = ph.getParam("prec", "2")                             # extension: 2 = f16f6 (default), 3 = f16x3 (fp32-grade)
nets = []
for k in (1, 2, 3):
    nets.append(dict(
        load_model_test=int(ph.getParam("load_model_test_%d" % k, -1)),
        load_model_no=int(ph.getParam("load_model_no_%d" % k, -1)),
        use_res_net=int(ph.getParam("use_res_net%d" % k, True)) > 0,
        add_adj=int(ph.getParam("add_adj_idcs%d" % k, False)) > 0,
        start_fms=int(ph.getParam("startFms%d" % k, 512)),
        max_fms=int(ph.getParam("maxFms%d" % k, 256)),
        filter_size=int(ph.getParam("filterSize%d" % k, 3))))
ph.checkUnusedParams()
from mpgan_amd import ops as _ops  # noqa: E402
prec = _ops.parse_prec(prec)

if useVorticities or useFlags or useK_Eps_Turb or usePixelShuffle:
    print("ERROR: vorticity / flag / turbulence channels and pixel shuffle are not part of the multi-pass hot path")
    exit(1)
if transposeAxis not in (0, 1, 2, 3):
    print("ERROR: transposeAxis %d (0..3)" % transposeAxis)
    exit(1)
os.environ.setdefault("HIP_VISIBLE_DEVICES", str(gpu))
device = "cuda:0"
simSizeHigh = simSizeLow * upRes
n_ch = 4 if useVelocities else 1

mfl = ["density"] + (["velocity"] if useVelocities else [])
floader = FDL.FluidDataLoader(print_info=3, base_path=packedSimPath, base_path_y=packedSimPath, numpy_seed=randSeed,
                              filename="density_low_%04d.uni", filename_index_min=frame_min, oldNamingScheme=False,
                              filename_y=None, filename_index_max=frame_max, indices=[fromSim], data_fraction=1.0,
                              multi_file_list=mfl, multi_file_list_y=["density"])
x_3d, _, _ = floader.get()
x_3d[:, :, :, :, 1:4] = velScale * x_3d[:, :, :, :, 1:4]       # out.py:138

gens = []
test_path = None
for k, n in enumerate(nets):
    if n["load_model_test"] == -1:
        continue
    if not os.path.exists(basePath + 'test_%04d/' % n["load_model_test"]):
        print('ERROR: Test to load does not exist.')
    path = checkpoint.model_path(basePath, n["load_model_test"], n["load_model_no"], ema=bool(load_emas))
    prefix = 'out_%04d-%04d' % (n["load_model_test"], n["load_model_no"])
    if os.path.isdir(basePath + 'test_%04d/' % n["load_model_test"]):
        test_path, _ = ph.getNextGenericPath(prefix, 0, basePath + 'test_%04d/' % n["load_model_test"])
    cfg = dict(tile_low=simSizeLow, up_res=upRes, channels=n_ch, first_gen=(k == 0), filter_size=n["filter_size"],
               start_fms=n["start_fms"], max_fms=n["max_fms"], add_adj=n["add_adj"] and k == 0,
               first_nn_arch=bool(firstNNArch) and k == 0, use_res_net=n["use_res_net"], pixel_norm=pixel_norm,
               batch_norm=batch_norm, upsample_mode=upsampleMode, add_bicubic=bool(addBicubicUpsample))
    try:
        params = checkpoint.load(path)
        print("Model %d restored from %s." % (k + 1, path))
    except FileNotFoundError as e:
        if not synthWeights:
            print("ERROR: %s" % e)
            exit(1)
        params = None
        print("Model %d: no checkpoint, seeded synthetic weights (synthWeights 1)" % (k + 1))
    gens.append(multipass.Generator("growing_gen", cfg, params, prec=prec, device=device, seed=randSeed + k))
if not gens:
    print("At least one network has to be loaded.")
    exit(1)

print('*****OUTPUT ONLY*****')
head_0, _ = uniio.readUni(packedSimPath + "sim_%04d/density_low_%04d.uni" % (fromSim, 0)) \
    if os.path.exists(packedSimPath + "sim_%04d/density_low_%04d.uni" % (fromSim, 0)) \
    else uniio.readUni(packedSimPath + "sim_%04d/density_low_%04d.uni" % (fromSim, frame_min))
for layerno in range(frame_min, frame_max):
    print(layerno)
    start = time.time()
    low = torch.as_tensor(np.ascontiguousarray(x_3d[layerno - frame_min])).to(device)
    vol = multipass.multipass_8x(gens, low, upRes, apply_cutoff=bool(generateUni), transpose_axis=transposeAxis)
    torch.cuda.synchronize()
    print("time for %d network(s): %.6f" % (len(gens), time.time() - start))
    if generateUni:
        head = dict(head_0)
        head['dimX'] = head['dimY'] = head['dimZ'] = simSizeHigh
        uniio.writeUniFromDevice(packedSimPath + '/sim_%04d/source_%04d.uni' % (fromSim, layerno), head, vol)
        print('stored .uni file')
print('Test finished, %d volumes written to %s.' % (frame_max - frame_min, packedSimPath))
