#!/usr/bin/env python3
"""8x progressive-growing training driver: the ``name value`` command line of the reference's
GAN/multipassGAN-8x.py (params :30-158) as example_run_training.py issues it for the first network
(upsamplingMode 2, upsampledData 0: low-res slices in, 8x slices out, three growing stages) and for the second /
third network (upsamplingMode 1 / 3, upsampledData 1: slices along the next axis, the previous network's output
volumes ``density_low_t%04d_2x2_%04d.uni`` / ``..._1x1_...`` as an extra high-res input channel, :231-238,:361-366).

What is rebuilt (reference line numbers): the data path :196-345 (FluidDataLoader slices of three
coherent frames with the add_adj_idcs neighbour channels, intermediate-resolution targets
``density_low_%i_%04d.uni`` for the 2x / 4x stages, ``density_high_%04d.uni`` for 8x), the growing schedule
:1885-1975 (stageIter iterations of fade-in + stageIter of stabilisation per stage, data re-loaded when a
stage completes), the iteration :1990-2060 (spatial critic, temporal critic on advected triples, generator)
through ``train.Trainer8x``, the polynomial learning-rate decay :995-1009 after 6 * stageIter iterations,
checkpoints ``model_%04d.ckpt.npz`` and the moving-average weights ``model_ema_%04d.ckpt.npz`` :1804-1812.

``out 1`` is the per-network output mode (see output_main).

Not rebuilt: upsamplingMode 0 (the linear-interpolation variant no example run uses), vorticity / flag / k-eps
inputs, PNG test images, TensorBoard.  ``lossScaling 1`` runs the dynamic loss scaling of :490-541 and every
stage uses its own optimisers over its own variable subset (:1305-1362) -- train.StagedAdam.
"""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import mpgan_amd  # noqa: E402,F401
from mpgan_amd import checkpoint  # noqa: E402
from mpgan_amd import fluiddataloader as FDL  # noqa: E402
from mpgan_amd import paramhelpers as ph  # noqa: E402
from mpgan_amd import tilecreator_t as tc  # noqa: E402

P = {}
for name, default in [
        ("out", False), ("basePath", '../2ddata_gan/'), ("randSeed", 1), ("load_model_test", -1), ("load_model_no", -1),
        ("simSize", 64), ("tileSize", 16), ("upRes", 4), ("packedSimPath", '/data/share/GANdata/2ddata_sim/'),
        ("fromSim", 1000), ("toSim", -1), ("dataDim", 2), ("numOut", 200), ("saveOut", False), ("loadOut", -1),
        ("img", True), ("gif", False), ("ref", False), ("frame_min", 0), ("genModel", 'gen_test'),
        ("discModel", 'disc_test'), ("learningRate", 0.0002), ("decayLR", False), ("dropout", 1.0), ("dropoutOutput", 1.0),
        ("adam_beta1", 0.5), ("adam_beta2", 0.999), ("weight_dld", 1.0), ("lambda", 1.0), ("lambda2", 0.0),
        ("lambda_f", 1.0), ("lambda2_f", 1.0), ("lambda2_l1", 1.0), ("lambda2_l2", 1.0), ("lambda2_l3", 1.0),
        ("lambda2_l4", 1.0), ("useTempoD", True), ("useTempoL2", False), ("lambda_t", 1.0), ("lambda_t_l2", 0.0),
        ("batchSize", 128), ("batchSizeDisc", 128), ("batchSizeGen", 128), ("trainGAN", True),
        ("trainingIterations", 100000), ("discRuns", 1), ("genRuns", 1), ("batchNorm", False), ("pixelNorm", True),
        ("bnDecay", 0.999), ("useVelocities", 0), ("useVorticities", 0), ("useFlags", 0), ("useK_Eps_Turb", 0),
        ("premadeTiles", 0), ("dataAugmentation", 0), ("minScale", 0.85), ("maxScale", 1.15), ("rot", 2),
        ("transposeAxis", 0), ("minAngle", -90.0), ("maxAngle", 90.0), ("flip", 1), ("pretrain", 0), ("pretrainDisc", 0),
        ("pretrainGen", 0), ("testPathStartNo", 0), ("testInterval", 100), ("numTests", 128), ("outputInterval", 100),
        ("saveInterval", 200), ("alwaysSave", True), ("keepMax", 3), ("genTestImg", -1), ("note", ""),
        ("data_fraction", 0.3), ("frame_max", 200), ("adv_flag", True), ("adv_mode", 1), ("change_velocity", False),
        ("saveMetaData", 0), ("use_spatialdisc", True), ("velScale", 1.0), ("upsamplingMode", 2), ("upsampledData", False),
        ("genUni", False), ("upsampleFirst", True), ("usePixelShuffle", False), ("addBicubicUpsample", False),
        ("startingIter", 0), ("loadEmas", False), ("useVelInTDisc", False), ("upsampleMode", 1), ("lossScaling", False),
        ("stageIter", 25000), ("decayIter", 25000), ("maxFms", 256), ("use_wgan_gp", False), ("use_res_net", False),
        ("use_mb_stddev", False), ("deviceTiles", 1), ("use_LSGAN", False), ("startFms", 512), ("filterSize", 3), ("outNNTestNo", 17),
        ("firstNNArch", False), ("gDrop", False), ("add_adj_idcs", False), ("gpu", 2), ("synthWeights", 0), ("prec", "2"), ("trainPrec", "3")]:
    P[name] = ph.getParam(name, default)
ph.checkUnusedParams()


def fail(msg):
    print("ERROR: " + msg)
    exit(1)


def output_main():
    """``out 1`` (generate3DUniForNewNetwork :1600-1780, output loop :2316-2324): ONE network per invocation, the
    volumes travel through .uni files as in the commented alternative of example_run_output.py:64-70:
      upsamplingMode 2, upsampledData 0 : first network          -> density_low_t%04d_2x2_%04d.uni   (t = load_model_test)
      upsamplingMode 1, upsampledData 1 : density_low_t<outNNTestNo>_2x2 in -> density_low_t%04d_1x1_%04d.uni
      upsamplingMode 3, upsampledData 1 : density_low_t<outNNTestNo>_1x1 in -> density_low_0x0_%04d.uni
    with the slicing axis given by transposeAxis (0..3) and the <5e-4 cutoff on every written volume."""
    from mpgan_amd import multipass, ops, uniio
    mode, upsampled = int(P["upsamplingMode"]), int(P["upsampledData"])
    if (mode, upsampled) not in ((2, 0), (1, 1), (3, 1)) or int(P["dataDim"]) != 2 or not int(P["upsampleFirst"]):
        fail("output mode: upsamplingMode 2 (upsampledData 0) or 1 / 3 (upsampledData 1), dataDim 2, upsampleFirst 1")
    if int(P["useVorticities"]) or int(P["useFlags"]) or int(P["useK_Eps_Turb"]) or int(P["usePixelShuffle"]):
        fail("vorticity / flag / k-eps inputs and pixel shuffle are not supported")
    ta = int(P["transposeAxis"])
    if ta not in (0, 1, 2, 3):
        fail("transposeAxis %d (0..3)" % ta)
    up, sim = int(P["upRes"]), int(P["simSize"])
    s = sim * up
    base, sims = P["basePath"], P["packedSimPath"]
    from_sim, f0, f1 = int(P["fromSim"]), int(P["frame_min"]), int(P["frame_max"])
    vel = int(P["useVelocities"]) > 0
    n_ch = 4 if vel else 1
    first = mode == 2
    test_no, model_no = int(P["load_model_test"]), int(P["load_model_no"])
    mfl = ["density"] + (["velocity"] if vel else [])
    fl = FDL.FluidDataLoader(print_info=3, base_path=sims, base_path_y=sims, numpy_seed=int(P["randSeed"]),
                             filename="density_low_%04d.uni", filename_index_min=f0, oldNamingScheme=False, filename_y=None,
                             filename_index_max=f1, indices=[from_sim], data_fraction=1.0, multi_file_list=mfl,
                             multi_file_list_y=["density"])
    x_3d, _, _ = fl.get()
    x_3d[:, :, :, :, 1:4] = float(P["velScale"]) * x_3d[:, :, :, :, 1:4]                     # :361
    x_2 = None
    if upsampled:
        prev_name = ("density_low_t%04d_2x2" if mode == 1 else "density_low_t%04d_1x1") % int(P["outNNTestNo"]) + "_%04d.uni"
        fl2 = FDL.FluidDataLoader(print_info=0, base_path=sims, numpy_seed=int(P["randSeed"]), filename=prev_name,
                                  filename_index_min=f0, oldNamingScheme=False, filename_index_max=f1, indices=[from_sim],
                                  data_fraction=1.0, multi_file_list=["density"])
        x_2, _, _ = fl2.get()
    cfg = dict(tile_low=sim, up_res=up, channels=n_ch, first_gen=first, filter_size=int(P["filterSize"]),
               start_fms=int(P["startFms"]), max_fms=int(P["maxFms"]), add_adj=int(P["add_adj_idcs"]) > 0 and first,
               first_nn_arch=int(P["firstNNArch"]) > 0 and first, use_res_net=int(P["use_res_net"]) > 0,
               pixel_norm=int(P["pixelNorm"]) > 0, batch_norm=int(P["batchNorm"]) > 0, upsample_mode=int(P["upsampleMode"]),
               add_bicubic=int(P["addBicubicUpsample"]) > 0)
    path = checkpoint.model_path(base, test_no, model_no, ema=int(P["loadEmas"]) > 0)
    try:
        params = checkpoint.load(path)
        print("Model restored from %s." % path)
    except FileNotFoundError as e:
        if not int(P["synthWeights"]):
            fail(str(e))
        params = None
        print("no checkpoint, seeded synthetic weights (synthWeights 1)")
    gen = multipass.Generator("growing_gen", cfg, params, prec=ops.parse_prec(P["prec"]), device="cuda:0",
                              seed=int(P["randSeed"]))
    out_name = {2: "density_low_t%04d_2x2" % test_no + "_%04d.uni", 1: "density_low_t%04d_1x1" % test_no + "_%04d.uni",
                3: "density_low_0x0_%04d.uni"}[mode]
    print('*****OUTPUT ONLY*****')
    probe = sims + "sim_%04d/density_low_%04d.uni" % (from_sim, 0)
    head_0, _ = uniio.readUni(probe if os.path.exists(probe) else sims + "sim_%04d/density_low_%04d.uni" % (from_sim, f0))
    gen_uni = int(P["genUni"]) > 0
    for layerno in range(f0, f1):
        print(layerno)
        t0 = time.time()
        low = torch.as_tensor(np.ascontiguousarray(x_3d[layerno - f0])).to("cuda:0")
        prev = None if x_2 is None else torch.as_tensor(np.ascontiguousarray(x_2[layerno - f0][..., 0])).to("cuda:0")
        vol = multipass.single_pass_8x(gen, low, prev, up, ta, batch=2, apply_cutoff=gen_uni)
        torch.cuda.synchronize()
        print("time for network: {0:.6f}".format(time.time() - t0))
        if gen_uni:
            head = dict(head_0)
            head['dimX'] = head['dimY'] = head['dimZ'] = s
            uniio.writeUniFromDevice(sims + '/sim_%04d/' % from_sim + out_name % layerno, head, vol)
    print('Test finished, %d volumes written to %s.' % (f1 - f0, sims))


if int(P["out"]) > 0:
    output_main()
    exit(0)
upsampling_mode, upsampled_data = int(P["upsamplingMode"]), int(P["upsampledData"])
if (upsampling_mode, upsampled_data) not in ((2, 0), (1, 1), (3, 1)) or int(P["dataDim"]) != 2:
    fail("training is implemented for the first network (upsamplingMode 2, upsampledData 0) and the second / third "
         "one (upsamplingMode 1 / 3, upsampledData 1), dataDim 2")
later_net = upsampling_mode != 2
if int(P["useVorticities"]) or int(P["useFlags"]) or int(P["useK_Eps_Turb"]) or int(P["premadeTiles"]):
    fail("vorticity / flag / k-eps inputs and premade tiles are not supported")
if int(P["usePixelShuffle"]) or int(P["gDrop"]) or int(P["useVelInTDisc"]):
    fail("usePixelShuffle / gDrop / useVelInTDisc are 0 in the reference runs and not built")
if (int(P["batchNorm"]) or int(P["use_mb_stddev"])) and int(P["use_wgan_gp"]):
    fail("batchNorm / use_mb_stddev (8x.py:85,149) train with use_wgan_gp 0 (LSGAN or sigmoid cross entropy): the gradient "
         "penalty would need second derivatives of the batch statistics, which are not built")
upRes = int(P["upRes"])
if upRes != 8:
    fail("the growing networks are built for upRes 8")
kt, kt_l = float(P["lambda_t"]), float(P["lambda_t_l2"])
useTempoD = kt > 1e-6                                                # 8x.py:191-199
if kt_l > 1e-6:
    fail("the l2 temporal loss (lambda_t_l2) is not built; use lambda_t")
if useTempoD and int(P["adv_flag"]) and int(P["adv_mode"]) and not int(P["useVelocities"]):
    fail("adv_mode 1 / 2 (GAN.advect) advects with the velocity channels of the low-res tiles: useVelocities 1")

basePath, packedSimPath = P["basePath"], P["packedSimPath"]
simSizeLow, tileSizeLow = int(P["simSize"]), int(P["tileSize"])
fromSim, toSim = int(P["fromSim"]), int(P["toSim"])
toSim = fromSim if toSim == -1 else toSim
frame_min, frame_max = int(P["frame_min"]), int(P["frame_max"])
randSeed = int(P["randSeed"])
useVelocities, add_adj_idcs = int(P["useVelocities"]) > 0, int(P["add_adj_idcs"]) > 0
stageIter, decayIter, startingIter = int(P["stageIter"]), int(P["decayIter"]), int(P["startingIter"])
trainingIterations = stageIter * 6 + decayIter                      # :160-161
data_fraction, min_data_fraction = float(P["data_fraction"]), 0.08
batch = int(P["batchSize"])
aug = int(P["dataAugmentation"]) > 0
device = "cuda:0"
device_tiles = int(P["deviceTiles"]) > 0
if device_tiles:
    from mpgan_amd.tiles_device import DeviceTileCreator  # noqa: E402

channelLayout_low, channelLayout_high, mfl, mfh = 'd', ('d,d' if later_net else 'd'), ["density"], ["density"]
if useVelocities:
    channelLayout_low += ',vx,vy,vz'
    mfl = mfl + ["velocity"]
if add_adj_idcs:
    channelLayout_low += ',d,d'
n_inputChannels = len(channelLayout_low.split(','))
if later_net and int(P["firstNNArch"]):
    fail("firstNNArch belongs to the first network")
# previous network's output volumes (:231-238) and the slicing axis of this pass (:301-307)
outNNTestNo = int(P["outNNTestNo"])
lowfilename_2 = {1: "density_low_t%04d_2x2" % outNNTestNo + "_%04d.uni", 3: "density_low_t%04d_1x1" % outNNTestNo + "_%04d.uni"}.get(upsampling_mode)
transpose_axis = {2: 0, 1: 2, 3: 1}[upsampling_mode]
dirIDs = np.linspace(fromSim, toSim, (toSim - fromSim + 1), dtype='int16')
mol = [o for o in range(3) for _ in mfl]
moh = [o for o in range(3) for _ in mfh]
stride = 3


def load_stage(currentUpres, first):
    """TileCreator + data of one growing stage (:290-345 at start-up, :1920-1960 at a stage change)"""
    tkw = dict(tileSizeLow=tileSizeLow, densityMinimum=0.002 if first else 0.01, channelLayout_high=channelLayout_high,
               simSizeLow=simSizeLow, dim=2, dim_t=3, channelLayout_low=channelLayout_low, upres=currentUpres, premadeTiles=False)
    # deviceTiles 1 (default): frames resident in HBM, batches cut / augmented by HIP kernels with the reference's random
    # decisions (tiles_device.DeviceTileCreator; tilecreator_t.py:457-546,1345-1414); 0: the host TileCreator
    tiCr = DeviceTileCreator(device=device, **tkw) if device_tiles else tc.TileCreator(**tkw)
    high = "density_high_%04d.uni" if currentUpres == upRes else "density_low_%i" % currentUpres + "_%04d.uni"
    off = 0 if first else stride * (int(round(math.log(currentUpres, 2))) - 1)
    common = dict(print_info=0, base_path=packedSimPath, base_path_y=packedSimPath, numpy_seed=randSeed,
                  add_adj_idcs=add_adj_idcs, conv_slices=True, conv_axis=transpose_axis,
                  select_random=(0.2 if later_net else 0.4) if first else 1.0, density_threshold=0.005 if first else 0.002,
                  axis_scaling_y=[1, 1, 1, 1], axis_scaling=[currentUpres, 1, 1, 1], filename="density_low_%04d.uni",
                  oldNamingScheme=False, filename_index_max=frame_max + off, filename_index_min=frame_min + off,
                  indices=dirIDs, multi_file_list_y=mfh * 3, multi_file_idxOff_y=moh)
    first_fraction = max(data_fraction * 2 / currentUpres, min_data_fraction)
    x_2 = None
    if later_net:      # the same slices of the previous network's output, read as the `y` of a density-only loader (:324-333)
        fl2 = FDL.FluidDataLoader(filename_y=lowfilename_2, data_fraction=first_fraction, multi_file_list=["density"] * 3,
                                  multi_file_idxOff=[0, 1, 2], **common)
    fl = FDL.FluidDataLoader(filename_y=high, data_fraction=first_fraction if first else data_fraction,
                             multi_file_list=mfl * 3, multi_file_idxOff=mol, **common)
    if aug:
        tiCr.initDataAugmentation(rot=int(P["rot"]), minScale=float(P["minScale"]), maxScale=float(P["maxScale"]),
                                  flip=int(P["flip"]))
    x, y, _ = fl.get()
    if later_net:
        _, x_2, _ = fl2.get()
    simSizeHigh = simSizeLow * upRes
    if not later_net:
        x = x.reshape(-1, 1, simSizeLow, simSizeLow, n_inputChannels * 3)
        y = y.reshape(-1, 1, simSizeLow * currentUpres, simSizeLow * currentUpres, 3)
    else:
        # (:372-376) per frame the pair (target, previous pass): the reference's reshape / concatenate / reshape of
        # the three-frame arrays is this interleave of their last axes
        x = x.reshape(-1, 1, simSizeLow, simSizeLow, n_inputChannels * 3)
        y = y.reshape(-1, 1, simSizeHigh, simSizeHigh, 3)
        x_2 = x_2.reshape(-1, 1, simSizeHigh, simSizeHigh, 3)
        y = np.stack((y, x_2), axis=-1).reshape(-1, 1, simSizeHigh, simSizeHigh, 6)
    tiCr.addData(x, y)
    return tiCr


currentUpres = 8 if later_net else min(2 ** (startingIter // (stageIter * 2) + 1), 8)   # :213-217
tiCr = load_stage(currentUpres, True)
print("Random seed: {}".format(randSeed))
np.random.seed(randSeed)
test_path, _ = ph.getNextTestPath(int(P["testPathStartNo"]), basePath)
print("\nUsing parameters:\n" + ph.paramsToString())
ph.writeParams(test_path + "params.json")

from mpgan_amd import ops  # noqa: E402
from mpgan_amd.arch import Cfg8x  # noqa: E402
from mpgan_amd.train import Trainer8x  # noqa: E402

cfg = Cfg8x(tileSizeLow=tileSizeLow, upRes=upRes, n_inputChannels=n_inputChannels, upsampling_mode=upsampling_mode,
            upsampleMode=int(P["upsampleMode"]), filterSize=int(P["filterSize"]), start_fms=int(P["startFms"]),
            max_fms=int(P["maxFms"]), first_nn_arch=int(P["firstNNArch"]) > 0, use_res_net=int(P["use_res_net"]) > 0,
            pixel_norm=int(P["pixelNorm"]) > 0, addBicubicUpsample=int(P["addBicubicUpsample"]) > 0,
            use_mb_stddev=int(P["use_mb_stddev"]) > 0, bn_decay=float(P["bnDecay"]))
learning_rate = float(P["learningRate"])
trainer = Trainer8x(cfg, device=device, learning_rate=learning_rate, beta1=float(P["adam_beta1"]),
                    beta2=float(P["adam_beta2"]), lambda_l1=float(P["lambda"]), lambda2=float(P["lambda2"]),
                    weight_dld=float(P["weight_dld"]), use_wgan_gp=int(P["use_wgan_gp"]) > 0,
                    use_LSGAN=int(P["use_LSGAN"]) > 0, seed=randSeed, use_tempo=useTempoD, lambda_t=kt,
                    adv_flag=int(P["adv_flag"]) > 0, loss_scaling=int(P["lossScaling"]) > 0,
                    adv_mode=int(P["adv_mode"]), batch_norm=int(P["batchNorm"]) > 0, prec=ops.parse_prec(P["trainPrec"]))
if int(P["load_model_test"]) >= 0:
    params = checkpoint.load(checkpoint.model_path(basePath, int(P["load_model_test"]), int(P["load_model_no"])))
    with torch.no_grad():
        for n_, t_ in trainer.sess.params.items():
            if n_ in params:
                t_.copy_(torch.as_tensor(params[n_], device=t_.device))
    n_slots = trainer.load_slot_state(params)                         # the Saver restores the optimiser slots too
    print("Model restored (%d optimiser slot pairs)." % n_slots)


def getinput():
    """:1497-1537 incl. the 1-in-20 empty-density batches"""
    if device_tiles:
        batch_xs, batch_ys = tiCr.selectRandomTilesDevice(batch, augment=aug)
    else:
        batch_xs, batch_ys = tiCr.selectRandomTiles(selectionSize=batch, augment=aug)
    if not min(np.random.randint(0, 20), 1):
        batch_xs[:, :, :, :, 0:1] = 0
        if add_adj_idcs:
            batch_xs[:, :, :, :, 4:6] = 0
        batch_xs[:, :, :, :, 1:4] *= (1.0 + np.random.rand() * 1.5)
        batch_ys[:, :, :, :, :] = 0
    return batch_xs.reshape(-1, cfg.n_input), batch_ys.reshape(batch, -1)


def getTempoinput():
    if device_tiles:
        bx, by, bp = tiCr.selectRandomTempoTilesDevice(batch, True, aug, 3, 0.5)
    else:
        bx, by, bp = tiCr.selectRandomTempoTiles(batch, True, aug, 3, 0.5)
    n = bx.shape[0]
    return bx.reshape(n, -1), by.reshape(n, -1), bp.reshape(n, -1)


save_no = 0


def saveModel():
    global save_no
    trainer.sess.sync_to_store()
    allp = trainer.sess.vars.numpy()
    full = dict(allp)
    full.update(trainer.slot_state())
    checkpoint.save(test_path + 'model_%04d.ckpt' % save_no, full)
    ema = dict(allp)
    for n_, e_ in zip(trainer.opt_g.names, trainer.ema):           # MovingAverageOptimizer.swapping_saver (:1366)
        ema[n_] = e_.detach().cpu().numpy()
    checkpoint.save(test_path + 'model_ema_%04d.ckpt' % save_no, ema)
    print('Saved Model %04d.' % save_no)
    save_no += 1


def poly_decay(step):
    """tf.train.polynomial_decay(lr, step, decayIter, lr * 0.05, power=1.1) (:1006-1009)"""
    s = min(step, decayIter)
    return (learning_rate - learning_rate * 0.05) * (1 - s / float(decayIter)) ** 1.1 + learning_rate * 0.05


# growing schedule (:1885-1975)
interpolate_Perc = True
start_interpol = stageIter * int(math.floor(startingIter // (stageIter * 2))) * 2 + stageIter
interpol_c = int(math.floor(startingIter // (stageIter * 2)) * stageIter)
if (startingIter // stageIter) % 2 == 0:
    interpol_c += (startingIter - interpol_c) % stageIter
    interpol_c += stageIter
else:
    interpolate_Perc = False
lrgs = 0
discRuns, genRuns = int(P["discRuns"]), int(P["genRuns"])
outputInterval, saveInterval = int(P["outputInterval"]), int(P["saveInterval"])
decayLR = int(P["decayLR"]) > 0
avg_d = avg_g = avg_l1 = 0.0
t0 = time.time()
print('\n*****TRAINING STARTED***** (stop with ctrl-c)\n')
for it in range(startingIter, trainingIterations):
    if it - start_interpol == 0:
        interpolate_Perc = False
    if it - start_interpol == stageIter and currentUpres < upRes:    # a stage is complete: next resolution, new data
        saveModel()
        currentUpres *= 2
        start_interpol = it + stageIter
        interpolate_Perc = True
        tiCr = load_stage(currentUpres, False)
        print("--------------------------NEW UPRES: %d--------------------------" % currentUpres)
    if interpolate_Perc:
        interpol_c += 1
        currBlendPer = interpol_c / stageIter
    else:
        currBlendPer = int(round(interpol_c / stageIter))
    currBlendPer = min(max(currBlendPer, 1.0), 3.0)
    if it >= stageIter * 6 and decayLR:
        lrgs += 1
    lr = poly_decay(lrgs) if decayLR else learning_rate
    for opt in [trainer.opt_d, trainer.opt_g] + ([trainer.opt_t] if useTempoD else []):
        opt.lr = lr
    index = int(round(math.log(currentUpres, 2)) - 1)               # the growing stage's optimisers (:1978)
    for _ in range(discRuns):
        bx, by = getinput()
        avg_d += float(trainer.disc_step(bx, by, currBlendPer, stage=index)["disc_loss"].detach())
    tempo = None
    if useTempoD:
        for _ in range(discRuns):
            tempo = getTempoinput()
            trainer.tempo_disc_step(tempo[0], tempo[1], tempo[2], currBlendPer, stage=index)
    for _ in range(genRuns):
        bx, by = getinput()
        if useTempoD:
            tempo = getTempoinput()
        L = trainer.gen_step(bx, by, currBlendPer, tempo, stage=index)
        avg_g += float(L["g_loss_d"].detach())
        avg_l1 += float(L["l1_loss"].detach())
    if (it + 1) % outputInterval == 0:
        k = float(outputInterval)
        print('\nIteration {:05d}/{}, Cost:'.format(it + 1, trainingIterations))
        print('\tdisc: loss: train_loss={:.6f}'.format(avg_d / (k * discRuns)))
        print('\tgen: loss: train={:.6f} L1={:.6f}'.format(avg_g / (k * genRuns), avg_l1 / (k * genRuns)))
        print('\t blending percentage: %f' % currBlendPer)
        print('\t{} iterations took {:.2f} seconds.'.format(outputInterval, time.time() - t0))
        avg_d = avg_g = avg_l1 = 0.0
        t0 = time.time()
    if (it + 1) % saveInterval == 0:
        saveModel()
saveModel()
print('\n*****TRAINING FINISHED*****')
print('Test path: %s' % test_path)
