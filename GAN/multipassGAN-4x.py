#!/usr/bin/env python3
"""4x driver: same ``name value`` command line and files as the reference's GAN/multipassGAN-4x.py
(params :32-144).

Output mode (``out 1``; generate3DUniForNewNetwork :1090-1169, output loop :1634-1646): one network per
invocation, the intermediate volume travels through ``density_low_2x2_%04d.uni`` exactly as in the
reference (example_run_output.py:4-8):

  upsamplingMode 2, upsampledData 0 : zoom z, slices along z  -> density_low_2x2_%04d.uni
  upsamplingMode 1, upsampledData 1 : slices along x          -> density_low_1x1_%04d.uni
  upsamplingMode 3, upsampledData 1 : slices along y          -> density_low_0x0_%04d.uni   (third network, :1121-1124)

Training mode (``out 0``) trains the first network (upsamplingMode 2, upsampledData 0) or the second one
(upsamplingMode 1, upsampledData 1: slices of the zoomed volumes with the first network's output as density) the way
the reference loop does (:196-300 data, :728-902 graph, :1300-1360 iteration): FluidDataLoader slices ->
TileCreator tiles (augmentation, coherent triples) -> ``train.Trainer4x`` (spatial + temporal
discriminator) -> ``basePath/test_%04d/model_%04d.ckpt.npz``.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import mpgan_amd  # noqa: E402,F401
from mpgan_amd import checkpoint, multipass, ops, uniio  # noqa: E402
from mpgan_amd import fluiddataloader as FDL  # noqa: E402
from mpgan_amd import paramhelpers as ph  # noqa: E402

# every parameter name of the reference is accepted (ph.checkUnusedParams aborts on unknown ones)
P = {}
for name, default in [
        ("out", False), ("basePath", '../2ddata_gan/'), ("randSeed", 1), ("load_model_test", -1), ("load_model_no", -1),
        ("simSize", 64), ("tileSize", 16), ("upRes", 4), ("packedSimPath", '/data/share/GANdata/2ddata_sim/'),
        ("fromSim", 1000), ("toSim", -1), ("dataDim", 2), ("numOut", 200), ("saveOut", False), ("loadOut", -1),
        ("img", True), ("gif", False), ("ref", False), ("frame_min", 0), ("genModel", 'gen_test'),
        ("discModel", 'disc_test'), ("learningRate", 0.0002), ("decayLR", False), ("dropout", 1.0),
        ("dropoutOutput", 1.0), ("adam_beta1", 0.5), ("weight_dld", 1.0), ("lambda", 1.0), ("lambda2", 0.0),
        ("lambda_f", 1.0), ("lambda2_f", 1.0), ("lambda2_l1", 1.0), ("lambda2_l2", 1.0), ("lambda2_l3", 1.0),
        ("lambda2_l4", 1.0), ("lambda_t", 1.0), ("lambda_t_l2", 0.0), ("batchSize", 128), ("batchSizeDisc", 128),
        ("batchSizeGen", 128), ("trainGAN", True), ("trainingEpochs", 100000), ("trainingIterations", 100000),
        ("discRuns", 1), ("genRuns", 1), ("batchNorm", True), ("bnDecay", 0.999), ("useVelocities", 0),
        ("useVorticities", 0), ("useFlags", 0), ("useK_Eps_Turb", 0), ("premadeTiles", 0), ("cropOverlap", 0),
        ("dataAugmentation", 0), ("minScale", 0.85), ("maxScale", 1.15), ("rot", 2), ("flip", 1), ("pretrain", 0),
        ("pretrainDisc", 0), ("pretrainGen", 0), ("testPathStartNo", 0), ("testInterval", 100), ("numTests", 10),
        ("outputInterval", 100), ("saveInterval", 200), ("alwaysSave", True), ("keepMax", 3), ("genTestImg", -1),
        ("note", ""), ("data_fraction", 0.3), ("frame_max", 200), ("adv_flag", True), ("change_velocity", False),
        ("saveMetaData", 0), ("use_spatialdisc", True), ("clamping", True), ("simLowLength", 64), ("simLowWidth", 64),
        ("simLowHeight", 64), ("overlappedpixel", 3), ("startIndex", 0), ("useAvgDepool", False), ("avgMode", 0),
        ("velScale", 1.0), ("upsamplingMode", 2), ("upsampledData", False), ("sliceMode", 0), ("interpMode", 1),
        ("genUni", False), ("setVelZero", False), ("upsampleFirst", True), ("synthWeights", 0), ("prec", "2"), ("trainPrec", "3"),
        ("deviceTiles", 1)]:
    P[name] = ph.getParam(name, default)
ph.checkUnusedParams()

outputOnly = int(P["out"]) > 0
basePath, packedSimPath = P["basePath"], P["packedSimPath"]
simSizeLow, upRes = int(P["simSize"]), int(P["upRes"])
fromSim, frame_min, frame_max = int(P["fromSim"]), int(P["frame_min"]), int(P["frame_max"])
useVelocities, velScale = int(P["useVelocities"]), float(P["velScale"])
upsampling_mode, upsampled_data = int(P["upsamplingMode"]), int(P["upsampledData"])
generateUni = int(P["genUni"])
batch_norm = int(P["batchNorm"]) > 0
load_model_test, load_model_no = int(P["load_model_test"]), int(P["load_model_no"])


def train_main():
    """multipassGAN-4x.py with `out 0`, first network"""
    from mpgan_amd import tilecreator_t as tc
    from mpgan_amd.train import Trainer4x
    mode, upsampled = int(P["upsamplingMode"]), int(P["upsampledData"])
    if int(P["dataDim"]) != 2 or (mode, upsampled) not in ((2, 0), (1, 1)):
        print("ERROR: training is implemented for the first network (upsamplingMode 2, upsampledData 0) and the second "
              "one (upsamplingMode 1, upsampledData 1), dataDim 2")
        exit(1)
    if int(P["useVorticities"]) or int(P["useFlags"]) or int(P["useK_Eps_Turb"]) or int(P["premadeTiles"]):
        print("ERROR: vorticity / flag / k-eps inputs and premade tiles are not supported")
        exit(1)
    if int(P["pretrain"]) or int(P["pretrainDisc"]) or int(P["pretrainGen"]):
        print("ERROR: the pretraining phases (pretrain / pretrainDisc / pretrainGen, 4x.py:1232-1296) are not built")
        exit(1)
    if int(P["genTestImg"]) > -1:
        print("ERROR: PNG test images during training (genTestImg, 4x.py:1581-1600) are not built")
        exit(1)
    # `dropout` / `dropoutOutput` feed keep_prob, which no layer of the reference graph reads (4x.py:647-657)
    tileSizeLow, toSim = int(P["tileSize"]), int(P["toSim"])
    toSim = fromSim if toSim == -1 else toSim
    randSeed = int(P["randSeed"])
    kt, kt_l = float(P["lambda_t"]), float(P["lambda_t_l2"])
    useTempoD, useTempoL2 = kt > 1e-6, kt_l > 1e-6                     # 4x.py:147-152
    channelLayout_low, mfl, mfh = 'd', ["density"], ["density"]
    if useVelocities:
        channelLayout_low += ',vx,vy,vz'
        mfl = mfl + ["velocity"]
    dirIDs = np.linspace(fromSim, toSim, (toSim - fromSim + 1), dtype='int16')
    data_fraction = float(P["data_fraction"])
    # deviceTiles 1 (default): the frames live in HBM and the batches are cut / augmented by HIP kernels
    # (tiles_device.DeviceTileCreator: the reference's random decisions in its draw order, tilecreator_t.py:457-546,
    # 1345-1414) and reach the trainer as device tensors; 0: the host TileCreator (numpy / scipy, one thread)
    device_tiles = int(P["deviceTiles"]) > 0
    if device_tiles:
        from mpgan_amd.tiles_device import DeviceTileCreator

        def Tiles(**kw):
            return DeviceTileCreator(device=device, **kw)
    else:
        Tiles = tc.TileCreator
    if mode == 1:
        # second network (:234,241-248,253-262,291-299): slices along x of volumes zoomed to the high resolution,
        # their density channel replaced by the first network's output (density_low_2x2_%04d.uni)
        n_t = 3
        mol = [o for o in range(3) for _ in mfl]
        moh = [o for o in range(3) for _ in mfh]
        tiCr = Tiles(tileSizeLow=tileSizeLow * upRes, densityMinimum=0.005, channelLayout_high='d',
                              simSizeLow=simSizeLow * upRes, dim=2, dim_t=3, channelLayout_low=channelLayout_low, upres=1,
                              premadeTiles=False)
        common = dict(print_info=0, base_path=packedSimPath, base_path_y=packedSimPath, numpy_seed=randSeed,
                      conv_slices=True, conv_axis=2, select_random=0.1, density_threshold=0.002,
                      axis_scaling_y=[1, 1, 1, 1], axis_scaling=[upRes, upRes, upRes, 1], filename="density_low_%04d.uni",
                      oldNamingScheme=False, filename_index_max=frame_max, filename_index_min=frame_min, indices=dirIDs,
                      data_fraction=data_fraction, multi_file_list_y=mfh * 3, multi_file_idxOff_y=moh)
        fl2 = FDL.FluidDataLoader(filename_y="density_low_2x2_%04d.uni", multi_file_list=["density"] * 3,
                                  multi_file_idxOff=[0, 1, 2], **common)
        fl = FDL.FluidDataLoader(filename_y="density_high_%04d.uni", multi_file_list=mfl * 3, multi_file_idxOff=mol, **common)
    elif not useTempoD:
        tiCr = Tiles(tileSizeLow=tileSizeLow, simSizeLow=simSizeLow, dim=2, dim_t=1, channelLayout_low=channelLayout_low,
                              upres=upRes, premadeTiles=False, channelLayout_high='d')
        fl = FDL.FluidDataLoader(print_info=1, base_path=packedSimPath, base_path_y=packedSimPath, numpy_seed=randSeed,
                                 conv_slices=True, conv_axis=0, select_random=0.1, density_threshold=0.002,
                                 axis_scaling_y=[1.0 / upRes, 1, 1, 1], axis_scaling=[1, 1, 1, 1],
                                 filename="density_low_%04d.uni", filename_index_min=frame_min, oldNamingScheme=False,
                                 filename_y="density_high_%04d.uni", filename_index_max=frame_max, indices=dirIDs,
                                 data_fraction=data_fraction, multi_file_list=mfl, multi_file_list_y=mfh)
        n_t = 1
    else:                                                              # three coherent frames per sample (:214-262)
        n_t = 3
        mol = [o for o in range(3) for _ in mfl]
        moh = [o for o in range(3) for _ in mfh]
        tiCr = Tiles(tileSizeLow=tileSizeLow, densityMinimum=0.005, channelLayout_high='d', simSizeLow=simSizeLow,
                              dim=2, dim_t=3, channelLayout_low=channelLayout_low, upres=upRes, premadeTiles=False)
        fl = FDL.FluidDataLoader(print_info=0, base_path=packedSimPath, base_path_y=packedSimPath, numpy_seed=randSeed,
                                 conv_slices=True, conv_axis=0, select_random=0.1, density_threshold=0.002,
                                 axis_scaling_y=[1.0 / upRes, 1, 1, 1], axis_scaling=[1, 1, 1, 1],
                                 filename="density_low_%04d.uni", oldNamingScheme=False, filename_y="density_high_%04d.uni",
                                 filename_index_max=frame_max, filename_index_min=frame_min, indices=dirIDs,
                                 data_fraction=data_fraction, multi_file_list=mfl * 3, multi_file_idxOff=mol,
                                 multi_file_list_y=mfh * 3, multi_file_idxOff_y=moh)
    if int(P["dataAugmentation"]):
        tiCr.initDataAugmentation(rot=int(P["rot"]), minScale=float(P["minScale"]), maxScale=float(P["maxScale"]),
                                  flip=int(P["flip"]))
    x, y, _ = fl.get()
    if mode == 1:
        _, x_2, _ = fl2.get()
        x = x.reshape(-1, 1, simSizeHigh, simSizeHigh, n_ch * n_t)    # :292-297
        x_2 = x_2.reshape(-1, 1, simSizeHigh, simSizeHigh, n_t)
        for i in range(n_ch * n_t):
            if i % n_ch == 0:
                x[:, :, :, :, i:i + 1] = x_2[:, :, :, :, i // n_ch:i // n_ch + 1]
    else:
        x = x.reshape(-1, 1, simSizeLow, simSizeLow, n_ch * n_t)      # :288-290
    y = y.reshape(-1, 1, simSizeHigh, simSizeHigh, n_t)
    tiCr.addData(x, y)
    np.random.seed(randSeed)
    test_path, _ = ph.getNextTestPath(int(P["testPathStartNo"]), basePath)
    print("\nUsing parameters:\n" + ph.paramsToString())
    ph.writeParams(test_path + "params.json")
    batch = int(P["batchSize"])
    trainer = Trainer4x(tileSizeLow=tileSizeLow, upRes=upRes, n_inputChannels=n_ch, batch_norm=batch_norm,
                        upsampling_mode=mode, device=device, learning_rate=float(P["learningRate"]),
                        beta1=float(P["adam_beta1"]), lambda_l1=float(P["lambda"]), lambda2=float(P["lambda2"]),
                        lambda2_l=tuple(float(P["lambda2_l%d" % i]) for i in (1, 2, 3, 4)),
                        weight_dld=float(P["weight_dld"]), bn_decay=float(P["bnDecay"]), seed=randSeed,
                        use_tempo=useTempoD, lambda_t=kt, adv_flag=int(P["adv_flag"]) > 0, clamping=int(P["clamping"]) > 0,
                        lambda_t_l2=kt_l, prec=ops.parse_prec(P["trainPrec"]))
    if load_model_test >= 0:
        params = checkpoint.load(checkpoint.model_path(basePath, load_model_test, load_model_no))
        with torch.no_grad():
            for n_, t_ in trainer.sess.params.items():
                if n_ in params:
                    t_.copy_(torch.as_tensor(params[n_], device=t_.device))
        n_slots = trainer.load_slot_state(params)                     # the Saver restores the Adam slots too (:961-975)
        print("Model restored (%d variables with optimiser slots)." % n_slots)
    aug = int(P["dataAugmentation"]) > 0
    n_out = (tileSizeLow * upRes) ** 2
    n_in = (tileSizeLow * tileSizeLow if mode == 2 else n_out) * n_ch

    def getinput():
        if device_tiles:
            bx, by = tiCr.selectRandomTilesDevice(batch, augment=aug)
        else:
            bx, by = tiCr.selectRandomTiles(selectionSize=batch, augment=aug)
        return bx.reshape(-1, n_in), by.reshape(-1, n_out)

    def gettempo():
        if device_tiles:
            return tiCr.selectRandomTempoTilesDevice(batch, True, aug, n_t=3, dt=0.5)
        return tiCr.selectRandomTempoTiles(batch, True, aug, n_t=3, dt=0.5)

    keep_max, kept = int(P["keepMax"]), []

    def save(no):
        trainer.sess.sync_to_store()
        state = dict(trainer.sess.vars.numpy())
        state.update(trainer.slot_state())
        checkpoint.save(test_path + 'model_%04d.ckpt' % no, state)
        kept.append(test_path + 'model_%04d.ckpt.npz' % no)
        while keep_max > 0 and len(kept) > keep_max:                   # tf.train.Saver(max_to_keep=maxToKeep), :958
            os.remove(kept.pop(0))
        print('Saved Model with number %d' % no)

    epochs = int(P["trainingEpochs"])
    lr0, decay_lr = float(P["learningRate"]), int(P["decayLR"]) > 0
    k_f, k2_f = float(P["lambda_f"]), float(P["lambda2_f"])

    def decayed_lr(epoch):
        """tf.train.polynomial_decay(lr, lrgs, epochs // 2, lr * 0.05, power=1.1) with lrgs = max(0, epoch - epochs // 2)
        (4x.py:773-774,1304)"""
        half = max(epochs // 2, 1)
        step = min(max(0, epoch - epochs // 2), half)
        return (lr0 - lr0 * 0.05) * (1.0 - step / float(half)) ** 1.1 + lr0 * 0.05
    discRuns, genRuns = int(P["discRuns"]), int(P["genRuns"])
    outputInterval, saveInterval = int(P["outputInterval"]), int(P["saveInterval"])
    save_no, t0 = 0, time.time()
    avg_d = avg_g = avg_l1 = 0.0
    print('\n*****TRAINING STARTED*****\n')
    for epoch in range(epochs):
        if decay_lr:
            trainer.set_learning_rate(decayed_lr(epoch))
        for _ in range(discRuns):
            bx, by = getinput()
            avg_d += float(trainer.disc_step(bx, by)["disc_loss"].detach())
        tempo = None
        if useTempoD:
            for _ in range(discRuns):
                tempo = gettempo()
                trainer.tempo_disc_step(*tempo)
        for _ in range(genRuns):
            bx, by = getinput()
            trainer.k, trainer.k2 = k_f * trainer.k, k2_f * trainer.k2   # :1342-1343
            if useTempoD or useTempoL2:                                  # :1352-1363
                tempo = gettempo()
                L = trainer.gen_step_tempo(bx, by, *tempo)
            else:
                L = trainer.gen_step(bx, by)
            avg_g += float(L["gen_loss"].detach())
            avg_l1 += float(L["gen_l1_loss"].detach())
        if (epoch + 1) % outputInterval == 0:
            k = float(outputInterval)
            print('\nEpoch {:05d}/{}, Cost:'.format(epoch + 1, epochs))
            print('\tdisc: loss: train_loss={:.6f}'.format(avg_d / (k * discRuns)))
            print('\tgen: loss: train={:.6f} L1={:.6f}'.format(avg_g / (k * genRuns), avg_l1 / (k * genRuns)))
            print('\t{} epochs took {:.2f} seconds.'.format(outputInterval, time.time() - t0))
            avg_d = avg_g = avg_l1 = 0.0
            t0 = time.time()
        if (epoch + 1) % saveInterval == 0:
            save(save_no)
            save_no += 1
    save(save_no)
    print('\n*****TRAINING FINISHED*****')
    print('Test path: %s' % test_path)


if not outputOnly:
    n_ch = 4 if useVelocities else 1
    simSizeHigh = simSizeLow * upRes
    device = "cuda:0"
    train_main()
    exit(0)
if upsampling_mode not in (1, 2, 3) or int(P["dataDim"]) != 2 or int(P["useAvgDepool"]):
    print("ERROR: upsamplingMode 2 (first network), 1 (second) and 3 (third network) of the 2D slice path are implemented; "
          "mode 0 (linear interpolation between the networks) is used by no example run")
    exit(1)
simSizeHigh = simSizeLow * upRes
n_ch = 4 if useVelocities else 1
device = "cuda:0"

mfl = ["density"] + (["velocity"] if useVelocities else [])
floader = FDL.FluidDataLoader(print_info=1, base_path=packedSimPath, base_path_y=packedSimPath, numpy_seed=int(P["randSeed"]),
                              filename="density_low_%04d.uni", filename_index_min=frame_min, oldNamingScheme=False,
                              filename_y=None, filename_index_max=frame_max, indices=[fromSim], data_fraction=1.0,
                              multi_file_list=mfl, multi_file_list_y=["density"])
x, _, _ = floader.get()
x_2 = None
if upsampled_data:
    # the previous network's volumes (4x.py:180-185)
    fl2 = FDL.FluidDataLoader(print_info=1, base_path=packedSimPath, numpy_seed=int(P["randSeed"]),
                              filename="density_low_1x1_%04d.uni" if upsampling_mode == 3 else "density_low_2x2_%04d.uni",
                              filename_index_min=frame_min, oldNamingScheme=False,
                              filename_index_max=frame_max, indices=[fromSim], data_fraction=1.0, multi_file_list=["density"])
    x_2, _, _ = fl2.get()

path = checkpoint.model_path(basePath, load_model_test, load_model_no)
try:
    params = checkpoint.load(path)
    print("Model restored from %s." % path)
except FileNotFoundError as e:
    if not int(P["synthWeights"]):
        print("ERROR: %s" % e)
        exit(1)
    params = None
gen = multipass.Generator("gen_resnet", dict(tile_low=simSizeLow, up_res=upRes, channels=n_ch,
                                             upsampling_mode=upsampling_mode, batch_norm=batch_norm),
                          params, prec=ops.parse_prec(P["prec"]), device=device, seed=int(P["randSeed"]))
print('*****OUTPUT ONLY*****')
s = simSizeHigh
for layerno in range(frame_min, frame_max):
    i = layerno - frame_min
    start = time.time()
    low = torch.as_tensor(np.ascontiguousarray(x[i])).to(device)
    if upsampling_mode == 2:
        if n_ch > 1:
            low[..., 1:4] *= velScale                                        # 4x.py:283
        xs = ops.axis_zoom_linear(low, 0, upRes)                             # 4x.py:1103
        out = multipass._run_pass(gen, xs, None, 0, s, 8)                    # (z, y, x)
        vol = ops.cutoff(out, multipass.CUTOFF) if generateUni else out
        name = 'density_low_2x2_%04d.uni'
    else:
        # upsamplingMode 1: planes (z,y) along x -> density_low_1x1; upsamplingMode 3: planes (z,x) along y -> density_low_0x0
        v1 = torch.as_tensor(np.ascontiguousarray(x_2[i][..., 0])).to(device)
        vol = multipass.refine_pass_4x(gen, low, v1, upRes, mode=upsampling_mode, batch=8, vel_scale=velScale,
                                       apply_cutoff=bool(generateUni))
        name = 'density_low_1x1_%04d.uni' if upsampling_mode == 1 else 'density_low_0x0_%04d.uni'
    torch.cuda.synchronize()
    print(time.time() - start)
    if generateUni:
        head, _ = uniio.readUni(packedSimPath + "sim_%04d/density_low_%04d.uni" % (fromSim, layerno))
        head['dimX'] = head['dimY'] = head['dimZ'] = simSizeHigh
        uniio.writeUniFromDevice(packedSimPath + '/sim_%04d/' % fromSim + name % layerno, head, vol)
    print('')
print('Test finished, %d volumes written to %s.' % (frame_max - frame_min, packedSimPath))
