"""Weights container of the MI355X path: one ``.npz`` per model holding the variables under
their TF names (``generator/g_cA0/weight`` ...).  The reference stores TF Saver-V2 checkpoints
``test_%04d/model_%04d.ckpt`` (GAN/multipassGAN-out.py:157,367-386); here the same path with
``.npz`` appended is used.  Importing Saver-V2 files directly is a listed next step."""
import os

import numpy as np


def model_path(base_path, test_no, model_no, ema=False):
    return base_path + "test_%04d/model_%s%04d.ckpt" % (test_no, "ema_" if ema else "", model_no)


def save(path, params):
    np.savez(path if path.endswith(".npz") else path + ".npz", **{k: np.asarray(v) for k, v in params.items()})


def load(path):
    p = path if path.endswith(".npz") else path + ".npz"
    if not os.path.exists(p):
        if os.path.exists(path + ".index"):
            raise FileNotFoundError("%s is a TensorFlow Saver-V2 checkpoint; convert it to %s first "
                                    "(no TF importer in this build yet)" % (path, p))
        raise FileNotFoundError(p)
    with np.load(p) as z:
        return {k: z[k] for k in z.files}
