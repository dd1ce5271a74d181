"""Weights container of the MI355X path: one ``.npz`` per model holding the variables under
their TF names (``generator/g_cA0/weight`` ...).  The reference stores TF Saver-V2 checkpoints
``test_%04d/model_%04d.ckpt`` (GAN/multipassGAN-out.py:157,367-386); here the same path with
``.npz`` appended is used; when only the TF files (``.index`` + ``.data-00000-of-00001``) exist, ``load``
reads them through ``tf_checkpoint`` (format restated without TensorFlow; unvalidated against a TF-written
file, see that module)."""
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
            # a TensorFlow Saver-V2 checkpoint of the reference (multipassGAN-out.py:367-386): read it directly
            from . import tf_checkpoint
            return {k: v for k, v in tf_checkpoint.read_checkpoint(path).items() if v.dtype == np.float32}
        raise FileNotFoundError(p)
    with np.load(p) as z:
        return {k: z[k] for k in z.files}
