"""CPU stand-in of the operator module for the multi-process (gloo) rehearsal of the sharding
logic: same function names as ``multi-pass-gan_amd/ops.py`` for the marshalling kernels, backed
by the oracle on CPU torch tensors.  Test infrastructure only."""
import numpy as np
import torch

from oracle import multipass as OM
from oracle import ops as O


def _np(t):
    return t.detach().cpu().numpy()


def axis_zoom_linear(v, axis, factor):
    return torch.as_tensor(O.zoom_axis_linear(_np(v), axis, factor))


def volume_transpose(v, perm, chan_map=None, cutoff=0.0):
    a = _np(v)
    a = a.transpose(tuple(perm) + ((3,) if a.ndim == 4 else ()))
    if chan_map is not None:
        a = a[..., list(chan_map)]
    a = np.ascontiguousarray(a)
    if cutoff > 0:
        a = OM.cutoff(a, cutoff)
    return torch.as_tensor(a)


def add_adjacent(x, s_off=0, s_cnt=None):
    a = OM.add_adjacent(_np(x), x.shape[-1])
    s_cnt = a.shape[0] - s_off if s_cnt is None else s_cnt
    return torch.as_tensor(np.ascontiguousarray(a[s_off:s_off + s_cnt]))


def cutoff(v, thr=0.0005, out=None):
    return torch.as_tensor(OM.cutoff(_np(v), thr))


def channel_gather(a, b, cmap, scales=None, scales2=None):
    src = _np(a) if b is None else np.concatenate([_np(a), _np(b)], axis=-1)
    out = src[..., list(cmap)].astype(np.float32)
    for sc in (scales, scales2):
        if sc is not None:
            out = out * np.asarray(sc, np.float32)
    return torch.as_tensor(np.ascontiguousarray(out))
