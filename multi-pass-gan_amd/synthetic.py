"""Synthetic inputs of the bench / tests: smooth smoke-like density in
[0,1] with ~40 % zeros and blurred velocities; seeds numpy default_rng(1234 + index).
Volumes are [z,y,x,c] like .uni payloads (tools_wscale/uniio.py:40-44)."""
import numpy as np


def synthetic_volume(sim, channels=1, index=0):
    import scipy.ndimage
    rng = np.random.default_rng(1234 + index)
    d = scipy.ndimage.gaussian_filter(rng.random((sim, sim, sim)), 3.0, mode="wrap")
    d = (d - d.mean()) / (d.std() + 1e-12)
    d = np.clip(d * 0.5 + 0.1, 0.0, 1.0)
    vol = np.zeros((sim, sim, sim, channels), dtype=np.float32)
    vol[..., 0] = d
    for c in range(1, channels):
        v = scipy.ndimage.gaussian_filter(rng.standard_normal((sim, sim, sim)) * 0.5, 2.0, mode="wrap")
        vol[..., c] = v * 4.0
    return vol
