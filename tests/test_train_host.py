"""Host-side logic of the trainers that needs no GPU: optimiser slot checkpoints, slot-key filtering of model loads."""
import numpy as np
import torch

import mpgan_amd  # noqa: F401
from mpgan_amd import train
from mpgan_amd.session import VariableStore


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    return {"generator/g_cA2/weight": torch.randn(3, 3, 2, 4, generator=g), "generator/g_cA2/bias": torch.randn(4, generator=g),
            "generator/g_cA8/weight": torch.randn(1, 1, 4, 2, generator=g)}


def test_adam_slots_round_trip():
    a = train.AdamTF(_params(), lr=1e-3, beta1=0.5)
    a.m.copy_(torch.randn(a.m.numel()))
    a.v.copy_(torch.rand(a.v.numel()))
    a.t = 17
    state = a.slot_state("gen")
    assert set(k for k in state if k.startswith("generator/")) == {n + s for n in a.names for s in ("/Adam", "/Adam_1")}
    assert np.isclose(state["gen/beta1_power"], 0.5 ** 18)
    b = train.AdamTF(_params(1), lr=1e-3, beta1=0.5)
    assert b.load_slot_state(state, "gen") == 3
    assert torch.equal(a.m, b.m) and torch.equal(a.v, b.v) and b.t == 17
    # another optimiser's tag leaves the step count alone
    c = train.AdamTF(_params(1), lr=1e-3, beta1=0.5)
    c.load_slot_state(state, "disc")
    assert c.t == 0


def test_staged_adam_slots_round_trip():
    a = train.StagedAdam(_params(), levels=3, loss_scaling=True)
    for z in range(3):
        a.ms[z].copy_(torch.randn(a.flat.numel()) * (a.masks[z] if a.masks[z] is not None else 1))
        a.vs[z].copy_(torch.rand(a.flat.numel()) * (a.masks[z] if a.masks[z] is not None else 1))
        a.state[z][3] = 5 + z
        a.state[z][0] = 60.0 - z
    state = a.slot_state("gen")
    # stage 0 owns the "1"/"2" variables only (multipassGAN-8x.py:1316-1321)
    assert "generator/g_cA2/weight/Adam" in state and "generator/g_cA8/weight/Adam" not in state
    assert "generator/g_cA8/weight/Adam_4" in state and "generator/g_cA8/weight/Adam_5" in state
    b = train.StagedAdam(_params(1), levels=3, loss_scaling=True)
    b.load_slot_state(state, "gen")
    for z in range(3):
        assert torch.equal(a.ms[z], b.ms[z]) and torch.equal(a.vs[z], b.vs[z])
        assert float(b.state[z][3]) == 5 + z and float(b.state[z][0]) == 60.0 - z


def test_model_load_skips_optimiser_slots():
    vs = VariableStore("cpu", seed=1)
    vs.load({"generator/g_cA2/weight": np.ones((1, 1, 1, 1), np.float32),
             "generator/g_cA2/weight/Adam": np.zeros((1, 1, 1, 1), np.float32),
             "generator/g_cA2/weight/Adam_3": np.zeros((1, 1, 1, 1), np.float32),
             "beta1_power": np.float32(0.5), "beta2_power_1": np.float32(0.9), "gen/stage1/adam_t": np.int64(3),
             "gen/stage1/ls_var": np.float32(64), "gen/adam_t": np.int64(2), "gen/beta1_power": np.float32(0.1)})
    assert list(vs.values) == ["generator/g_cA2/weight"]
