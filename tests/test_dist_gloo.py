"""world_size-2 rehearsal (gloo, CPU) of the slice-axis sharding + all-gather between passes:
the sharded pipelines must reproduce the single-rank result bit for bit.  The generators are
the oracle's (CPU); the partition / collective / transpose logic under test is the product's
``multipass.two_pass_4x`` / ``multipass_8x`` and ``dist.Comm``."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleGen(object):
    """callable with the Generator interface, evaluated by the numpy oracle"""

    def __init__(self, kind, seed, cfg):
        from oracle import nets as ON
        self.ON, self.kind, self.cfg = ON, kind, dict(cfg)
        self.ps = ON.ParamSource(seed=seed)

    def __call__(self, x, y=None):
        ON, c = self.ON, self.cfg
        xn = x.numpy()
        if self.kind == "gen_resnet":
            out = ON.gen_resnet(self.ps, xn, c["up_res"], c["mode"], True)
        elif c["first_gen"]:
            out = ON.growing_gen(self.ps, xn, c["up_res"], True, c["filter_size"], c["start_fms"], c["max_fms"],
                                 c.get("first_nn_arch", False), True)
        else:
            yin = y.numpy().reshape(xn.shape[0], y.shape[1], y.shape[2], 1)
            out = ON.growing_gen(self.ps, ON.gen2_input(yin, xn, yin.shape[1]), c["up_res"], False, c["filter_size"],
                                 c["start_fms"], c["max_fms"], False, c.get("use_res_net", True))
        return torch.as_tensor(out[..., 0])


def _pipelines(comm, exchange="all_gather"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mpgan_amd  # noqa: F401
    from mpgan_amd import multipass as MP
    from mpgan_amd.synthetic import synthetic_volume
    import cpu_backend
    outs = {}
    for nch in (1, 4):
        low = torch.as_tensor(synthetic_volume(4, nch, 3))
        g1 = OracleGen("gen_resnet", 5, dict(up_res=4, mode=2))
        g2 = OracleGen("gen_resnet", 6, dict(up_res=4, mode=1))
        final, v1 = MP.two_pass_4x(g1, g2, low, 4, batch=3, comm=comm, backend=cpu_backend, vel_scale=0.5, exchange=exchange)
        outs["4x_c%d" % nch] = final.numpy()
        if v1 is not None:          # the all-to-all exchange never assembles the pass-1 volume on several ranks
            outs["4x_c%d_v1" % nch] = v1.numpy()
        # the pipelined batch form (exchange of volume i under the passes of its neighbours) is the same arithmetic
        low_b = torch.as_tensor(synthetic_volume(4, nch, 5))
        fb = MP.two_pass_4x_batch(g1, g2, [low, low_b, low], 4, batch=3, comm=comm, backend=cpu_backend, vel_scale=0.5,
                                  exchange=exchange)
        assert np.array_equal(fb[0].numpy(), final.numpy()) and np.array_equal(fb[2].numpy(), final.numpy())
        outs["4x_c%d_batch1" % nch] = fb[1].numpy()
    low = torch.as_tensor(synthetic_volume(2, 4, 4))
    cfgs = [dict(up_res=8, first_gen=True, filter_size=3, start_fms=32, max_fms=32, first_nn_arch=True, add_adj=True),
            dict(up_res=8, first_gen=False, filter_size=3, start_fms=32, max_fms=32),
            dict(up_res=8, first_gen=False, filter_size=3, start_fms=32, max_fms=16, use_res_net=False)]
    gens = [OracleGen("growing_gen", 7 + i, c) for i, c in enumerate(cfgs)]
    for g, c in zip(gens, cfgs):
        g.cfg = c
    for n in (1, 2, 3):
        outs["8x_%dnets" % n] = MP.multipass_8x(gens[:n], low, 8, batches=(4, 2, 2), comm=comm, backend=cpu_backend,
                                                exchange=exchange).numpy()
    return outs


def _worker(rank, world, port, path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import mpgan_amd  # noqa: F401
    from mpgan_amd import dist as mdist
    comm = mdist.Comm()
    assert (comm.rank, comm.world) == (rank, world)
    outs = _pipelines(comm)
    # the all-to-all hand-over between the passes (1/R of the bytes) is the same arithmetic as the all-gather
    for k, v in _pipelines(comm, "all_to_all").items():
        assert np.array_equal(v, outs[k]), k
    blk = torch.arange(2 * 4 * 6, dtype=torch.float32).reshape(2, 4, 6) + 100.0 * rank     # slab of a [4, 4, 6] volume
    got = comm.all_to_all_blocks(blk, 2)
    want = torch.cat([torch.arange(2 * 4 * 6, dtype=torch.float32).reshape(2, 4, 6) + 100.0 * q for q in range(world)], 0)
    assert torch.equal(got, want[:, :, rank * 3:(rank + 1) * 3])
    assert comm.max_float(float(rank), torch.device("cpu")) == world - 1
    # data-parallel training: the flat gradient bucket of an optimiser is averaged over the ranks before
    # the Adam kernel (train.AdamTF.step); here with the kernel stubbed out, on CPU buffers
    from mpgan_amd import train as mtrain
    params = {"a/weight": torch.zeros(3, 2, requires_grad=True), "b/bias": torch.zeros(4, requires_grad=True)}
    opt = mtrain.AdamTF(params, comm=comm)
    seen = {}
    mtrain.train_ops.adam_step = lambda flat, grad, m, v, lr_t, b1, b2, eps: seen.update(grad=grad.clone(), lr=float(lr_t))
    opt.step([torch.full((3, 2), float(rank + 1)), None if rank == 0 else torch.full((4,), 2.0)])
    outs["dp_grad"] = seen["grad"].numpy()
    outs["dp_lr_t"] = np.float64(seen["lr"])
    comm.barrier()
    np.savez(path % rank, **outs)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_pipelines_match_single_rank(tmp_path):
    ref = _pipelines(None)
    world = 2
    path = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(world, _free_port(), path), nprocs=world, join=True)
    for r in range(world):
        got = np.load(path % r)
        for k, v in ref.items():
            assert np.array_equal(got[k], v), (r, k)
        # mean over ranks of (1, 2) and of (unconnected -> 0, 2); Adam step size of t = 1
        assert np.allclose(got["dp_grad"], [1.5] * 6 + [1.0] * 4)
        assert abs(float(got["dp_lr_t"]) - 2e-4 * np.sqrt(1 - 0.999) / (1 - 0.5)) < 1e-12


def test_slice_range_and_errors(mpg):
    from mpgan_amd import multipass as MP

    class C(object):
        rank, world = 1, 4

    assert MP.slice_range(256, C()) == (64, 128)
    with pytest.raises(ValueError):
        MP.slice_range(30, C())
    assert MP.slice_range(10, MP.LocalComm()) == (0, 10)
