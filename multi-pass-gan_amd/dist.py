"""Slice-axis sharding over the GPUs of one node: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" for the
CPU rehearsal of the partition logic).

The reference is single-process / single-GPU (GAN/multipassGAN-out.py:96-97); the
only exchange this path needs is the re-assembly of a pass's output volume
before the next pass slices it along another axis: an all-gather of the
per-rank slabs (S^3*4/R bytes each; 67 MB at 512^3, R = 8).
"""
import os

# dmabuf IPC (RCCL and tensor sharing across processes on this driver): read when the HIP runtime starts, so it is
# exported at import, before anything can have touched the GPU -- launch.rank_env / bench.py set it even earlier
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


class _Gathered(object):
    def __init__(self, full, work, keep=None):
        self.full, self.work, self.keep = full, work, keep

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = self.keep = None
        return self.full


class Comm(object):
    """`group`: the process group of the data exchanges (RCCL for device tensors).  `host_group`: a gloo group for the
    control traffic of a measurement (barrier, max over ranks of a host float): it does not depend on RCCL, so a
    run without a data-path collective (whole volumes per rank) needs no RCCL at all."""

    def __init__(self, group=None, host_group=None):
        self.group = group
        self.host_group = host_group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def _data_group(self, t):
        if t.is_cuda and self.group is None and dist.get_backend() == "gloo":
            # default group is the gloo control group: the RCCL group is made at the first device exchange
            # (every rank reaches it at the same point: new_group is collective)
            self.group = dist.new_group(backend="nccl")
        return self.group

    def all_gather_slabs(self, local, total):
        """local: this rank's contiguous slab [total/world, ...] -> full [total, ...] on every rank"""
        if self.world == 1:
            return local
        per = total // self.world
        if local.shape[0] != per:
            raise ValueError("slab has %d slices, expected %d" % (local.shape[0], per))
        local = local.contiguous()
        full = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if local.is_cuda:
            dist.all_gather_into_tensor(full, local, group=self._data_group(local))
        else:
            dist.all_gather([full[i * per:(i + 1) * per] for i in range(self.world)], local, group=self.group)
        return full

    def all_gather_slabs_start(self, local, total):
        """non-blocking all_gather_slabs: returns a handle whose wait() makes the current stream (RCCL) or
        the host (gloo) wait and returns the full tensor, so the exchange overlaps the work issued in between"""
        if self.world == 1:
            return _Gathered(local, None)
        per = total // self.world
        if local.shape[0] != per:
            raise ValueError("slab has %d slices, expected %d" % (local.shape[0], per))
        local = local.contiguous()
        full = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if local.is_cuda:
            # RCCL: one flat receive buffer (the slabs are contiguous along the slice axis), no staging copies
            work = dist.all_gather_into_tensor(full, local, group=self._data_group(local), async_op=True)
        else:
            work = dist.all_gather([full[i * per:(i + 1) * per] for i in range(self.world)], local, group=self.group,
                                   async_op=True)
        return _Gathered(full, work, keep=local)

    def all_to_all_blocks_start(self, local, dim):
        """The other way to hand a pass's output to the next pass: rank r holds the slab [S/R, ...] of a
        volume along axis 0 and the next pass needs, on rank r, the volume's range r along axis `dim` (its own slice
        axis) -- but ALL of axis 0.  Every rank sends rank q the block `local[..., range q along dim, ...]` and receives
        R blocks of S^3/R^2 elements: 1/R of the bytes an all-gather moves.  Returns a handle whose wait() gives
        [S, ...] with axis `dim` cut to this rank's range (block q at rows [q S/R, (q+1) S/R))."""
        if self.world == 1:
            return _Gathered(local, None)
        r = self.world
        if dim < 1 or dim >= local.dim() or local.shape[dim] % r:
            raise ValueError("all_to_all_blocks: axis %d of %s does not divide over %d ranks" % (dim, tuple(local.shape), r))
        per = local.shape[dim] // r
        shp = list(local.shape)
        # [p, .., R, per, ..] -> [R, p, .., per, ..]: block q is contiguous
        send = local.reshape(shp[:dim] + [r, per] + shp[dim + 1:]).movedim(dim, 0).contiguous()
        recv = torch.empty_like(send)
        # one flat buffer each way (equal splits); gloo implements this form too (the CPU rehearsal)
        work = dist.all_to_all_single(recv, send, group=self._data_group(local) if local.is_cuda else self.group, async_op=True)
        out_shape = [r * shp[0]] + shp[1:dim] + [per] + shp[dim + 1:]
        return _Gathered(recv.reshape(out_shape), work, keep=send)

    def all_to_all_blocks(self, local, dim):
        return self.all_to_all_blocks_start(local, dim).wait()

    def all_reduce_mean(self, flat):
        """data-parallel training: average one flat gradient buffer (one bucket per optimiser) in place"""
        if self.world == 1:
            return flat
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self._data_group(flat))
        flat.mul_(1.0 / self.world)
        return flat

    def barrier(self):
        if self.host_group is not None or dist.get_backend() == "gloo":
            dist.barrier(group=self.host_group)
        else:
            dist.barrier(group=self.group)

    def max_float(self, v, device):
        if self.host_group is not None or dist.get_backend() == "gloo":
            t = torch.tensor([float(v)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.host_group)
        else:
            t = torch.tensor([float(v)], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())


def init_from_env(backend=None, timeout_s=600):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as set by torch.distributed.run; returns (comm, device).
    On GPUs the default process group is gloo (control traffic: barriers, the max over ranks of a time) and the RCCL
    group for the device exchanges is created at the first one (Comm._data_group): a job without a data-path
    collective never initialises RCCL.  backend="nccl" / "gloo" forces one group for everything."""
    import datetime
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if os.environ.get("MPGAN_SHARE_DEVICE"):          # rehearsal of an N-rank job on a one-GPU box (no RCCL possible)
        local_rank = 0
    device = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world == 1:
        return None, device
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if not dist.is_initialized():
        to = datetime.timedelta(seconds=timeout_s)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=to)
        else:
            dist.init_process_group("gloo", timeout=to)
    return Comm(), device
