"""One process per GPU, started by a parent that never touches the GPU.

``python bench.py --gpus N`` (no torch.distributed.run in front) lands here: the parent
process must not initialise HIP (no ``torch.cuda.*`` call, not even ``is_available()``):
a process that did may neither fork usable children nor exec.  It starts N fresh
children of the same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT in their environment (what torch.distributed.run would set), relays rank 0's
stdout and exits with the worst child return code.

The reference is single-process / single-GPU (GAN/multipassGAN-out.py:96-97): there is
no launcher to mirror.
"""
import os
import socket
import subprocess
import sys
import time


def free_port(host="127.0.0.1"):
    s = socket.socket()
    s.bind((host, 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    """environment of one rank: the variables torch.distributed.run exports, plus the dmabuf-IPC
    switch RCCL needs on this driver (it must be in the environment BEFORE the HIP runtime starts)"""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def spawn_ranks(script, argv, world, timeout=None, poll=0.2, python=None):
    """Start `world` children ``python script argv...``; returns (worst_rc, rank0_stdout).
    Rank 0's stdout is captured (and returned); every rank's stderr and the other ranks' stdout go
    to this process's stderr.  When one child fails the others are killed by PID."""
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([python or sys.executable, script] + list(argv), env=rank_env(r, world, port),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    t0 = time.time()
    worst = 0
    live = set(range(world))
    out0 = b""
    try:
        # rank 0's pipe is drained by communicate() at the end; its output is one JSON line, far below
        # the pipe buffer, so polling the return codes first cannot dead-lock on a full pipe
        while live:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0:
                    worst = worst or rc
            if worst and live:
                break
            if timeout is not None and time.time() - t0 > timeout:
                worst = worst or 124
                break
            if live:
                time.sleep(poll)
    finally:
        for r in sorted(live):
            procs[r].kill()
        for p in procs:
            try:
                o, _ = p.communicate(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
                o, _ = p.communicate()
            if p is procs[0] and o:
                out0 = o
            if p.returncode and not worst:
                worst = p.returncode
    return worst, out0.decode("utf-8", "replace")


def last_json_line(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            return line
    return None
