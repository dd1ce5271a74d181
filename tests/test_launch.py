"""The self-launch path of `bench.py --gpus N` (multi-pass-gan_amd/launch.py), rehearsed on CPU:
N fresh children with the torch.distributed.run environment, rank 0's JSON line relayed, worst
return code propagated.  The parent must never initialise the GPU runtime."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ECHO = os.path.join(ROOT, "tests", "_rank_echo.py")


@pytest.mark.timeout(300)
def test_spawn_ranks_relays_rank0_json(mpg):
    from mpgan_amd import launch
    rc, out = launch.spawn_ranks(ECHO, [], 2, timeout=240)
    assert rc == 0, out
    line = launch.last_json_line(out)
    assert json.loads(line) == {"n_gpus": 2, "sum": 3.0}
    assert "rank 1" not in out            # only rank 0's stdout is relayed


@pytest.mark.timeout(300)
def test_spawn_ranks_propagates_failure(mpg):
    from mpgan_amd import launch
    rc, out = launch.spawn_ranks(ECHO, ["1"], 2, timeout=240)
    assert rc == 3


def test_rank_env_and_json_picker(mpg):
    from mpgan_amd import launch
    env = launch.rank_env(1, 4, 1234, base={})
    assert env["RANK"] == "1" and env["LOCAL_RANK"] == "1" and env["WORLD_SIZE"] == "4"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "1234"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert launch.last_json_line("warning\n{\"a\": 1}\ntrailing text\n") == "{\"a\": 1}"
    assert launch.last_json_line("no json here") is None


@pytest.mark.timeout(600)
def test_bench_parent_launches_children_without_touching_the_gpu():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent spawns two ranks.  Without a GPU the ranks stop
    with "needs an MI355X"; the parent then runs the exchange-free partition once with fresh ranks (for the record
    only) and reports the FAILED benchmark: rc != 0 and a JSON line whose `value` is null -- a failed sharded run must
    never look like a pass -- and it does not hang.
    The parent path does not even import torch (checked below), so it cannot have initialised the runtime."""
    chk = ("import sys; sys.path.insert(0, %r); import bench, mpgan_amd.launch; "
           "assert 'torch' not in sys.modules, 'parent path imports torch'" % ROOT)
    subprocess.run([sys.executable, "-c", chk], check=True, timeout=120)
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=500)
    assert p.returncode != 0
    # the sharded job fails, the parent then tries the exchange-free partition (whole volumes per rank), which fails too
    # (a rank may be stopped by the launcher before it has printed, once its sibling has failed: 2..4 messages)
    assert 2 <= p.stderr.count("needs an MI355X") <= 4, p.stderr[-2000:]
    assert "measuring whole volumes per rank for the record" in p.stderr
    import json
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 2 and "sharded_error" in line and "value_replicas" not in line


@pytest.mark.timeout(600)
def test_bench_train_parent_launches_children_without_touching_the_gpu():
    """`python bench_train.py --gpus 2` with no WORLD_SIZE starts two fresh ranks through launch.spawn_ranks (it used to
    exit with "launch with torch.distributed.run": VERDICT r2, missing 3); the parent path imports no torch.  Without a
    GPU the ranks stop with "no GPU visible" and the parent reports that (rc != 0, no JSON line)."""
    chk = ("import sys; sys.path.insert(0, %r); import bench_train; "
           "assert 'torch' not in sys.modules, 'parent path imports torch'" % ROOT)
    subprocess.run([sys.executable, "-c", chk], check=True, timeout=120)
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench_train.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=500)
    assert p.returncode != 0
    assert 1 <= p.stderr.count("no GPU visible") <= 2, p.stderr[-2000:]
    assert "{" not in p.stdout


def test_bench_workloads_and_exchanges_parse():
    """`bench.py --workload c4 --exchange all_to_all` (BASELINE configs[3] with the block exchange) is a known command line"""
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--gpus", "8", "--workload", "c4", "--exchange", "all_to_all"])
    assert (a.workload, a.exchange, a.gpus) == ("c4", "all_to_all", 8)
    assert bench.parse_args([]).workload == "c2" and bench.parse_args([]).exchange == "all_gather"
    with pytest.raises(SystemExit):
        bench.parse_args(["--exchange", "ring"])
