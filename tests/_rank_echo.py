"""Child of tests/test_launch.py: one gloo rank started by mpgan_amd.launch.spawn_ranks.
argv: [fail_rank] -- that rank exits with code 3 before the rendezvous completes its work."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"]
assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
fail_rank = int(sys.argv[1]) if len(sys.argv) > 1 else -1
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
dist.barrier()
if rank == fail_rank:
    sys.exit(3)
print("rank %d noise on stdout" % rank)
if rank == 0:
    print(json.dumps({"n_gpus": world, "sum": float(t.item())}))
dist.destroy_process_group()
