"""Rank body for tests/test_bench_launcher.py: what bench.py's ranks do around the model - read the launcher's
environment, join the process group, one collective, rank 0 prints one JSON line.  CPU only (gloo)."""
import json
import os

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
ranks = [None] * world
dist.all_gather_object(ranks, int(os.environ["LOCAL_RANK"]))
if rank == 0:
    print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(t.item()), "local_ranks": ranks,
                      "master": os.environ["MASTER_ADDR"]}), flush=True)
dist.barrier()
dist.destroy_process_group()
