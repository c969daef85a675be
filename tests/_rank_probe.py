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
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import bench
from vitamd.ddp import DataParallel
net = DataParallel(torch.nn.Linear(8, 4))            # CPU module: no probes run, diagnostics() still answers
info = bench.gather_dist_info(None, 10.0 + rank, 0.5 * rank, net.diagnostics())
if rank == 0:
    print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(t.item()), "local_ranks": ranks,
                      "master": os.environ["MASTER_ADDR"], "dist": info}), flush=True)
dist.barrier()
dist.destroy_process_group()
