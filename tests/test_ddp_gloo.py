"""world_size-2 rehearsal of the data-parallel path on CPU (gloo): averaged gradients of the
sharded batch must equal the single-process gradients of the whole batch, buckets must fire in
backward order, and a second step must reuse the bucket views."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.GELU(),
                               torch.nn.Linear(64, 8))


def _worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "vit-is-all-you-need_amd"))
    from vitamd.ddp import DataParallel, shard_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)          # deliberately different init: broadcast must fix it
    m = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.GELU(),
                            torch.nn.Linear(64, 8))
    if rank == 0:
        m.load_state_dict(_model().state_dict())
    ddp = DataParallel(m, bucket_mb=0.01)   # tiny buckets -> several of them
    assert len(ddp.buckets) >= 3
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(16, 32, generator=g), torch.randn(16, 8, generator=g)
    lo, hi = shard_batch(16, rank, world)
    out = []
    for step in range(2):
        ddp.zero_grad()
        loss = ((ddp(x[lo:hi]) - y[lo:hi]) ** 2).mean()
        loss.backward()
        ddp.finish()
        out.append([p.grad.detach().numpy().copy() for p in m.parameters()])   # plain arrays: no fd passing through the queue
        in_bucket = all(p.grad.data_ptr() == ddp._slot[p][0].view(ddp._slot[p][1]).data_ptr() for p in m.parameters())
        assert in_bucket
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_world2_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _model()
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(16, 32, generator=g), torch.randn(16, 8, generator=g)
    ((ref(x) - y) ** 2).mean().backward()
    for step in range(2):
        for r in range(world):
            for got, p in zip(results[r][step], ref.parameters()):
                assert torch.allclose(torch.from_numpy(got), p.grad, rtol=1e-5, atol=1e-6)
