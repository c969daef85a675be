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


# ------------------------------------------------------------------------------------------------------------------------------
# bucket state machine: accumulation, set_to_none=False, parameters without gradient, misuse
class _Branchy(torch.nn.Module):
    """two heads; `use_b=False` leaves head_b without gradient for the step"""

    def __init__(self):
        super().__init__()
        self.body = torch.nn.Linear(16, 32)
        self.head_a = torch.nn.Linear(32, 4)
        self.head_b = torch.nn.Linear(32, 4)

    def forward(self, x, use_b=True):
        h = torch.nn.functional.gelu(self.body(x))
        return self.head_a(h) + (self.head_b(h) if use_b else 0.0)


def _state_worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "vit-is-all-you-need_amd"))
    from vitamd.ddp import DataParallel, shard_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    m = _Branchy()
    ddp = DataParallel(m, bucket_mb=0.002)
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(16, 16, generator=g), torch.randn(16, 4, generator=g)
    lo, hi = shard_batch(16, rank, world)
    mid = (lo + hi) // 2
    out = {}

    def grads():
        return [None if p.grad is None else p.grad.detach().numpy().copy() for p in m.parameters()]

    # 1. gradient accumulation: first micro-batch under no_sync, second outside; mean-of-sums convention = sum of the two micro losses
    ddp.zero_grad()
    with ddp.no_sync():
        ((ddp(x[lo:mid]) - y[lo:mid]) ** 2).sum().backward()
        ddp.finish()                               # a finish() inside no_sync keeps the local sums
    ((ddp(x[mid:hi]) - y[mid:hi]) ** 2).sum().backward()
    ddp.finish()
    out["accum"] = grads()
    # 2. zero_grad(set_to_none=False) on the wrapped module: .grad stays the (zeroed) bucket view
    m.zero_grad(set_to_none=False)
    ((ddp(x[lo:hi]) - y[lo:hi]) ** 2).sum().backward()
    ddp.finish()
    out["keep_views"] = grads()
    # 3. a parameter without gradient this step (same on every rank): its bucket is still reduced, .grad stays None
    ddp.zero_grad()
    ((ddp(x[lo:hi], use_b=False) - y[lo:hi]) ** 2).sum().backward()
    ddp.finish()
    out["unused"] = grads()
    # 4. misuse: a second backward while the first one's buckets are being reduced
    ddp.zero_grad()
    ((ddp(x[lo:hi]) - y[lo:hi]) ** 2).sum().backward()
    try:
        ((ddp(x[lo:hi]) - y[lo:hi]) ** 2).sum().backward()
        out["twice"] = "no error"
    except RuntimeError as e:
        out["twice"] = "raised" if "finish()" in str(e) else f"other: {e}"
    ddp.finish()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_bucket_state_machine_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_state_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    ref = _Branchy()
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(16, 16, generator=g), torch.randn(16, 4, generator=g)

    def ref_grads(use_b=True):
        ref.zero_grad(set_to_none=True)
        (((ref(x, use_b) - y) ** 2).sum() / world).backward()      # ranks average their per-shard SUM losses
        return [None if p.grad is None else p.grad for p in ref.parameters()]

    full, partial = ref_grads(True), ref_grads(False)
    for r in range(world):
        for case, want in (("accum", full), ("keep_views", full), ("unused", partial)):
            for got, w in zip(results[r][case], want):
                if w is None:
                    assert got is None, case
                else:
                    assert got is not None and torch.allclose(torch.from_numpy(got), w, rtol=1e-5, atol=1e-6), (r, case)
        assert results[r]["twice"] == "raised", results[r]["twice"]


# ---------------------------------------------------------------- run-time launch form and the hardware-queue guard
def test_select_launch_form_rule():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-is-all-you-need_amd"))
    from vitamd import ddp
    assert ddp.select_launch_form(30.3, 31.4) is True          # persistent faster: keep it
    assert ddp.select_launch_form(35.3, 33.0) is False         # resident collectives hurt the persistent form more: one workgroup per tile
    assert ddp.select_launch_form(30.00, 29.90) is True        # inside the 1 % margin: a tie keeps the persistent form


def test_choose_launch_form_with_stub_timer_sets_runtime_switches():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-is-all-you-need_amd"))
    from vitamd import ddp, ops, functions
    keep = ops.NT_PERSISTENT
    try:
        seen = []
        rec = ddp.choose_launch_form("cpu", measure=lambda form: (seen.append((form, ops.NT_PERSISTENT)), 35.3 if form else 33.0)[1])
        assert seen == [(True, True), (False, False)]            # each form is measured with the switch really set
        assert rec["chosen"] == "per_tile" and rec["persistent_ms"] == 35.3 and rec["per_tile_ms"] == 33.0 and rec["source"] == "measured"
        assert ops.NT_PERSISTENT is False and functions.tn_target_wgs() == 128      # the weight-gradient cut follows at run time
        rec = ddp.choose_launch_form("cpu", measure=lambda form: 30.3 if form else 31.4)
        assert rec["chosen"] == "persistent" and ops.NT_PERSISTENT is True and functions.tn_target_wgs() == 252
        functions.TN_TARGET_WGS = 192                                # an explicit value overrides the rule
        assert functions.tn_target_wgs() == 192
    finally:
        ops.NT_PERSISTENT = keep
        functions.TN_TARGET_WGS = None


def _form_worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "vit-is-all-you-need_amd"))
    from vitamd import ddp, ops
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keep = ops.NT_PERSISTENT

    def measure(form):                       # rank 1 cannot measure the first form (an out-of-memory, say); rank 0 would pick per_tile
        if rank == 1 and form:
            raise RuntimeError("HIP out of memory (stub)")
        return 35.3 if form else 33.0

    rec = ddp.choose_launch_form("cpu", measure=measure)
    t = torch.tensor([float(rank + 1)])      # the NEXT collective of the job: both ranks must still be in step
    dist.all_reduce(t)
    q.put((rank, rec["source"], rec["chosen"], ops.NT_PERSISTENT == keep, float(t.item()), len(rec.get("errors", []))))
    dist.barrier()
    dist.destroy_process_group()


def test_choose_launch_form_survives_a_failing_rank_world2():
    """a measurement that fails on ONE rank must not leave the others blocked in the MAX all-reduces or with a different launch form:
    the failing rank feeds +inf into the same collectives, every rank keeps the default, and the job's next collective still matches."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_form_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict((r[0], r[1:]) for r in (q.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        source, chosen, kept, total, nerr = results[r]
        assert source.startswith("default") and kept and total == 3.0, results
    assert results[0][1] == results[1][1] and results[1][4] == 1 and results[0][4] == 0


def test_hw_queue_guard():
    import sys
    import pytest
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-is-all-you-need_amd"))
    from vitamd import ddp
    shared, apart = [("box", 0), ("box", 0)], [("box", 0), ("box", 1)]
    ddp.check_hw_queues(shared, environ={})                          # default queue count: fine
    ddp.check_hw_queues(shared, environ={"GPU_MAX_HW_QUEUES": "4"})  # ... also when spelled out
    ddp.check_hw_queues(apart, environ={"GPU_MAX_HW_QUEUES": "8"})   # one rank per GPU: fine
    ddp.check_hw_queues([None, None], environ={"GPU_MAX_HW_QUEUES": "8"})   # CPU ranks
    with pytest.raises(ddp.SharedDeviceQueuesError, match="share one GPU"):
        ddp.check_hw_queues(shared, environ={"GPU_MAX_HW_QUEUES": "8"})
