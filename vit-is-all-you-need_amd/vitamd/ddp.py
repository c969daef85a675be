"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl" on
ROCm) over xGMI.  New functionality — the reference has no distributed code.

Scheme: the global batch is split into contiguous per-rank shards; weights are
replicated (broadcast from rank 0); parameter gradients are packed into flat fp32 buckets in
reverse registration order (= the order backward produces them) and each bucket is all-reduced
asynchronously as soon as its last gradient has been accumulated, so the exchange overlaps the
rest of backward.  `finish()` waits for the outstanding buckets and leaves the averaged gradients in
`p.grad` (views into the buckets: one copy in, none out).

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so ring collectives are per-link bound:
buckets default to ~32 MB (one ViT-B layer = 28.3 MB) — large enough to run at link rate, small
enough that the last bucket's latency after backward ends is < 1 ms.
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.distributed as dist


CLEAN, LOCAL, REDUCING = 0, 1, 2


class SharedDeviceQueuesError(RuntimeError):
    pass


def ranks_share_device(device_ids):
    """device_ids: one (host, device index) per rank (None for ranks that are not on a GPU).  True when two ranks sit on the same card."""
    seen = set()
    for d in device_ids:
        if d is None:
            continue
        if d in seen:
            return True
        seen.add(d)
    return False


def check_hw_queues(device_ids, environ=None):
    """Refuse GPU_MAX_HW_QUEUES together with several ranks on one card.

    HIP maps every stream of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With ONE process per GPU raising it to 8
    is harmless (it also keeps the weight-gradient side stream off the main stream's queue).  With TWO processes on one GPU and 8
    queues each - the rehearsal of the multi-rank path on a one-GPU box - the step hung on both ranks (round 2):
    each rank then asks for 8 compute queues plus its copy queues, more than the card keeps resident at once, so the scheduler
    time-slices the queues; this path's cross-stream dependencies are barrier packets that wait, ON the hardware queue, for a signal
    another queue of the same process must produce, and a waiting queue that holds its hardware slot while the producing queue is
    swapped out never gets its signal.  With the default 4 queues per process every queue stays resident and the same rehearsal runs.
    So the combination is an error here, not a warning."""
    environ = os.environ if environ is None else environ
    try:
        raised = int(environ.get("GPU_MAX_HW_QUEUES") or 0) > 4          # the default (4), spelled out, is the configuration that works
    except ValueError:
        raised = True
    if raised and ranks_share_device(device_ids):
        raise SharedDeviceQueuesError(
            f"GPU_MAX_HW_QUEUES={environ['GPU_MAX_HW_QUEUES']} is set and several ranks share one GPU: this configuration hung in round 2 "
            "(oversubscribed hardware queues under cross-stream barrier packets, DESIGN.md section 7).  Unset GPU_MAX_HW_QUEUES "
            "(the default 4 queues per process work), or run one rank per GPU.")


def select_launch_form(persistent_ms, per_tile_ms, margin=0.01):
    """Pure decision rule of choose_launch_form: persistent NT launches unless one workgroup per tile is faster by more than `margin`
    (relative): a tie keeps the form that is faster when no collective is resident."""
    return not (per_tile_ms < persistent_ms * (1.0 - margin))


def choose_launch_form(device, group=None, measure=None, rows=16384, width=768, iters=4, bucket_mb=28.3):
    """Pick the NT GEMM launch form for a multi-rank job FROM A MEASUREMENT on this job's own devices and fabric, and set it
    (ops.NT_PERSISTENT; functions.tn_target_wgs follows).  A persistent workgroup whose CU is held by a resident communication kernel
    starts late with its whole tile list still to do (one resident foreign workgroup 30.3 -> 35.3 ms/step in the
    persistent form, 31.4 -> 33.0 with one workgroup per tile), so which form wins depends on how long the collectives of a step are
    resident - which only the running job can tell.  Measured here: `iters` x [bucket-sized all-reduce enqueued on the side stream +
    the six NT launches of one layer at (rows, width)] in each form; every rank takes the MAX over ranks, so all ranks decide alike.
    `measure(form) -> ms` replaces the GPU measurement in the CPU unit test.  Returns the record kept as DataParallel.launch_form."""
    from . import ops
    if measure is None:
        measure = lambda form: _measure_form(form, torch.device(device), group, rows, width, iters, bucket_mb)   # noqa: E731
    keep = ops.NT_PERSISTENT
    times, errors = {}, []
    import time as _time
    t_start = _time.perf_counter()
    try:
        for form in (True, False):
            ops.NT_PERSISTENT = form
            # A measurement that FAILS on one rank (out of memory, a HIP error) must not desynchronise the job: the failing rank feeds +inf into
            # the SAME two MAX all-reduces every other rank runs, so all ranks see the failure, keep the default together and go on to the next
            # collective in step (an exception that skipped these all-reduces left the other ranks blocked in them).
            try:
                ms = float(measure(form))
            except Exception as e:          # noqa: BLE001 - any failure of the measurement counts, the decision must still be collective
                ms = float("inf")
                errors.append(f"{type(e).__name__}: {e}")
            t = torch.tensor([ms], dtype=torch.float64)
            if dist.is_initialized() and dist.get_world_size(group) > 1:
                t = t.to(device) if dist.get_backend(group) == "nccl" else t
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            times[form] = float(t.item())
    finally:
        ops.NT_PERSISTENT = keep
    if not all(t < float("inf") for t in times.values()):       # some rank could not measure: every rank keeps the default
        return {"persistent_ms": None, "per_tile_ms": None, "chosen": "persistent" if keep else "per_tile",
                "source": "default (the measurement failed on a rank)", "errors": errors, "calibration_s": round(_time.perf_counter() - t_start, 3)}
    chosen = select_launch_form(times[True], times[False])
    ops.NT_PERSISTENT = chosen
    return {"persistent_ms": round(times[True], 4), "per_tile_ms": round(times[False], 4), "chosen": "persistent" if chosen else "per_tile",
            "source": "measured", "measured_at": {"rows": rows, "width": width},
            "calibration_s": round(_time.perf_counter() - t_start, 3)}      # what the start-up measurement cost this job (reported, not hidden)


def _measure_form(form, device, group, rows, width, iters, bucket_mb):
    """ms per iteration of [all-reduce of one layer bucket on the side stream] beside [the six NT launches of a layer] (the current
    ops.NT_PERSISTENT decides the launch form)"""
    from . import ops, functions
    D = width
    bf = torch.bfloat16
    x1, x3, x4 = (torch.randn(rows, n, device=device).to(bf) for n in (D, 3 * D, 4 * D))
    ws = [torch.randn(n, k, device=device).mul_(0.03).to(bf) for n, k in ((3 * D, D), (4 * D, D), (D, 4 * D), (4 * D, D), (D, 4 * D), (D, 3 * D))]
    res = torch.randn(rows, D, device=device)
    cs = torch.zeros(4 * D, device=device)
    bucket = torch.zeros(int(bucket_mb * (1 << 20) / 4), device=device)
    side = functions.SIDE.stream(device)
    main = torch.cuda.current_stream(device)

    def once():
        ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
        with torch.cuda.stream(side):
            work = dist.all_reduce(bucket, group=group, async_op=True)
        ops.gemm_nt(x1, ws[0], ops.EPI_BIAS_BF16)
        ops.gemm_nt(x1, ws[1], ops.EPI_GELU_DG)
        ops.gemm_nt(x4, ws[2], ops.EPI_RESID_F32, aux=res)
        ops.gemm_nt(x1, ws[3], ops.EPI_DMUL, aux=x4, colsum=cs)
        ops.gemm_nt(x4, ws[4], ops.EPI_BIAS_BF16)
        ops.gemm_nt(x3, ws[5], ops.EPI_BIAS_BF16)
        work.wait()

    once()
    torch.cuda.synchronize(device)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(iters):
        once()
    b.record(main)
    b.synchronize()
    return a.elapsed_time(b) / iters


class _Bucket:
    """One flat fp32 all-reduce unit.  state: CLEAN = no gradient of this step yet; LOCAL = holds this rank's (possibly
    accumulated) gradients, not reduced; REDUCING = its all-reduce has been enqueued (nothing may write it before finish())."""
    __slots__ = ("flat", "params", "offsets", "pending", "work", "launched", "state", "seen")

    def __init__(self, params, device):
        self.params = params
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64  # 256-B aligned slices
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.pending = len(params)
        self.work = None
        self.launched = False          # written in place by the transformer stack's backward this step
        self.state = CLEAN
        self.seen = [False] * len(params)

    def view(self, i):
        p = self.params[i]
        return self.flat[self.offsets[i]: self.offsets[i] + p.numel()].view_as(p)

    def reset(self):
        self.pending = len(self.params)
        self.work = None
        self.launched = False
        self.state = CLEAN
        self.seen = [False] * len(self.params)


class DataParallel(torch.nn.Module):
    """Wraps a module; forward is a pass-through, gradients are averaged across ranks.

        model = DataParallel(ViTClassifier(cfg).cuda())
        loss = loss_fn(model(x_shard), y_shard); loss.backward(); model.finish(); optim.step()

    Contract (checked, not assumed):
      * one backward per finish().  A second backward while buckets are being reduced raises; for gradient accumulation run the
        first micro-batches under `with model.no_sync():` (gradients add up locally, nothing is exchanged) and the last one outside.
      * `zero_grad(set_to_none=False)` on the wrapped module is fine: when a parameter's .grad already aliases its bucket slice the
        stack backward writes into a private arena and autograd adds that into the slice (the in-place fast path needs .grad = None).
      * a parameter that received no gradient this step counts as zero: finish() zeroes its slice and reduces the bucket then.  The
        SET of such parameters must be the same on every rank (collectives are issued in bucket order), as in torch's DDP.
      * every rank must construct its DataParallel wrappers in the same order."""

    def __init__(self, module: torch.nn.Module, bucket_mb: float = 32.0, process_group=None, broadcast: bool = True):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._sync = True
        self.side_overlap_ratio = None      # two-stream / one-stream time of the probe below (multi-rank GPU jobs)
        self.launch_form = None             # record of choose_launch_form (multi-rank GPU jobs)
        if self.world > 1:
            import socket
            p0 = next(module.parameters(), None)
            mine = (socket.gethostname(), p0.device.index) if p0 is not None and p0.is_cuda else None
            ids = [None] * self.world
            dist.all_gather_object(ids, mine, group=process_group)
            check_hw_queues(ids)            # raises on every rank alike
        params = [p for p in module.parameters() if p.requires_grad]
        if broadcast and self.world > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=process_group)
        cap = int(bucket_mb * (1 << 20) / 4)
        self.buckets, cur, size = [], [], 0
        # transformer stacks get one bucket per layer holding the six gradients the kernels' arena has (dWqkv, dbqkv, dW1, db1, dW2, db2),
        # matrices first: their backward writes into the bucket directly and starts the all-reduce per layer (functions.GradSink)
        self._stack_layers = {}
        in_stack = set()
        try:
            from transformer import Transformer
            for m in module.modules():
                if isinstance(m, Transformer) and all(p.requires_grad for p in m.parameters()) and next(m.parameters()).is_cuda:
                    key = tuple(id(p) for layer in m.layers for p in layer._params())
                    layer_buckets = []
                    for layer in m.layers:
                        lp = list(layer._params())          # (Wqkv, bqkv, W1, b1, W2, b2)
                        lp = [lp[0], lp[2], lp[4], lp[1], lp[3], lp[5]]   # bucket order: the three matrices, then the three bias vectors
                        layer_buckets.append(_Bucket(lp, lp[0].device))
                        in_stack.update(id(p) for p in lp)
                    self._stack_layers[key] = layer_buckets
        except ImportError:
            pass
        params = [p for p in params if id(p) not in in_stack]
        for p in reversed(params):  # backward produces gradients roughly in reverse registration order
            if cur and size + p.numel() > cap:
                self.buckets.append(_Bucket(cur, p.device))
                cur, size = [], 0
            cur.append(p)
            size += p.numel()
        if cur:
            self.buckets.append(_Bucket(cur, cur[0].device))
        for lb in self._stack_layers.values():
            self.buckets.extend(reversed(lb))
        from . import functions
        if self._stack_layers:
            functions.GradSink.register(self)
            dev0 = next(iter(self._stack_layers.values()))[0].flat.device
            if self.world > 1 and dev0.type == "cuda" and functions.SIDE.enabled:
                # the weight-gradient side stream must not share a hardware queue with the main stream (whichever of RCCL and the
                # model was set up first decides that: DESIGN.md section 7) - measured here, repaired if necessary
                try:
                    self.side_overlap_ratio = functions.ensure_side_overlap(dev0)
                except Exception as e:      # a failed probe must not stop training: the streams stay as they are
                    import warnings
                    warnings.warn(f"vitamd.ddp: could not probe the side stream ({type(e).__name__}: {e})")
                    self.side_overlap_ratio = 0.0
                if self.side_overlap_ratio > 1.5:
                    import warnings
                    warnings.warn("vitamd.ddp: the weight-gradient side stream is serialised with the main stream (shared hardware queue, "
                                  f"two-stream / one-stream time {self.side_overlap_ratio:.2f}); call vitamd.functions.claim_streams(device) "
                                  "before init_process_group")
            if self.world > 1 and dev0.type == "cuda":
                # NT launch form: an explicit VITAMD_NT_PERSISTENT wins; otherwise measured beside this job's own collectives
                if "VITAMD_NT_PERSISTENT" in os.environ:
                    from . import ops
                    self.launch_form = {"chosen": "persistent" if ops.NT_PERSISTENT else "per_tile", "source": "VITAMD_NT_PERSISTENT"}
                else:
                    try:
                        # measured at the wrapped model's own width where it says so (a ViT: .vit.config / .config -> n_embd), else ViT-B's
                        cfg = getattr(getattr(module, "vit", module), "config", None)
                        width = int(getattr(getattr(cfg, "trans_config", cfg), "n_embd", 768) or 768)
                        self.launch_form = choose_launch_form(dev0, process_group, width=width if width % 64 == 0 else 768)
                    except Exception as e:
                        import warnings
                        warnings.warn(f"vitamd.ddp: launch-form measurement failed ({type(e).__name__}: {e}); keeping the default")
                        self.launch_form = {"chosen": "persistent", "source": "default (measurement failed)"}
        # the broadcast above wrote parameters through .data (no version bump): drop every cached bf16 weight copy
        functions.WEIGHTS.clear()
        self._slot = {}
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._slot[p] = (b, i)
                p.register_post_accumulate_grad_hook(self._on_grad)

    def forward(self, *a, **kw):
        for b in self.buckets:         # a new pass: arrival tracking starts over (bucket CONTENTS and state persist)
            if b.state != REDUCING:
                b.pending = len(b.params)
                b.seen = [False] * len(b.params)
                b.launched = False
        return self.module(*a, **kw)

    @contextlib.contextmanager
    def no_sync(self):
        """gradient accumulation: backward passes inside add into the buckets locally; the first backward outside reduces the sum"""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def diagnostics(self):
        """what a multi-rank run should report next to its throughput (bench.py `dist`)"""
        from . import ops, functions
        return {"side_overlap_ratio": None if self.side_overlap_ratio is None else round(float(self.side_overlap_ratio), 3),
                "nt_launch_form": self.launch_form or {"chosen": "persistent" if ops.NT_PERSISTENT else "per_tile", "source": "default"},
                "tn_target_wgs": functions.tn_target_wgs(), "buckets": len(self.buckets),
                "bucket_mb": round(sum(b.flat.numel() for b in self.buckets) * 4 / (1 << 20), 1)}

    def _reduce(self, b, stream=None):
        """enqueue the averaging all-reduce of bucket b (on `stream` if given)"""
        b.state = REDUCING
        if self.world == 1:
            return
        with torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext():
            b.flat.div_(self.world)  # pre-divide: sum of shares = mean, works on every backend
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    _TWICE = ("vitamd.ddp: backward ran again before finish() - the buckets of the previous backward are still being "
              "all-reduced.  Call finish() after every backward, or run the earlier micro-batches of a gradient-accumulation "
              "step under `with model.no_sync():`")

    # ---- gradient sink interface (called from functions.TransformerStackFn.backward) ----
    def owns(self, params):
        return tuple(id(p) for p in params) in self._stack_layers

    def arena_for(self, params, n_layers):
        """Per-layer gradient views INTO the buckets (the stack backward then writes them in place), or None when that is not
        safe this time - the caller then uses a private arena and the hooks copy it / autograd accumulates it into the buckets."""
        lb = self._stack_layers.get(tuple(id(p) for p in params))
        if lb is None or len(lb) != n_layers:
            return None
        if any(b.state == REDUCING for b in lb):
            raise RuntimeError(self._TWICE)
        if not self._sync or any(b.state != CLEAN for b in lb) or any(p.grad is not None for b in lb for p in b.params):
            return None               # accumulation, or .grad already aliases the bucket (zero_grad(set_to_none=False))
        out = []
        for b in lb:
            b.flat[b.offsets[3]:].zero_()     # the kernels OVERWRITE the weight gradients and accumulate (atomics / column sums) only into the bias gradients at the bucket's tail
            b.launched = True
            out.append((b.view(0), b.view(3), b.view(1), b.view(4), b.view(2), b.view(5)))   # back in (dWqkv, dbqkv, dW1, db1, dW2, db2) order
        return out

    def layer_ready(self, params, i):
        """every gradient of layer i is enqueued (input-gradient chain on the current stream, weight
        gradients on the side stream): reduce its bucket on the side stream, behind both"""
        b = self._stack_layers[tuple(id(p) for p in params)][i]
        if not b.launched:            # this backward went through the private arena: the hooks drive the bucket
            return
        if self.world == 1:
            b.state = REDUCING
            return
        from .functions import SIDE
        main = torch.cuda.current_stream()
        side = SIDE.stream(b.flat.device) if SIDE.enabled else main
        if side is not main:
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
        self._reduce(b, side if side is not main else None)

    def _on_grad(self, p):
        b, i = self._slot[p]
        v = b.view(i)
        if b.launched:
            p.grad = v           # already written in place (and possibly already being reduced)
            b.seen[i] = True
            return
        if b.state == REDUCING:
            raise RuntimeError(self._TWICE)
        if p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)      # .grad was None before this pass (fresh tensor): whatever the slice held is stale
            p.grad = v           # the bucket slice IS the gradient from here on; later passes accumulate into it in place
        if not b.seen[i]:
            b.seen[i] = True
            b.pending -= 1
        b.state = LOCAL
        if b.pending == 0 and self._sync:
            self._reduce(b)

    def finish(self):
        """Wait for every outstanding bucket (call after backward, before the optimiser step).  Buckets that hold gradients
        but were not launched (some parameter got no gradient this step) are reduced here, in bucket order, with the slices of
        parameters whose .grad is None counted as zero.  Under no_sync() this only keeps the local sums."""
        if self._sync:
            for b in self.buckets:
                if b.state == LOCAL:
                    for i, p in enumerate(b.params):
                        if p.grad is None:
                            b.view(i).zero_()
                    self._reduce(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
            if b.state == REDUCING:
                b.reset()

    def zero_grad(self, set_to_none: bool = True):
        # gradients live in the buckets; dropping the references is enough (hooks re-attach views)
        self.module.zero_grad(set_to_none=True)


def shard_batch(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of a global batch for `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
