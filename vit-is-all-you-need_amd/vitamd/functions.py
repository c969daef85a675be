"""autograd.Function glue: each Function strings libvitamd kernels together for one reference
module's forward and hand-written backward.  dtype flow = the reference's autocast flow with bf16
as the low-precision type: fp32 residual stream and LayerNorm, bf16 GEMM /
attention / GELU operands and outputs, fp32 accumulation, fp32 parameter gradients.
"""
from __future__ import annotations

import weakref

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32

# The reference's loops call the model under torch.autocast("cuda") with a GradScaler (train_vit.py:84,100-106).  Our Functions own their
# precision flow (fp32 masters and residual stream, bf16 kernels): inside an autocast region every floating tensor argument is taken as
# fp32 and autocast is switched off for the body, so the modules compute exactly what they compute outside of it.
_amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")


class WeightCache:
    """bf16 (and transposed bf16) copies of fp32 parameters — autocast's per-step weight cast,
    reference train_vit.py:100.  An entry is fresh while the parameter's version counter, storage
    and the cache epoch are unchanged (`clear()` bumps the epoch: every optimiser step / bench step).
    Entries are tied to the parameter OBJECT (weak reference), never to its address alone, and keep
    their bf16 buffers across refreshes.  `prepare()` refreshes a whole list of weights with ONE
    batched kernel launch instead of one launch per tensor."""

    def __init__(self):
        self._c = {}
        self._groups = {}
        self.epoch = 0

    def clear(self):
        self.epoch += 1

    def _fresh(self, ent, w, want_t):
        return (ent is not None and ent["ref"]() is w and ent["ver"] == w._version and ent["ptr"] == w.data_ptr()
                and ent["epoch"] == self.epoch and (ent["wbt"] is not None or not want_t))

    @staticmethod
    def _as2d(w):
        w2 = w.detach()
        return w2 if w2.dim() == 2 else w2.reshape(w2.shape[0], -1)

    def _entry(self, w, want_t):
        """entry with buffers allocated (contents possibly stale)"""
        ent = self._c.get(id(w))
        w2 = self._as2d(w)
        if ent is None or ent["ref"]() is not w or ent["ptr"] != w.data_ptr() or tuple(ent["wb"].shape) != tuple(w2.shape):
            if len(self._c) > 4096:
                self._c = {k: v for k, v in self._c.items() if v["ref"]() is not None}
                self._groups.clear()
            ent = self._c[id(w)] = {"ref": weakref.ref(w), "ver": -1, "ptr": w.data_ptr(), "epoch": -1,
                                    "wb": torch.empty(w2.shape, dtype=BF16, device=w.device), "wbt": None}
        if want_t and ent["wbt"] is None:
            ent["wbt"] = torch.empty((w2.shape[1], w2.shape[0]), dtype=BF16, device=w.device)
            ent["ver"] = -1
        return ent

    def get(self, w: torch.Tensor, want_t: bool):
        if not w.is_cuda:
            raise ops._lib.VitamdError("weight: expected a ROCm device tensor (the HIP kernels are the only implementation)")
        ent = self._c.get(id(w))
        if self._fresh(ent, w, want_t):
            return ent["wb"], ent["wbt"]
        ent = self._entry(w, want_t)
        w2 = self._as2d(w).contiguous()
        N, K = w2.shape
        ops._lib.check(ops._L().vitamd_cast_transpose_weight(w2.data_ptr(), ent["wb"].data_ptr(),
                                                             ent["wbt"].data_ptr() if ent["wbt"] is not None else None, N, K,
                                                             ops._stream()), "cast_transpose_weight")
        ent["ver"], ent["epoch"] = w._version, self.epoch
        return ent["wb"], ent["wbt"]

    def prepare(self, weights, want_t: bool):
        """make every weight in the list fresh with one batched launch (no-op when they all are)"""
        for w in weights:
            ops._need(w.detach(), F32, "weight")          # device / dtype / contiguity: fail loudly, no fallback
        if all(self._fresh(self._c.get(id(w)), w, want_t) for w in weights):
            return
        import numpy as np
        ents = [self._entry(w, want_t) for w in weights]
        key = (tuple(id(w) for w in weights), want_t)
        grp = self._groups.get(key)
        sig = tuple((e["ptr"], e["wb"].data_ptr(), e["wbt"].data_ptr() if e["wbt"] is not None else 0) for e in ents)
        if grp is None or grp["sig"] != sig or any(not w.is_contiguous() for w in weights):
            if any(not w.is_contiguous() for w in weights):
                for w in weights:
                    self.get(w, want_t)
                return
            dt = np.dtype([("w", "<u8"), ("wb", "<u8"), ("wbt", "<u8"), ("N", "<i4"), ("K", "<i4"), ("first", "<i4"), ("tk", "<i4")])
            tab = np.zeros(len(weights), dtype=dt)
            first = 0
            for i, (w, e) in enumerate(zip(weights, ents)):
                N, K = e["wb"].shape
                tab[i] = (e["ptr"], sig[i][1], sig[i][2], N, K, first, (K + 63) // 64)
                first += ((N + 63) // 64) * ((K + 63) // 64)
            dev_tab = torch.from_numpy(tab.view(np.uint8).copy()).to(weights[0].device)
            grp = self._groups[key] = {"sig": sig, "table": dev_tab, "tiles": first}
        ops._lib.check(ops._L().vitamd_cast_transpose_batched(grp["table"].data_ptr(), len(weights), grp["tiles"], ops._stream()),
                       "cast_transpose_batched")
        for w, e in zip(weights, ents):
            e["ver"], e["epoch"] = w._version, self.epoch


WEIGHTS = WeightCache()


ATTN_FUSED_RESID = False  # A/B knob: x1 = x0 + attention written by the attention forward kernel (N <= 256): correct, measured EQUAL (34.47 vs 34.43 ms)
TN_TARGET_WGS = None   # None = follow the NT launch form (252 beside persistent NT launches, 128 beside one workgroup per tile), read on every call; an int overrides.  Workgroups per layer weight-gradient GEMM (tiles x split-K factor); 0 = the kernel's own rule (~256 = every CU).  Whole-step A/B (tools/ab_splits.py) with the PERSISTENT NT launches: 96 -> 31.21 ms, 128 -> 30.86, 160 -> 30.79, 192 -> 30.55, 216 -> 30.44, 252 -> 30.3, 288 -> 31.0, 504 -> 31.8 (with one NT workgroup per tile the optimum was 128-144: 31.45 vs 32.26 at 252)


TN_FORM_POLICY = "auto"   # which weight-gradient kernel (ops.TN_FORM_*; same results).  "auto": the 12-wave EXCLUSIVE form for the fc2 weight gradient -
# launched at the start of a layer's backward, beside the two input-gradient GEMMs of the MLP, whose workgroups fill their CUs anyway - and for every weight
# gradient when there is no second stream; the 8-wave SHARED form elsewhere (the fc1 / QKV weight gradients run beside LayerNorm backward, whose waves share
# CUs with it).  Whole-step A/B (tools/ab_dbg.py, profiles/r03/ab_tn_loader_*.log): fc2 only -0.10 / -0.36 ms on two boxes; all three +0.46 ms; fc1 or QKV
# alone +0.26 / +0.29; without the second stream all three -0.62 ms.  "shared" / "exclusive" force one form.


def _tn_form(which):
    if TN_FORM_POLICY == "shared":
        return ops.TN_FORM_SHARED
    if TN_FORM_POLICY == "exclusive" or not SIDE.enabled:
        return ops.TN_FORM_EXCLUSIVE
    return ops.TN_FORM_EXCLUSIVE if which == "fc2" else ops.TN_FORM_SHARED


def _tn_splits(dW):
    """Split-K factor for a weight-gradient GEMM that runs on the side stream BESIDE the input-gradient chain: a little under one
    workgroup per CU (180 of 256) - fewer, longer workgroups write fewer fp32 partial tiles for the reduce pass, and the CUs they
    leave free are taken by the main stream's kernels anyway.  Whole-step A/B (tools/ab_splits.py): 5 splits instead of 7 on the
    36-tile GEMMs of ViT-B: -0.2 ms/step; 3 or fewer lose (36.9 ms at 3, 46 ms at 2)."""
    target = tn_target_wgs()
    if not target or not SIDE.enabled:      # alone on the chip (no side stream) the kernel's own rule - every CU - is right
        return 0
    ntile = ((dW.shape[0] + 255) // 256) * ((dW.shape[1] + 255) // 256)
    return max(1, round(target / ntile))


def tn_target_wgs():
    """workgroups a layer's weight-gradient GEMM is cut into: the explicit TN_TARGET_WGS, else the value that goes with the current
    NT launch form (ops.NT_PERSISTENT, a run-time switch)"""
    if TN_TARGET_WGS is not None:
        return TN_TARGET_WGS
    return 252 if ops.NT_PERSISTENT else 128


SIDE_POLICY = 0           # A/B knob (tools/ab_side.py): when the MLP weight-gradient GEMMs enter the side stream: 0 = as soon as their inputs exist (beside the
                          # input-gradient GEMMs), 1 = both after dgrad-fc1 (beside LayerNorm / attention backward), 2 = dW2 beside dgrad-fc1, dW1 after it
LN_BWD_XHAT = True        # A/B knob (tools/ab_gelu.py): LayerNorm backward reads xhat from the saved bf16 LN output instead of recomputing it from fp32 x
GELU_STORED_GRAD = True   # A/B knob (tools/ab_gelu.py); False = keep the pre-activation and evaluate gelu' in the backward
DEFER_RESID = True        # inside a stack: fc2 writes bf16 y and the NEXT layer's first LayerNorm adds it to the residual stream (x2 = x1 + y, the same
                          # fp32 + bf16 sum the fused epilogue formed: bit-identical).  The same HBM bytes in total, but they move from the GEMM's
                          # epilogue (40 fp32 loads + 40 fp32 stores per wave and tile on the CU's one path to L1, matrix pipes idle) into an
                          # HBM-bound LayerNorm kernel, and fc2 becomes a plain bias GEMM: whole-step A/B (tools/ab_flags.py) in DESIGN.md section 4.2


def streams_overlap(device, side=None):
    """Measure whether kernels on `side` (default: the weight-gradient side stream) run beside kernels on the current stream: time of
    two few-workgroup GEMMs, one per stream, over the time of one.  ~1.0 = concurrent, ~2.0 = the two streams share a hardware
    queue and are serialised (see claim_streams)."""
    device = torch.device(device)
    side = side if side is not None else SIDE.stream(device)
    R, P = 16384, 512
    l = torch.ones(R, P, device=device, dtype=torch.bfloat16)
    o1, o2 = torch.empty(P, P, device=device), torch.empty(P, P, device=device)
    cur = torch.cuda.current_stream(device)

    def run(both):
        cur.synchronize(); side.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(cur)
        if both:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(3):
                    ops.gemm_tn(l, l, o2, accumulate=False, splits=1)
        for _ in range(3):
            ops.gemm_tn(l, l, o1, accumulate=False, splits=1)
        if both:
            cur.wait_stream(side)
        b.record(cur)
        b.synchronize()
        return a.elapsed_time(b)

    run(True)
    one = min(run(False) for _ in range(2))
    two = min(run(True) for _ in range(2))
    return two / max(one, 1e-6)


def ensure_side_overlap(device, attempts=6):
    """Make sure the side stream does not share a hardware queue with the current stream: probe it (streams_overlap) and, if the two
    are serialised, replace it by fresh streams until one runs concurrently.  DataParallel calls this for multi-rank jobs, so the order
    in which the caller initialised RCCL and first used the model does not matter.  Returns the final ratio."""
    device = _indexed(device)
    ratio = streams_overlap(device)
    kept = []                          # candidates that collided stay alive until the end, so the next one gets another queue
    while ratio > 1.5 and attempts > 0:
        kept.append(SIDE._streams[device])
        SIDE._streams[device] = torch.cuda.Stream(device=device)
        ratio = streams_overlap(device)
        attempts -= 1
    return ratio


def _f32c(t):
    return t.detach().to(F32).contiguous()


# ------------------------------------------------------------------------------------------------
# one pre-LN transformer layer (reference transformer.py:42-45)
# ------------------------------------------------------------------------------------------------
def new_seed():
    """64-bit dropout seed from torch's CPU generator (so torch.manual_seed makes runs repeatable)"""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def layer_forward(x0, wqkv, bqkv, w1, b1, w2, b2, B, N, H, causal, need_grad, p_attn=0.0, p_mlp=0.0, pending=None, defer=False):
    """x0 fp32 [M,D] -> x2 fp32 [M,D], the tensors backward needs, and the dropout record
    (p_attn, seed_attn, p_mlp, seed_mlp).  p_attn: SDPA dropout_p (transformer.py:28); p_mlp: the
    nn.Dropout after fc2 (transformer.py:40).
    pending: bf16 [M,D] MLP output of the layer below that has not been added to x0 yet (this layer's first LayerNorm adds it).
    defer: return (x1, y) with y = bf16 fc2 output instead of x2 = x1 + y (the caller hands y to the next layer as `pending`)."""
    drop = (p_attn, new_seed() if p_attn > 0 else 0, p_mlp, new_seed() if p_mlp > 0 else 0)
    wqkv_b, _ = WEIGHTS.get(wqkv, need_grad)
    w1_b, _ = WEIGHTS.get(w1, need_grad)
    w2_b, _ = WEIGHTS.get(w2, need_grad)
    x0, a, mean1, rstd1 = ops.layernorm_fwd(x0, addend=pending)                  # (residual of the layer below +) LN1   transformer.py:43-44
    qkv = ops.gemm_nt(a, wqkv_b, ops.EPI_BIAS_BF16, bias=bqkv)                   # fused QKV      transformer.py:27
    if ATTN_FUSED_RESID and N <= ops.ATTN_RESID_MAX_N:
        # SDPA (transformer.py:28-29) with the residual add of transformer.py:44 in its epilogue: the LayerNorm below then reads the
        # fp32 stream once (6 B/element) instead of reading x0 and o and writing x1 (12 B/element)
        o, lse, x1 = ops.attention_fwd(qkv, B, N, H, causal, dropout=drop[:2], resid=x0)
        _, bln, mean2, rstd2 = ops.layernorm_fwd(x1)                             # LN2            transformer.py:43
    else:
        o, lse = ops.attention_fwd(qkv, B, N, H, causal, dropout=drop[:2])       # SDPA           transformer.py:28-29
        x1, bln, mean2, rstd2 = ops.layernorm_fwd(x0, addend=o)                  # residual + LN2 transformer.py:43-44
    # fc1 + GELU (transformer.py:37-38); `pre` holds bf16(gelu'(fc1 out)) for the backward - the derivative is evaluated
    # here, where its exp is shared with the erf and the VALU work hides under the output stores (-85 us per layer in dgrad fc2)
    pre, h = ops.gemm_nt(bln, w1_b, ops.EPI_GELU_DG if GELU_STORED_GRAD else ops.EPI_GELU, bias=b1)
    y = None
    if p_mlp > 0:
        x2 = ops.linear_dropout_resid(h, w2_b, b2, x1, drop[2:])                 # fc2 + dropout + residual
    elif defer:
        x2, y = x1, ops.gemm_nt(h, w2_b, ops.EPI_BIAS_BF16, bias=b2)             # fc2; x2 = x1 + y is formed by the next layer's LayerNorm
    else:
        x2 = ops.gemm_nt(h, w2_b, ops.EPI_RESID_F32, bias=b2, aux=x1)            # fc2 + residual transformer.py:39,44
    saved = (x0, mean1, rstd1, a, qkv, o, lse, x1, mean2, rstd2, bln, pre, h) if need_grad else None
    return (x2, y) if defer else x2, saved, drop


def _indexed(device):
    """torch.device with an explicit index ("cuda" -> the current device), so that streams claimed for "cuda" and looked up by a
    tensor's device (cuda:0) are the same entry"""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


class _Side:
    """A second HIP stream for the weight-gradient GEMMs and bias-gradient column sums.  They
    depend on the input-gradient chain but nothing in that chain depends on them, so running them
    beside it lets their workgroups fill the CUs a GEMM's last partial round leaves idle (e.g.
    591 tiles on 256 CUs = 2.31 rounds) and hides each kernel's prologue/epilogue under the other."""

    def __init__(self):
        self._streams = {}
        self.enabled = True

    def stream(self, device):
        device = _indexed(device)
        s = self._streams.get(device)
        if s is None:
            s = self._streams[device] = torch.cuda.Stream(device=device)
        return s


SIDE = _Side()


def claim_streams(device):
    """Create the side stream and run a first (empty) kernel on it.  Call this BEFORE torch.distributed.init_process_group("nccl")
    in a multi-GPU job: HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in order of first use, and with
    RCCL's streams created first the side stream was measured to land on the main stream's queue - the two are then serialised
    with barrier packets, 35.5 instead of 31.4 ms/step on every rank (tools/ddp_bisect.py).  (Do NOT raise GPU_MAX_HW_QUEUES instead
    when ranks share a device: see vitamd.ddp.check_hw_queues.)"""
    device = _indexed(device)
    with torch.cuda.stream(SIDE.stream(device)):
        torch.zeros(16, device=device).add_(1)
    torch.cuda.current_stream(device).synchronize()


class GradSink:
    """Optional hook for data-parallel training (vitamd/ddp.py registers one per DataParallel wrapper): lets the transformer
    stack's backward write each layer's parameter gradients straight into that layer's all-reduce
    bucket and start the bucket's all-reduce the moment the layer is finished, instead of after the
    whole stack (which is a single autograd node) has returned."""
    _sinks = []          # weak references: a wrapper that is garbage-collected drops out by itself

    @classmethod
    def register(cls, sink):
        cls._sinks = [r for r in cls._sinks if r() is not None and r() is not sink]
        cls._sinks.append(weakref.ref(sink))

    @classmethod
    def find(cls, params):
        """the registered sink that owns exactly this transformer stack's parameters, or None"""
        if params is None:
            return None
        for r in cls._sinks:
            s = r()
            if s is not None and s.owns(params):
                return s
        return None


def layer_sizes(D):
    return (3 * D * D, 3 * D, 4 * D * D, 4 * D, 4 * D * D, D), ((3 * D, D), (3 * D,), (4 * D, D), (4 * D,), (D, 4 * D), (D,))


def grad_arena(D, n_layers, device, params=None):
    """fp32 storage for every parameter gradient of `n_layers` layers; returns per-layer
    (dWqkv, dbqkv, dW1, db1, dW2, db2) views (bias gradients zeroed).  Two buffers by default; the gradient
    sink's per-layer buckets when one is installed for these parameters."""
    sizes, shapes = layer_sizes(D)
    sink = GradSink.find(params)
    if sink is not None:
        got = sink.arena_for(params, n_layers)
        if got is not None:
            return got, sink
    # weight gradients are OVERWRITTEN by the split-K reduce pass, only the bias gradients are accumulated into (atomics / column
    # sums): the matrices live in an uninitialised buffer, the vectors in a small zeroed one (a 9-KB memset instead of 340 MB)
    per_w, per_b = sum(sizes[0::2]), sum(sizes[1::2])
    flat_w = torch.empty(per_w * n_layers, dtype=F32, device=device)
    flat_b = torch.zeros(per_b * n_layers, dtype=F32, device=device)
    out = []
    for i in range(n_layers):
        ow, ob, views = i * per_w, i * per_b, []
        for j, (n, sh) in enumerate(zip(sizes, shapes)):
            if j % 2 == 0:
                views.append(flat_w[ow: ow + n].view(sh))
                ow += n
            else:
                views.append(flat_b[ob: ob + n].view(sh))
                ob += n
        out.append(tuple(views))
    return out, None


def layer_backward(g2, saved, wqkv, w1, w2, B, N, H, causal, grads, dy2=None, have_db2=False, emit_bf16=False,
                   emit_colsum=None, drop=(0.0, 0, 0.0, 0), emit_dropout=(0.0, 0)):
    """g2 fp32 [M,D] = dL/dx2.  Fills `grads` = (dWqkv, dbqkv, dW1, db1, dW2, db2) (zero-initialised
    fp32, accumulated into) and returns (g0, bf16(g0) or None).
    dy2: bf16(g2) if a previous kernel already produced it (then db2 is already in grads[5] when
    have_db2).  emit_bf16/emit_colsum: also produce bf16(g0) and add its column sums to emit_colsum
    (the fc2 bias gradient of the layer below); emit_dropout = that layer's fc2 dropout (p, seed), whose
    mask the emitted copy must carry.  drop = this layer's dropout record from layer_forward."""
    x0, mean1, rstd1, a, qkv, o, lse, x1, mean2, rstd2, bln, pre, h = saved
    dWqkv, dbqkv, dW1, db1, dW2, db2 = grads
    main = torch.cuda.current_stream()
    side = SIDE.stream(g2.device) if SIDE.enabled else main

    def on_side(fn, *tensors):
        """run fn on the side stream after everything enqueued so far on the main stream"""
        if side is main:
            fn()
            return
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            fn()
        # The side stream reads / writes these main-stream allocations.  They are kept alive until join_side() has ENQUEUED the main
        # stream's wait on the side stream: whatever the allocator hands out after that point is ordered behind the side work on
        # the GPU, so the blocks go back to the main-stream pool as ordinary frees.  (tensor.record_stream() is the textbook tool,
        # but it defers each reuse until the side stream has really passed the free - with the host running several steps ahead of
        # the GPU every step then needed fresh hipMallocs: 22 GiB of live data became 68 GiB reserved for ViT-B, 155 GiB for ViT-L.)
        _SIDE_KEEP.extend(tensors)

    _, wqkv_t = WEIGHTS.get(wqkv, True)
    _, w1_t = WEIGHTS.get(w1, True)
    _, w2_t = WEIGHTS.get(w2, True)
    if dy2 is None:
        dy2 = ops.cast_bf16_dropout(g2, drop[2:]) if drop[2] > 0 else ops.cast_bf16(g2)
    # ---- MLP
    def wgrad_fc2():
        ops.gemm_tn(dy2, h, dW2, accumulate=False, splits=_tn_splits(dW2), form=_tn_form("fc2"))
        if not have_db2:
            ops.colsum(dy2, db2)
    if SIDE_POLICY == 0:
        on_side(wgrad_fc2, dy2, h, dW2, db2)
    dpre = ops.gemm_nt(dy2, w2_t, ops.EPI_DMUL if GELU_STORED_GRAD else ops.EPI_DGELU, aux=pre, colsum=db1)   # dgrad fc2 . gelu'
    if SIDE_POLICY == 0:
        on_side(lambda: ops.gemm_tn(dpre, bln, dW1, accumulate=False, splits=_tn_splits(dW1), form=_tn_form("fc1")), dpre, bln, dW1)
    if SIDE_POLICY == 2:
        on_side(wgrad_fc2, dy2, h, dW2, db2)
    dbln = ops.gemm_nt(dpre, w1_t, ops.EPI_BIAS_BF16)                            # dgrad fc1
    if SIDE_POLICY == 1:
        on_side(wgrad_fc2, dy2, h, dW2, db2)
    if SIDE_POLICY in (1, 2):
        on_side(lambda: ops.gemm_tn(dpre, bln, dW1, accumulate=False, splits=_tn_splits(dW1), form=_tn_form("fc1")), dpre, bln, dW1)
    g1, d_o = ops.layernorm_bwd(dbln, x1, mean2, rstd2, g_res=g2, want_bf16=True, xhat=bln if LN_BWD_XHAT else None)
    # ---- attention
    dqkv = ops.attention_bwd(qkv, o, lse, d_o, B, N, H, causal, dbias=dbqkv, dropout=drop[:2])   # also adds the QKV bias gradient
    on_side(lambda: ops.gemm_tn(dqkv, a, dWqkv, accumulate=False, splits=_tn_splits(dWqkv), form=_tn_form("qkv")), dqkv, a, dWqkv)
    da = ops.gemm_nt(dqkv, wqkv_t, ops.EPI_BIAS_BF16)                            # dgrad qkv
    g0, g0b = ops.layernorm_bwd(da, x0, mean1, rstd1, g_res=g1, want_bf16=emit_bf16, colsum=emit_colsum, dropout=emit_dropout,
                                xhat=a if LN_BWD_XHAT else None)
    return g0, g0b


_SIDE_KEEP = []     # tensors in use by side-stream work that the main stream has not been ordered behind yet (current layer)
_SIDE_DONE = []     # (side-stream event, tensors) of finished layers, oldest first


def side_checkpoint(device):
    """Called after a layer's backward has been enqueued.  Marks the side stream's position and releases the operands of the layer
    BEFORE this one: the main stream waits for that older mark (work enqueued a whole layer earlier - finished long ago in practice,
    so the wait costs nothing and does not serialise the two streams), after which the allocator may hand those blocks out again.
    Peak memory held for the side stream is two layers' operands instead of the whole stack's."""
    if not SIDE.enabled:
        _SIDE_KEEP.clear()
        return
    ev = torch.cuda.Event()
    ev.record(SIDE.stream(device))
    _SIDE_DONE.append((ev, list(_SIDE_KEEP)))
    _SIDE_KEEP.clear()
    while len(_SIDE_DONE) > 1:
        old_ev, _ = _SIDE_DONE.pop(0)
        torch.cuda.current_stream().wait_event(old_ev)


def join_side(device):
    """main stream waits for everything enqueued on the side stream; releases the tensors held for it"""
    if SIDE.enabled:
        ev = torch.cuda.Event()
        ev.record(SIDE.stream(device))
        torch.cuda.current_stream().wait_event(ev)
    _SIDE_KEEP.clear()
    _SIDE_DONE.clear()


class TransformerLayerFn(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, x, wqkv, bqkv, w1, b1, w2, b2, n_heads, causal, p_attn=0.0, p_mlp=0.0):
        B, N, D = x.shape
        need_grad = any(ctx.needs_input_grad)
        x2, saved, drop = layer_forward(_f32c(x).view(B * N, D), wqkv, _f32c(bqkv), w1, _f32c(b1), w2, _f32c(b2), B, N, n_heads,
                                        causal, need_grad, p_attn, p_mlp)
        if need_grad:
            ctx.save_for_backward(*saved)
            ctx.weights = (wqkv, w1, w2)
        ctx.drop = drop
        ctx.meta = (B, N, D, n_heads, causal, x.dtype)
        return x2.view(B, N, D).to(x.dtype)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        B, N, D, H, causal, xdtype = ctx.meta
        wqkv, w1, w2 = ctx.weights
        (grads,), _ = grad_arena(D, 1, g.device)
        g0, _ = layer_backward(_f32c(g).view(B * N, D), ctx.saved_tensors, wqkv, w1, w2, B, N, H, causal, grads, drop=ctx.drop)
        join_side(g.device)
        return (g0.view(B, N, D).to(xdtype), *grads, None, None, None, None)


class TransformerStackFn(torch.autograd.Function):
    """All layers of reference transformer.Transformer (transformer.py:52-54) in one autograd node:
    the backward of layer i+1's first LayerNorm hands bf16(g) and its column sums straight to layer
    i's fc2 weight/bias gradient, so no separate cast / column-sum passes exist between layers."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, n_heads, causal, p_attn, p_mlp, *params):
        B, N, D = x.shape
        L = len(params) // 6
        need_grad = any(ctx.needs_input_grad)
        cur = _f32c(x).view(B * N, D)
        saved_all, drops = [], []
        WEIGHTS.prepare([params[6 * i + j] for i in range(L) for j in (0, 2, 4)], need_grad)
        pending = None
        for i in range(L):
            wqkv, bqkv, w1, b1, w2, b2 = params[6 * i: 6 * i + 6]
            defer = DEFER_RESID and i + 1 < L and p_mlp == 0          # the last layer (no LayerNorm follows) and dropout keep the fused epilogue
            out, saved, drop = layer_forward(cur, wqkv, _f32c(bqkv), w1, _f32c(b1), w2, _f32c(b2), B, N, n_heads, causal, need_grad,
                                             p_attn, p_mlp, pending=pending, defer=defer)
            cur, pending = out if defer else (out, None)
            drops.append(drop)
            if need_grad:
                saved_all.extend(saved)
        if need_grad:
            ctx.save_for_backward(*saved_all)
            ctx.params = params
        ctx.drops = drops
        ctx.meta = (B, N, D, n_heads, causal, L, x.dtype)
        return cur.view(B, N, D).to(x.dtype)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        B, N, D, H, causal, L, xdtype = ctx.meta
        saved_all = ctx.saved_tensors
        params = ctx.params
        n_saved = len(saved_all) // L
        cur = _f32c(g).view(B * N, D)
        arena, sink = grad_arena(D, L, cur.device, params)     # sink: the DDP wrapper whose buckets the arena lives in (or None)
        dy2 = None
        try:
            for i in reversed(range(L)):
                wqkv, _, w1, _, w2, _ = params[6 * i: 6 * i + 6]
                nxt_db2 = arena[i - 1][5] if i > 0 else None      # layer i's first LN backward feeds layer i-1's fc2 bias grad
                cur, dy2 = layer_backward(cur, saved_all[n_saved * i: n_saved * (i + 1)], wqkv, w1, w2, B, N, H, causal, arena[i],
                                          dy2=dy2, have_db2=dy2 is not None, emit_bf16=i > 0, emit_colsum=nxt_db2,
                                          drop=ctx.drops[i], emit_dropout=ctx.drops[i - 1][2:] if i > 0 else (0.0, 0))
                if sink is not None:
                    # bucket i is complete now: five gradients from this call, and its fc2 bias gradient was
                    # added by layer i+1's LN1 backward (or by this call's own column sum for the top layer)
                    sink.layer_ready(params, i)
                side_checkpoint(cur.device)
        finally:
            join_side(cur.device)       # also on an exception: nothing stays pinned for the side stream
        grads = [t for layer in arena for t in layer]
        return (cur.view(B, N, D).to(xdtype), None, None, None, None, *grads)


# ------------------------------------------------------------------------------------------------
# fused-QKV attention as a stand-alone module (reference transformer.py:26-29)
# ------------------------------------------------------------------------------------------------
class AttentionFn(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, x, wqkv, bqkv, n_heads, causal, p_attn=0.0):
        B, N, D = x.shape
        need_grad = any(ctx.needs_input_grad)
        wb, _ = WEIGHTS.get(wqkv, need_grad)
        xb = ops.cast_bf16(_f32c(x).view(B * N, D))
        qkv = ops.gemm_nt(xb, wb, ops.EPI_BIAS_BF16, bias=_f32c(bqkv))
        ctx.drop = (p_attn, new_seed() if p_attn > 0 else 0)
        o, lse = ops.attention_fwd(qkv, B, N, n_heads, causal, dropout=ctx.drop)
        if need_grad:
            ctx.save_for_backward(xb, qkv, o, lse)
            ctx.wqkv = wqkv
        ctx.meta = (B, N, D, n_heads, causal, x.dtype)
        return o.view(B, N, D).to(x.dtype)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        B, N, D, H, causal, xdtype = ctx.meta
        xb, qkv, o, lse = ctx.saved_tensors
        _, wt = WEIGHTS.get(ctx.wqkv, True)
        d_o = ops.cast_bf16(_f32c(g).view(B * N, D))
        db = torch.zeros((3 * D,), dtype=F32, device=g.device)
        dqkv = ops.attention_bwd(qkv, o, lse, d_o, B, N, H, causal, dbias=db, dropout=ctx.drop)
        dW = torch.empty((3 * D, D), dtype=F32, device=g.device)
        ops.gemm_tn(dqkv, xb, dW, accumulate=False)
        dx = ops.gemm_nt(dqkv, wt, ops.EPI_BIAS_BF16)
        return dx.view(B, N, D).to(xdtype), dW, db, None, None, None


# ------------------------------------------------------------------------------------------------
# patch embedding + token assembly (reference train_vit.py:38-44)
# ------------------------------------------------------------------------------------------------
class PatchEmbedFn(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, images, conv_w, conv_b, pos_w, extra_w, patch, n_patches):
        B = images.shape[0]
        D = conv_w.shape[0]
        extra = extra_w.shape[0]
        seq = n_patches + extra
        need_grad = any(ctx.needs_input_grad)
        patches = ops.im2col(_f32c(images), patch)  # [B*np, C*p*p] bf16
        if patches.shape[0] != B * n_patches:
            raise ops._lib.VitamdError(f"patch grid gives {patches.shape[0] // B} patches, config says {n_patches}")
        wb, _ = WEIGHTS.get(conv_w, ctx.needs_input_grad[0])
        x = torch.empty((B * seq, D), dtype=F32, device=images.device)
        ops.gemm_nt(patches, wb, ops.EPI_PATCH_F32, bias=_f32c(conv_b), aux=_f32c(pos_w)[:n_patches].contiguous(), out=x,
                    n_patches=n_patches, seq=seq, extra=extra)
        x = x.view(B, seq, D)
        if extra > 0:
            x[:, :extra] = extra_w.detach().to(F32)  # learned extra tokens are PREPENDED (train_vit.py:43-44)
        if need_grad:
            ctx.save_for_backward(patches)
            ctx.conv_w = conv_w
        ctx.meta = (B, D, extra, seq, n_patches, tuple(conv_w.shape), pos_w.shape[0], tuple(images.shape), patch)
        return x

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        B, D, extra, seq, n_patches, wshape, pos_rows, ishape, patch = ctx.meta
        (patches,) = ctx.saved_tensors
        dpos, dextra, dyp, dbias = ops.embed_bwd(_f32c(g).view(B * seq, D), B, seq, extra, D)
        dW = torch.empty((D, patches.shape[1]), dtype=F32, device=g.device)
        ops.gemm_tn(dyp, patches, dW, accumulate=False)
        dimg = None
        if ctx.needs_input_grad[0]:
            # d(patches) = dY . W ; patches do not overlap, so the inverse gather is a pure permutation
            _, wt = WEIGHTS.get(ctx.conv_w, True)
            dpatch = ops.gemm_nt(dyp, wt, ops.EPI_BIAS_BF16)            # [B*np, C*p*p]
            _, C, Hh, Ww = ishape
            gh, gw = Hh // patch, Ww // patch
            dimg = dpatch.view(B, gh, gw, C, patch, patch).permute(0, 3, 1, 4, 2, 5).reshape(B, C, gh * patch, gw * patch).to(F32)
            if (gh * patch, gw * patch) != (Hh, Ww):
                full = torch.zeros(ishape, dtype=F32, device=g.device)
                full[:, :, : gh * patch, : gw * patch] = dimg
                dimg = full
        if pos_rows != n_patches:
            full = torch.zeros((pos_rows, D), dtype=F32, device=g.device)
            full[:n_patches] = dpos
            dpos = full
        return dimg, dW.view(wshape), dbias, dpos, dextra, None, None


# ------------------------------------------------------------------------------------------------
# generic Linear on the kernels (classifier head, reference train_vit.py:51,53)
# ------------------------------------------------------------------------------------------------
def _pad64(n):
    return (n + 63) // 64 * 64


_PADDED = {}     # id(weight) -> zero-padded bf16 copies (and padded fp32 bias) of a generic Linear's parameters


def _padded_weight(w, b, Np, Kp):
    """bf16 [Np,Kp] / transposed [Kp,Np] copies of w zero-padded to multiples of 64, and the padded fp32 bias; re-made only when the
    parameter (or the bias) changed or the weight-cache epoch moved on (once per optimiser step), not on every call"""
    base = w._base if w._base is not None else w          # callers hand in fresh views of a parameter (conv weights as [out, in*k*k]): key on the parameter
    ent = _PADDED.get(id(base))
    bver = -1 if b is None else b._version
    if (ent is None or ent["ref"]() is not base or ent["ver"] != w._version or ent["ptr"] != w.data_ptr() or ent["epoch"] != WEIGHTS.epoch
            or ent["bver"] != bver or ent["shape"] != (Np, Kp)):
        if len(_PADDED) > 1024:
            for k in [k for k, v in _PADDED.items() if v["ref"]() is None]:
                del _PADDED[k]
        Nout, K = w.shape
        wp = torch.zeros((Np, Kp), dtype=F32, device=w.device)
        wp[:Nout, :K] = w.detach()
        bp = torch.zeros((Np,), dtype=F32, device=w.device)
        if b is not None:
            bp[:Nout] = b.detach()
        wb, wbt = ops.cast_weight(wp, True, True)
        ent = _PADDED[id(base)] = {"ref": weakref.ref(base), "ver": w._version, "ptr": w.data_ptr(), "epoch": WEIGHTS.epoch, "bver": bver,
                                "shape": (Np, Kp), "wb": wb, "wbt": wbt, "bp": bp}
    return ent["wb"], ent["wbt"], ent["bp"]


class LinearFn(torch.autograd.Function):
    """y = x W^T + b through the MFMA GEMMs for arbitrary (small, odd) feature sizes: the output dim
    is zero-padded to a multiple of 64 (it is the reduction dim of the input-gradient GEMM) and the
    input dim likewise (reduction dim of the forward GEMM).  x: [M, K] -> [M, Nout] fp32."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w, b):
        M, K = x.shape
        Nout = w.shape[0]
        Np, Kp = _pad64(Nout), _pad64(K)
        dev = x.device
        wb, wbt, bp = _padded_weight(w, b, Np, Kp)
        xf = _f32c(x)
        if Kp != K:
            xpad = torch.zeros((M, Kp), dtype=F32, device=dev)
            xpad[:, :K] = xf
            xf = xpad
        xb = ops.cast_bf16(xf)
        y = ops.gemm_nt(xb, wb, ops.EPI_BIAS_BF16, bias=bp)
        ctx.save_for_backward(xb, wbt)
        ctx.meta = (M, K, Kp, Nout, Np, x.dtype, b is not None)
        return y[:, :Nout].to(F32)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        M, K, Kp, Nout, Np, xdtype, has_b = ctx.meta
        xb, wbt = ctx.saved_tensors
        gp = torch.zeros((M, Np), dtype=F32, device=g.device)
        gp[:, :Nout] = g
        gb = ops.cast_bf16(gp)
        dx = ops.gemm_nt(gb, wbt, ops.EPI_BIAS_BF16) if ctx.needs_input_grad[0] else None
        dWp = torch.empty((Np, Kp), dtype=F32, device=g.device)
        ops.gemm_tn(gb, xb, dWp, accumulate=False)
        db = ops.colsum(gb)[:Nout].clone() if has_b else None
        return (dx[:, :K].to(xdtype) if dx is not None else None), dWp[:Nout, :K].clone(), db


def linear(x, weight, bias):
    """nn.Linear semantics for any leading shape on the HIP path."""
    lead = x.shape[:-1]
    y = LinearFn.apply(x.reshape(-1, x.shape[-1]), weight.reshape(weight.shape[0], -1), bias)
    return y.view(*lead, weight.shape[0])
