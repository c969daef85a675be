"""Host wrappers: torch device tensors in, libvitamd.so kernel launches on torch's current HIP stream.

PyTorch is plumbing here (HBM allocation, streams); every computation on the path is a kernel of
libvitamd.so.  All functions require ROCm device tensors and raise otherwise — no CPU fallback.
"""
from __future__ import annotations

import torch

from . import lib as _lib

EPI_BIAS_BF16, EPI_GELU, EPI_RESID_F32, EPI_DGELU, EPI_PATCH_F32, EPI_F32, EPI_GELU_DG, EPI_DMUL = range(8)
# Large NT GEMMs are launched persistent by default (one workgroup per CU: -0.3 ms/step inside the training step and the reason the
# weight-gradient stream can use the whole chip).  A persistent workgroup whose CU is held by another
# long-running kernel starts late with its whole tile list still to do, so the form is sensitive to resident foreign kernels
# (DESIGN.md section 7); VITAMD_NT_PERSISTENT=0 (or ops.NT_PERSISTENT = False at any time: it is read on every call, and the
# weight-gradient split rule of functions._tn_splits follows it) selects one workgroup per tile instead.  vitamd.ddp.DataParallel
# measures both forms beside its collectives at construction and sets it for multi-rank jobs (ddp.choose_launch_form).
import os as _os
NT_PERSISTENT = _os.environ.get("VITAMD_NT_PERSISTENT", "1") != "0"
# The seam form of the persistent NT kernel (csrc/gemm_nt_seam.h) issues its epilogue as a burst of buffer stores from inline asm.  On one box of the
# pool, round 3 saw buffer stores issued from inside a GEMM run several times slower than anywhere else (profiles/r03/store_trickle_README.md, last
# row); the form is therefore PROBED once per process and DEVICE against the plain persistent form on a QKV-sized problem and switched off there if it
# loses by more than 25 % (VITAMD_NT_SEAM=1 / 0 decides for every device and skips the probe).  Every GELU epilogue reads the same table, so the
# decision changes timing only, never bits.
_seam_env = _os.environ.get("VITAMD_NT_SEAM", "auto")
NT_SEAM = {"1": True, "0": False}.get(_seam_env)      # process-wide override (None = per device, by probe)
SEAM_PROBE = {}          # device index -> {"seam_us", "plain_us", "enabled"}: the source of truth for the per-device decision
NT_FORM_SEAM, NT_FORM_LOADER = 4, 5         # include/vitamd.h VITAMD_NT_FORM_* (vitamd_gemm_nt_plan): the two forms whose epilogue is a burst of asm buffer stores
_RAW_AUTO = -1
LN_EPS = 1e-5
BF16, F32 = torch.bfloat16, torch.float32


def _L():
    return _lib.load()


_INITIALISED = set()     # device indices vitamd_init has run for


def init(device=None):
    """Per-device set-up of the library (C ABI vitamd_init: builds the 16-KiB erf-GELU table image - the only allocation / synchronisation the
    library ever makes).  Idempotent and cheap after the first call; called by the GEMM wrapper before a GELU launch, by functions.claim_streams
    and by GraphedStep's eager warm-up, so it never first happens inside a stream capture."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _INITIALISED:
        return
    if torch.cuda.is_current_stream_capturing():
        raise _lib.VitamdError("vitamd.ops.init: first use of a device inside a stream capture; run one eager step (or ops.init(device)) before capturing")
    _lib.check(_L().vitamd_init(idx, torch.cuda.current_stream(device).cuda_stream), f"vitamd_init[device {idx}]")
    _INITIALISED.add(idx)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _need(t, dtype, name, ndim=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.VitamdError(f"{name}: expected a ROCm device tensor (the HIP kernels are the only implementation)")
    if t.dtype != dtype:
        raise _lib.VitamdError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.VitamdError(f"{name}: must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise _lib.VitamdError(f"{name}: expected {ndim}-d, got {t.dim()}-d")
    return t


# ------------------------------------------------------------------------------------------ GEMMs
def gemm_nt(a, b, epi, *, bias=None, aux=None, out=None, out2=None, colsum=None, n_patches=0, seq=0, extra=0,
            out_rows=None, tile=0):
    """out = epilogue(a[M,K] @ b[N,K]^T).  Returns out (and out2 for EPI_GELU / EPI_GELU_DG)."""
    _need(a, BF16, "a", 2); _need(b, BF16, "b", 2)
    M, K = a.shape
    N, K2 = b.shape
    if K != K2:
        raise _lib.VitamdError(f"gemm_nt: K mismatch {K} vs {K2}")
    out_dtype = F32 if epi in (EPI_RESID_F32, EPI_PATCH_F32, EPI_F32) else BF16
    if out is None:
        out = torch.empty((out_rows if out_rows is not None else M, N), dtype=out_dtype, device=a.device)
    _need(out, out_dtype, "out")
    if epi in (EPI_GELU, EPI_GELU_DG) and out2 is None:
        out2 = torch.empty((M, N), dtype=BF16, device=a.device)
    if bias is not None:
        _need(bias, F32, "bias", 1)
    if epi in (EPI_GELU, EPI_GELU_DG):
        init(a.device)
    if tile == 0:
        tile = auto_tile(a.device, M, N, K, epi)
    elif tile == _RAW_AUTO:             # the library's own automatic choice, no host-side policy (the probe's seam arm)
        tile = 0
    code = _L().vitamd_gemm_nt_bf16(_p(a), _p(b), _p(out), _p(out2), _p(bias), _p(aux), _p(colsum), M, N, K, N, epi,
                                    n_patches, seq, extra, tile, _stream())
    _lib.check(code, f"gemm_nt[M={M},N={N},K={K},epi={epi}]")
    return (out, out2) if epi in (EPI_GELU, EPI_GELU_DG) else out


def seam_enabled(device):
    """The seam form's switch for `device`: the process-wide override, else this device's probe result, else None (not probed yet)."""
    if NT_SEAM is not None:
        return NT_SEAM
    idx = device.index if device.index is not None else torch.cuda.current_device()
    rec = SEAM_PROBE.get(idx)
    return None if rec is None else rec["enabled"]


def auto_tile(device, M, N, K, epi):
    """The ABI `tile` code behind tile = 0: 512 (one workgroup per tile) when persistent launches are off; 1024 (persistent, no seam form) on a
    device whose seam / loader forms are switched off - ALWAYS, whatever the shape: the library treats 1024 as plain auto where its seam rule would not
    apply anyway; 0 otherwise.  The device is probed the first time the library's own rule (vitamd_gemm_nt_plan) would pick the seam or the loader form for a
    launch; never inside a stream capture (the form then stays on, unrecorded, until an eager launch probes)."""
    if not NT_PERSISTENT:
        return 512
    on = seam_enabled(device)
    if on is None:
        if _L().vitamd_gemm_nt_plan(M, N, K, N, epi, 0) & 0x7f not in (NT_FORM_SEAM, NT_FORM_LOADER) or torch.cuda.is_current_stream_capturing():
            return 0
        on = seam_probe(device)
    return 0 if on else 1024


def seam_probe(device, rows=49152, reps=3):
    """Time the seam form against the plain persistent form (QKV shape of ViT-B at `rows` rows: 192 x 9 tiles = 6.75 per CU, K = 768; the seam form is worth ~5 % there and costs ~5 % at 3.4 tiles per CU) on `device`; the result is kept per device in SEAM_PROBE.
    ~3 ms once per process and device."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    x = torch.randn(rows, 768, device=device).to(BF16)
    w = torch.randn(2304, 768, device=device).mul_(0.03).to(BF16)
    out = torch.empty((rows, 2304), dtype=BF16, device=device)
    times = {}
    for name, tile in (("seam_us", _RAW_AUTO), ("plain_us", 1024)):
        launch = lambda: gemm_nt(x, w, EPI_BIAS_BF16, out=out, tile=tile)
        launch()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            launch()
        e.record()
        e.synchronize()
        times[name] = s.elapsed_time(e) / reps * 1e3
    on = times["seam_us"] <= 1.25 * times["plain_us"]           # the probe guards against a pathologically slow store path, not against a 5 % difference
    SEAM_PROBE[device.index] = {"seam_us": round(times["seam_us"], 1), "plain_us": round(times["plain_us"], 1), "enabled": on}
    if not on:
        import warnings
        warnings.warn(f"vitamd: the seam form of the NT GEMM is slower than the plain persistent form on {device} "
                      f"({times['seam_us']:.0f} vs {times['plain_us']:.0f} us): switched off for this device")
    return on


_WORKSPACES = {}


def _workspace(device, nbytes):
    """split-K scratch, one per (device, stream): kernels on different streams never share it"""
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = _WORKSPACES[key] = torch.empty((nbytes + 3) // 4, dtype=F32, device=device)
    return ws


TN_FORM_SHARED, TN_FORM_EXCLUSIVE = 0, 1      # include/vitamd.h VITAMD_TN_FORM_*


def gemm_tn(l, r, out, splits=0, accumulate=True, atomic=False, form=TN_FORM_SHARED):
    """out[P,Q] (fp32) (+)= l[R,P]^T @ r[R,Q].  Split-K partials go through a workspace + reduce
    pass (reproducible) unless atomic=True (fp32 atomics straight into `out`, accumulate only).
    form: TN_FORM_SHARED (8-wave workgroups that leave room on the CU for another stream's LayerNorm waves) or TN_FORM_EXCLUSIVE (12 waves,
    four of them dedicated to the LDS-DMA requests: 15 % faster alone, fills the CU); bit-identical results."""
    _need(l, BF16, "l", 2); _need(r, BF16, "r", 2); _need(out, F32, "out", 2)
    R, P = l.shape
    R2, Q = r.shape
    if R != R2 or tuple(out.shape) != (P, Q):
        raise _lib.VitamdError("gemm_tn: shape mismatch")
    if atomic:
        if not accumulate:
            raise _lib.VitamdError("gemm_tn: the atomic form can only accumulate")
        code = _L().vitamd_gemm_tn_bf16(_p(l), _p(r), _p(out), R, P, Q, P, Q, Q, splits, _stream())
    else:
        nbytes = _L().vitamd_gemm_tn_ws_bytes(R, P, Q, splits)
        ws = _workspace(l.device, nbytes)
        code = _L().vitamd_gemm_tn_bf16_ws(_p(l), _p(r), _p(out), R, P, Q, P, Q, Q, splits, _p(ws), ws.numel() * 4, int(accumulate),
                                           int(form), _stream())
    _lib.check(code, f"gemm_tn[R={R},P={P},Q={Q}]")
    return out


# ------------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, addend=None):
    """x fp32 [M,D] (+ addend bf16) -> (x_sum fp32 or x itself, y bf16, mean, rstd)."""
    _need(x, F32, "x", 2)
    M, D = x.shape
    y = torch.empty((M, D), dtype=BF16, device=x.device)
    mean = torch.empty((M,), dtype=F32, device=x.device)
    rstd = torch.empty((M,), dtype=F32, device=x.device)
    x_out = None
    if addend is not None:
        _need(addend, BF16, "addend", 2)
        x_out = torch.empty_like(x)
    code = _L().vitamd_layernorm_fwd(_p(x), _p(addend), _p(x_out), _p(y), _p(mean), _p(rstd), M, D, LN_EPS, _stream())
    _lib.check(code, f"layernorm_fwd[M={M},D={D}]")
    return (x_out if addend is not None else x), y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, g_res=None, want_bf16=False, colsum=None, dropout=(0.0, 0), xhat=None):
    """g = (g_res or 0) + LN'(dy); returns (g fp32, bf16(g) or None).  dropout=(p, seed): the bf16 copy
    also gets that dropout mask (it is then the gradient of a dropped-out Linear output).
    xhat: the forward's bf16 output (= xhat for this non-affine LayerNorm); given, and D in {256,512,768,1024}, the kernel reads
    it instead of recomputing xhat from the fp32 x (2 B instead of 4 B per element of an HBM-bound kernel)."""
    _need(dy, BF16, "dy", 2); _need(x, F32, "x", 2)
    M, D = x.shape
    g = torch.empty_like(x)
    gb = torch.empty((M, D), dtype=BF16, device=x.device) if want_bf16 else None
    if g_res is not None:
        _need(g_res, F32, "g_res", 2)
    if xhat is not None and D in (256, 512, 768, 1024):
        _need(xhat, BF16, "xhat", 2)
        code = _L().vitamd_layernorm_bwd_xhat(_p(dy), _p(xhat), _p(rstd), _p(g_res), _p(g), _p(gb), _p(colsum), M, D,
                                              float(dropout[0]), int(dropout[1]), _stream())
        _lib.check(code, f"layernorm_bwd_xhat[M={M},D={D}]")
        return g, gb
    code = _L().vitamd_layernorm_bwd_dropout(_p(dy), _p(x), _p(mean), _p(rstd), _p(g_res), _p(g), _p(gb), _p(colsum), M, D,
                                             float(dropout[0]), int(dropout[1]), _stream())
    _lib.check(code, f"layernorm_bwd[M={M},D={D}]")
    return g, gb


def layernorm_affine_fwd(x, gamma, beta, eps=LN_EPS):
    """x fp32 [M,D] -> (y bf16 = LN(x)*gamma+beta, mean, rstd)."""
    _need(x, F32, "x", 2); _need(gamma, F32, "gamma", 1); _need(beta, F32, "beta", 1)
    M, D = x.shape
    y = torch.empty((M, D), dtype=BF16, device=x.device)
    mean = torch.empty((M,), dtype=F32, device=x.device)
    rstd = torch.empty((M,), dtype=F32, device=x.device)
    _lib.check(_L().vitamd_layernorm_affine_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, D, float(eps), _stream()),
               "layernorm_affine_fwd")
    return y, mean, rstd


def layernorm_affine_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, g_res=None, want_bf16=False, colsum=None):
    """returns (g fp32, bf16(g) or None); dgamma / dbeta (fp32 [D]) are accumulated into."""
    _need(dy, BF16, "dy", 2); _need(x, F32, "x", 2); _need(gamma, F32, "gamma", 1)
    _need(dgamma, F32, "dgamma", 1); _need(dbeta, F32, "dbeta", 1)
    M, D = x.shape
    g = torch.empty_like(x)
    gb = torch.empty((M, D), dtype=BF16, device=x.device) if want_bf16 else None
    _lib.check(_L().vitamd_layernorm_affine_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(g_res), _p(g), _p(gb), _p(colsum),
                                                _p(dgamma), _p(dbeta), M, D, _stream()), "layernorm_affine_bwd")
    return g, gb


def layernorm_affine_fwd_f32(x, gamma, beta, eps=LN_EPS):
    """x fp32 [M,D] -> (y fp32 = LN(x)*gamma+beta, mean, rstd): the LayerNorm whose output is not a GEMM operand."""
    _need(x, F32, "x", 2); _need(gamma, F32, "gamma", 1); _need(beta, F32, "beta", 1)
    M, D = x.shape
    y = torch.empty_like(x)
    mean = torch.empty((M,), dtype=F32, device=x.device)
    rstd = torch.empty((M,), dtype=F32, device=x.device)
    _lib.check(_L().vitamd_layernorm_affine_fwd_f32(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, D, float(eps), _stream()),
               "layernorm_affine_fwd_f32")
    return y, mean, rstd


def layernorm_affine_bwd_f32(dy, x, mean, rstd, gamma, dgamma, dbeta):
    """dy fp32 -> dx fp32; dgamma / dbeta (fp32 [D]) are accumulated into."""
    _need(dy, F32, "dy", 2); _need(x, F32, "x", 2); _need(gamma, F32, "gamma", 1)
    _need(dgamma, F32, "dgamma", 1); _need(dbeta, F32, "dbeta", 1)
    M, D = x.shape
    g = torch.empty_like(x)
    _lib.check(_L().vitamd_layernorm_affine_bwd_f32(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(g), _p(dgamma), _p(dbeta), M, D,
                                                    _stream()), "layernorm_affine_bwd_f32")
    return g


# ------------------------------------------------------------------------------------------ attention
ATTN_RESID_MAX_N = 256    # the fused residual add lives in the register-resident-softmax forward kernel


def attention_fwd(qkv, B, N, H, causal=False, dropout=(0.0, 0), resid=None):
    """qkv bf16 [B*N, 3*H*64] (packed (qkv, head, dh)) -> o bf16 [B*N, H*64], lse2 fp32 [B,H,N].
    dropout=(p, seed): dropout on the softmax probabilities.
    resid (fp32 [B*N, H*64], N <= 256): also returns x1 = resid + o, written by the attention kernel itself -> (o, lse2, x1)."""
    _need(qkv, BF16, "qkv", 2)
    D = H * 64
    if tuple(qkv.shape) != (B * N, 3 * D):
        raise _lib.VitamdError("attention_fwd: qkv must be [B*N, 3*H*64] (head_dim 64 only)")
    o = torch.empty((B * N, D), dtype=BF16, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=F32, device=qkv.device)
    if resid is not None:
        _need(resid, F32, "resid", 2)
        if tuple(resid.shape) != (B * N, D):
            raise _lib.VitamdError("attention_fwd: resid must be [B*N, H*64]")
        x1 = torch.empty_like(resid)
        code = _L().vitamd_attention_fwd_resid(_p(qkv), _p(o), _p(lse), _p(resid), _p(x1), B, N, H, 64, int(causal), float(dropout[0]),
                                               int(dropout[1]), _stream())
        _lib.check(code, f"attention_fwd_resid[B={B},N={N},H={H}]")
        return o, lse, x1
    code = _L().vitamd_attention_fwd(_p(qkv), _p(o), _p(lse), B, N, H, 64, int(causal), float(dropout[0]), int(dropout[1]), _stream())
    _lib.check(code, f"attention_fwd[B={B},N={N},H={H}]")
    return o, lse


def attention_bwd(qkv, o, lse, d_o, B, N, H, causal=False, dbias=None, dropout=(0.0, 0)):
    """dqkv bf16 [B*N, 3*H*64]; the column sums of dqkv (QKV bias gradient) are added to `dbias` if given."""
    _need(qkv, BF16, "qkv", 2); _need(o, BF16, "o", 2); _need(d_o, BF16, "d_o", 2); _need(lse, F32, "lse")
    if dbias is not None:
        _need(dbias, F32, "dbias", 1)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    code = _L().vitamd_attention_bwd(_p(qkv), _p(o), _p(lse), _p(d_o), _p(dqkv), _p(delta), _p(dbias), B, N, H, 64, int(causal),
                                     float(dropout[0]), int(dropout[1]), _stream())
    _lib.check(code, f"attention_bwd[B={B},N={N},H={H}]")
    return dqkv


def linear_dropout_resid(a, b, bias, resid, dropout):
    """out f32 = resid + dropout_p(bf16(a @ b^T + bias)); dropout = (p, seed)."""
    _need(a, BF16, "a", 2); _need(b, BF16, "b", 2); _need(resid, F32, "resid", 2); _need(bias, F32, "bias", 1)
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty((M, N), dtype=F32, device=a.device)
    code = _L().vitamd_linear_dropout_resid_bf16(_p(a), _p(b), _p(out), _p(bias), _p(resid), M, N, K, float(dropout[0]), int(dropout[1]),
                                                 0 if NT_PERSISTENT else 512, _stream())
    _lib.check(code, f"linear_dropout_resid[M={M},N={N},K={K}]")
    return out


# ------------------------------------------------------------------------------------------ helpers
def cast_bf16_dropout(x, dropout):
    _need(x, F32, "x")
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    _lib.check(_L().vitamd_cast_f32_bf16_dropout(_p(x), _p(out), x.numel(), float(dropout[0]), int(dropout[1]), _stream()), "cast_dropout")
    return out


def dropout(x, p, seed, group=1, inplace=False):
    """x (bf16 or fp32) * keep(seed, index // group): training-mode nn.Dropout (group 1) / DropPath (group = elements per sample)."""
    if x.dtype not in (BF16, F32):
        raise _lib.VitamdError(f"dropout: expected bf16 or fp32, got {x.dtype}")
    _need(x, x.dtype, "x")
    out = x if inplace else torch.empty_like(x)
    fn = _L().vitamd_dropout_bf16 if x.dtype == BF16 else _L().vitamd_dropout_f32
    _lib.check(fn(_p(x), _p(out), x.numel(), int(group), float(p), int(seed), _stream()), "dropout")
    return out


def cast_bf16(x):
    _need(x, F32, "x")
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    _lib.check(_L().vitamd_cast_f32_bf16(_p(x), _p(out), x.numel(), _stream()), "cast_f32_bf16")
    return out


def cast_weight(w, want_plain=True, want_transposed=False):
    """fp32 [N,K] weight -> (bf16 [N,K] or None, bf16 [K,N] or None)."""
    _need(w, F32, "w", 2)
    N, K = w.shape
    wb = torch.empty((N, K), dtype=BF16, device=w.device) if want_plain else None
    wbt = torch.empty((K, N), dtype=BF16, device=w.device) if want_transposed else None
    _lib.check(_L().vitamd_cast_transpose_weight(_p(w), _p(wb), _p(wbt), N, K, _stream()), "cast_transpose_weight")
    return wb, wbt


def im2col(img, p):
    _need(img, F32, "img", 4)
    B, C, H, W = img.shape
    out = torch.empty((B * (H // p) * (W // p), C * p * p), dtype=BF16, device=img.device)
    _lib.check(_L().vitamd_im2col_bf16(_p(img), _p(out), B, C, H, W, p, _stream()), "im2col")
    return out


def colsum(x, out=None):
    _need(x, BF16, "x", 2)
    M, N = x.shape
    if out is None:
        out = torch.zeros((N,), dtype=F32, device=x.device)
    _lib.check(_L().vitamd_colsum_bf16(_p(x), _p(out), M, N, N, _stream()), "colsum")
    return out


def embed_bwd(g, B, seq, extra, D):
    """g fp32 [B*seq, D] -> dpos [seq-extra, D], dextra [extra, D], dyp bf16 [B*(seq-extra), D], dbias [D]."""
    _need(g, F32, "g", 2)
    n_p = seq - extra
    dev = g.device
    dpos = torch.zeros((n_p, D), dtype=F32, device=dev)
    dextra = torch.zeros((extra, D), dtype=F32, device=dev)
    dyp = torch.empty((B * n_p, D), dtype=BF16, device=dev)
    dbias = torch.zeros((D,), dtype=F32, device=dev)
    rows = torch.zeros((max(n_p, 1), D), dtype=F32, device=dev)
    _lib.check(_L().vitamd_embed_bwd(_p(g), _p(dpos), _p(dextra) if extra > 0 else None, _p(dyp), _p(dbias), _p(rows), B, seq, extra, D,
                                     _stream()), "embed_bwd")
    return dpos, dextra, dyp, dbias


def vq_nearest(x, codebook):
    """x fp32 [M,d], codebook fp32 [K,d] -> int64 [M] index of the nearest code (first minimum)."""
    _need(x, F32, "x", 2); _need(codebook, F32, "codebook", 2)
    M, d = x.shape
    K, d2 = codebook.shape
    if d != d2:
        raise _lib.VitamdError("vq_nearest: dim mismatch")
    idx = torch.empty((M,), dtype=torch.int64, device=x.device)
    _lib.check(_L().vitamd_vq_nearest(_p(x), _p(codebook), _p(idx), M, K, d, _stream()), "vq_nearest")
    return idx


def conv3x3_fwd(x, w, bias):
    """x fp32 [B,3,H,W], w fp32 [3,3,3,3], bias fp32 [3] or None -> y fp32 [B,3,H,W] (stride 1, zero padding 1)."""
    _need(x, F32, "x", 4); _need(w, F32, "w", 4)
    B, C, H, W = x.shape
    Co = w.shape[0]
    if tuple(w.shape) != (Co, C, 3, 3):
        raise _lib.VitamdError("conv3x3: weight must be [Cout, Cin, 3, 3]")
    y = torch.empty((B, Co, H, W), dtype=F32, device=x.device)
    _lib.check(_L().vitamd_conv3x3_fwd(_p(x), _p(w), _p(bias), _p(y), B, C, Co, H, W, _stream()), f"conv3x3_fwd[Cin={C},Cout={Co}]")
    return y


def conv3x3_bwd(x, w, dy, need_dx=True, need_dw=True, has_bias=True):
    """-> (dx or None, dw or None, db or None), all fp32."""
    _need(x, F32, "x", 4); _need(w, F32, "w", 4); _need(dy, F32, "dy", 4)
    B, C, H, W = x.shape
    Co = w.shape[0]
    if tuple(dy.shape) != (B, Co, H, W):
        raise _lib.VitamdError("conv3x3_bwd: dy shape mismatch")
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.zeros_like(w) if need_dw else None
    db = torch.zeros((Co,), dtype=F32, device=x.device) if (need_dw and has_bias) else None
    _lib.check(_L().vitamd_conv3x3_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), B, C, Co, H, W, _stream()), "conv3x3_bwd")
    return dx, dw, db
