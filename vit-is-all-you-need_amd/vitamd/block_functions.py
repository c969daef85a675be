"""autograd glue for the `blocks.py` surface of the reference: pre-LN blocks with
AFFINE LayerNorm and an attention OUTPUT projection (blocks.ResidualAttentionBlock, blocks.UViTBlock),
stand-alone blocks.Attention and blocks.Mlp.  Same kernels and the same bf16 dtype flow as
vitamd/functions.py (fp32 residual stream, bf16 GEMM / attention operands, fp32 parameter gradients).
Not on the measured ViT path; kept apart from functions.py so that path stays untouched."""
from __future__ import annotations

import torch

from . import ops
from .functions import _amp_fwd, _amp_bwd
from .functions import WEIGHTS, _f32c

BF16, F32 = torch.bfloat16, torch.float32


def _zeros(n, dev):
    return torch.zeros((n,), dtype=F32, device=dev)


class BlockFn(torch.autograd.Function):
    """x = x + proj(attn(LN1(x))) ; x = x + fc2(gelu(fc1(LN2(x))))   (reference blocks.py:62-70, 193-201)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, g1, be1, wqkv, bqkv, wo, bo, g2, be2, w1, b1, w2, b2, n_heads, has_mlp):
        B, N, D = x.shape
        H = n_heads
        x0 = _f32c(x).view(B * N, D)
        wqkv_b, _ = WEIGHTS.get(wqkv, True)
        wo_b, _ = WEIGHTS.get(wo, True)
        a, mean1, rstd1 = ops.layernorm_affine_fwd(x0, _f32c(g1), _f32c(be1))
        qkv = ops.gemm_nt(a, wqkv_b, ops.EPI_BIAS_BF16, bias=_f32c(bqkv) if bqkv is not None else None)
        o, lse = ops.attention_fwd(qkv, B, N, H, False)
        x1 = ops.gemm_nt(o, wo_b, ops.EPI_RESID_F32, bias=_f32c(bo), aux=x0)          # out-projection + residual
        saved = [x0, mean1, rstd1, a, qkv, o, lse, x1]
        out = x1
        if has_mlp:
            w1_b, _ = WEIGHTS.get(w1, True)
            w2_b, _ = WEIGHTS.get(w2, True)
            bln, mean2, rstd2 = ops.layernorm_affine_fwd(x1, _f32c(g2), _f32c(be2))
            pre, h = ops.gemm_nt(bln, w1_b, ops.EPI_GELU_DG, bias=_f32c(b1))
            out = ops.gemm_nt(h, w2_b, ops.EPI_RESID_F32, bias=_f32c(b2), aux=x1)
            saved += [mean2, rstd2, bln, pre, h]
        ctx.save_for_backward(*saved)
        ctx.params = (g1, wqkv, bqkv, wo, g2, w1, w2)
        ctx.meta = (B, N, D, H, has_mlp, x.dtype)
        return out.view(B, N, D).to(x.dtype)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        B, N, D, H, has_mlp, xdtype = ctx.meta
        g1p, wqkv, bqkv, wo, g2p, w1, w2 = ctx.params
        sv = ctx.saved_tensors
        x0, mean1, rstd1, a, qkv, o, lse, x1 = sv[:8]
        dev = g.device
        gcur = _f32c(g).view(B * N, D)
        dW1 = db1 = dW2 = db2 = dg2 = dbe2 = None
        if has_mlp:
            mean2, rstd2, bln, pre, h = sv[8:]
            Dh = h.shape[1]
            _, w1_t = WEIGHTS.get(w1, True)
            _, w2_t = WEIGHTS.get(w2, True)
            dy2 = ops.cast_bf16(gcur)
            db2 = ops.colsum(dy2)
            dW2 = torch.empty((D, Dh), dtype=F32, device=dev)
            ops.gemm_tn(dy2, h, dW2, accumulate=False)
            db1 = _zeros(Dh, dev)
            dpre = ops.gemm_nt(dy2, w2_t, ops.EPI_DMUL, aux=pre, colsum=db1)
            dW1 = torch.empty((Dh, D), dtype=F32, device=dev)
            ops.gemm_tn(dpre, bln, dW1, accumulate=False)
            dbln = ops.gemm_nt(dpre, w1_t, ops.EPI_BIAS_BF16)
            dg2, dbe2 = _zeros(D, dev), _zeros(D, dev)
            dbo = _zeros(D, dev)
            gcur, dyo = ops.layernorm_affine_bwd(dbln, x1, mean2, rstd2, _f32c(g2p), dg2, dbe2, g_res=gcur, want_bf16=True, colsum=dbo)
        else:
            dyo = ops.cast_bf16(gcur)
            dbo = ops.colsum(dyo)
        # ---- attention with output projection
        _, wo_t = WEIGHTS.get(wo, True)
        _, wqkv_t = WEIGHTS.get(wqkv, True)
        dWo = torch.empty((D, D), dtype=F32, device=dev)
        ops.gemm_tn(dyo, o, dWo, accumulate=False)
        d_o = ops.gemm_nt(dyo, wo_t, ops.EPI_BIAS_BF16)
        dbqkv = _zeros(3 * D, dev)
        dqkv = ops.attention_bwd(qkv, o, lse, d_o, B, N, H, False, dbias=dbqkv)
        dWqkv = torch.empty((3 * D, D), dtype=F32, device=dev)
        ops.gemm_tn(dqkv, a, dWqkv, accumulate=False)
        da = ops.gemm_nt(dqkv, wqkv_t, ops.EPI_BIAS_BF16)
        dg1, dbe1 = _zeros(D, dev), _zeros(D, dev)
        g0, _ = ops.layernorm_affine_bwd(da, x0, mean1, rstd1, _f32c(g1p), dg1, dbe1, g_res=gcur)
        return (g0.view(B, N, D).to(xdtype), dg1, dbe1, dWqkv, dbqkv if bqkv is not None else None, dWo, dbo, dg2, dbe2,
                dW1, db1, dW2, db2, None, None)


def _seeds(*rates):
    """one fresh 64-bit seed per non-zero rate (0 otherwise); drawn from torch's CPU generator like functions.new_seed()"""
    from .functions import new_seed
    return tuple(new_seed() if r > 0 else 0 for r in rates)


class DropPathFn(torch.autograd.Function):
    """Stochastic depth on a stand-alone tensor (reference blocks.py:124-152): one keep decision per sample, scaled by 1/keep_prob."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, p):
        xf = _f32c(x)
        (seed,) = _seeds(p)
        ctx.meta = (p, seed, xf[0].numel() if xf.dim() > 0 else 1, x.dtype)
        return ops.dropout(xf, p, seed, group=ctx.meta[2]).to(x.dtype)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        p, seed, group, xdtype = ctx.meta
        return ops.dropout(_f32c(g), p, seed, group=group).to(xdtype), None


class AttnProjFn(torch.autograd.Function):
    """drop_path(proj_drop(proj(attention(qkv(x)))))  — reference blocks.Attention (blocks.py:84-121) and the DropPath a UViTBlock wraps
    around it (blocks.py:194-199).  p_proj / p_path = 0 (eval mode) is the identity on those two."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, wqkv, bqkv, wo, bo, n_heads, p_proj=0.0, p_path=0.0):
        B, N, D = x.shape
        xb = ops.cast_bf16(_f32c(x).view(B * N, D))
        wqkv_b, _ = WEIGHTS.get(wqkv, True)
        wo_b, _ = WEIGHTS.get(wo, True)
        qkv = ops.gemm_nt(xb, wqkv_b, ops.EPI_BIAS_BF16, bias=_f32c(bqkv) if bqkv is not None else None)
        o, lse = ops.attention_fwd(qkv, B, N, n_heads, False)
        y = ops.gemm_nt(o, wo_b, ops.EPI_BIAS_BF16, bias=_f32c(bo))
        ctx.drops = (p_proj, p_path) + _seeds(p_proj, p_path)
        if p_proj > 0:
            ops.dropout(y, p_proj, ctx.drops[2], inplace=True)
        if p_path > 0:
            ops.dropout(y, p_path, ctx.drops[3], group=N * D, inplace=True)
        ctx.save_for_backward(xb, qkv, o, lse)
        ctx.params = (wqkv, bqkv, wo)
        ctx.meta = (B, N, D, n_heads, x.dtype)
        return y.view(B, N, D).to(x.dtype)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        B, N, D, H, xdtype = ctx.meta
        wqkv, bqkv, wo = ctx.params
        xb, qkv, o, lse = ctx.saved_tensors
        dev = g.device
        dy = ops.cast_bf16(_f32c(g).view(B * N, D))
        p_proj, p_path, s_proj, s_path = ctx.drops
        if p_path > 0:
            ops.dropout(dy, p_path, s_path, group=N * D, inplace=True)
        if p_proj > 0:
            ops.dropout(dy, p_proj, s_proj, inplace=True)
        dbo = ops.colsum(dy)
        _, wo_t = WEIGHTS.get(wo, True)
        _, wqkv_t = WEIGHTS.get(wqkv, True)
        dWo = torch.empty((D, D), dtype=F32, device=dev)
        ops.gemm_tn(dy, o, dWo, accumulate=False)
        d_o = ops.gemm_nt(dy, wo_t, ops.EPI_BIAS_BF16)
        dbqkv = _zeros(3 * D, dev)
        dqkv = ops.attention_bwd(qkv, o, lse, d_o, B, N, H, False, dbias=dbqkv)
        dWqkv = torch.empty((3 * D, D), dtype=F32, device=dev)
        ops.gemm_tn(dqkv, xb, dWqkv, accumulate=False)
        dx = ops.gemm_nt(dqkv, wqkv_t, ops.EPI_BIAS_BF16)
        return dx.view(B, N, D).to(xdtype), dWqkv, dbqkv if bqkv is not None else None, dWo, dbo, None, None, None


class MlpFn(torch.autograd.Function):
    """drop_path(drop(fc2(drop(gelu(fc1(x))))))  — reference blocks.Mlp (blocks.py:155-171: the same nn.Dropout after the activation and
    after fc2, two independent masks) and the DropPath a UViTBlock wraps around it.  p_drop / p_path = 0 is the identity on those."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w1, b1, w2, b2, p_drop=0.0, p_path=0.0):
        lead, K = x.shape[:-1], x.shape[-1]
        xb = ops.cast_bf16(_f32c(x).reshape(-1, K))
        w1_b, _ = WEIGHTS.get(w1, True)
        w2_b, _ = WEIGHTS.get(w2, True)
        pre, h = ops.gemm_nt(xb, w1_b, ops.EPI_GELU_DG, bias=_f32c(b1))
        ctx.drops = (p_drop, p_path) + _seeds(p_drop, p_drop, p_path)
        if p_drop > 0:
            # the mask of the first dropout goes onto BOTH gelu(pre) and the stored gelu'(pre): the backward's fused multiply by
            # the stored derivative then carries it, and the fc1 bias gradient (column sums of that product) comes out right
            ops.dropout(h, p_drop, ctx.drops[2], inplace=True)
            ops.dropout(pre, p_drop, ctx.drops[2], inplace=True)
        y = ops.gemm_nt(h, w2_b, ops.EPI_BIAS_BF16, bias=_f32c(b2))
        if p_drop > 0:
            ops.dropout(y, p_drop, ctx.drops[3], inplace=True)
        if p_path > 0:
            ops.dropout(y, p_path, ctx.drops[4], group=y.numel() // lead[0] if len(lead) > 1 else y.shape[1], inplace=True)
        ctx.save_for_backward(xb, pre, h)
        ctx.params = (w1, w2)
        ctx.meta = (lead, K, x.dtype)
        return y.view(*lead, w2.shape[0]).to(x.dtype)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        lead, K, xdtype = ctx.meta
        w1, w2 = ctx.params
        xb, pre, h = ctx.saved_tensors
        dev = g.device
        Dh, Dout = h.shape[1], w2.shape[0]
        dy = ops.cast_bf16(_f32c(g).reshape(-1, Dout))
        p_drop, p_path, _, s_out, s_path = ctx.drops
        if p_path > 0:
            ops.dropout(dy, p_path, s_path, group=dy.numel() // lead[0] if len(lead) > 1 else dy.shape[1], inplace=True)
        if p_drop > 0:
            ops.dropout(dy, p_drop, s_out, inplace=True)
        db2 = ops.colsum(dy)
        _, w1_t = WEIGHTS.get(w1, True)
        _, w2_t = WEIGHTS.get(w2, True)
        dW2 = torch.empty((Dout, Dh), dtype=F32, device=dev)
        ops.gemm_tn(dy, h, dW2, accumulate=False)
        db1 = _zeros(Dh, dev)
        dpre = ops.gemm_nt(dy, w2_t, ops.EPI_DMUL, aux=pre, colsum=db1)
        dW1 = torch.empty((Dh, K), dtype=F32, device=dev)
        ops.gemm_tn(dpre, xb, dW1, accumulate=False)
        dx = ops.gemm_nt(dpre, w1_t, ops.EPI_BIAS_BF16)
        return dx.view(*lead, K).to(xdtype), dW1, db1, dW2, db2, None, None


class LayerNormAffineFn(torch.autograd.Function):
    """Stand-alone nn.LayerNorm (weight + bias) in fp32: the tokenizers' ln_pre / ln_post (reference blocks.py:247,253,
    320,326), whose input and output both live in the fp32 token stream."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, gamma, beta, eps):
        D = x.shape[-1]
        x2 = _f32c(x).reshape(-1, D)
        y, mean, rstd = ops.layernorm_affine_fwd_f32(x2, _f32c(gamma), _f32c(beta), eps)
        ctx.save_for_backward(x2, mean, rstd, gamma)
        ctx.meta = (tuple(x.shape), x.dtype)
        return y.view(x.shape)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        x2, mean, rstd, gamma = ctx.saved_tensors
        shape, xdtype = ctx.meta
        D = shape[-1]
        dg, db = _zeros(D, g.device), _zeros(D, g.device)
        dx = ops.layernorm_affine_bwd_f32(_f32c(g).reshape(-1, D), x2, mean, rstd, _f32c(gamma), dg, db)
        return dx.view(shape).to(xdtype), dg, db, None


class Conv3x3Fn(torch.autograd.Function):
    """nn.Conv2d(3, 3, 3, padding=1) on NCHW fp32 images (reference blocks.py:333 `conv_out`)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w, b):
        xf, wf = _f32c(x), _f32c(w)
        y = ops.conv3x3_fwd(xf, wf, _f32c(b) if b is not None else None)
        ctx.save_for_backward(xf, wf)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        xf, wf = ctx.saved_tensors
        dx, dw, db = ops.conv3x3_bwd(xf, wf, _f32c(g), need_dx=ctx.needs_input_grad[0], need_dw=True, has_bias=ctx.has_bias)
        return dx, dw, db
