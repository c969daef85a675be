"""CPU oracle for the ViT training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a functional, fp32, CPU restatement of the reference's algorithm for the hot
path.  It is NOT part of the product: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and there only
as the checker.  The product path (`vit-is-all-you-need_amd/`) never imports it and fails
loudly when the HIP library is missing.

Parity status: PINNED.  `oracle/gen_golden.py` imports the reference's own modules from
/root/reference in the build container (`transformer.py` as-is; the model classes of
`train_vit.py`) and writes golden input/output vectors to `tests/golden/`;
`tests/test_oracle.py` checks every function below against them (fp32, ~1e-6).

All functions take plain tensors and a `state_dict`-style mapping that uses the reference's
parameter names, so the checkpoint-key contract is exercised too.
Everything is written with explicit matmul/softmax arithmetic (no nn.Module, no SDPA, no
F.layer_norm) so that it is an independent statement of the maths.

`lowp=True` turns on an emulation of the reference's autocast dtype flow with bf16 as the
low-precision type (SURVEY.md section 5 "mixed precision": LayerNorm and the residual stream
in fp32; Linear / attention / GELU inputs and outputs rounded to bf16, fp32 accumulation;
gradients of bf16 tensors rounded to bf16).  That is the arithmetic the HIP kernels implement,
so kernel-vs-oracle(lowp) is the tight check and oracle(lowp)-vs-oracle(fp32) is the measured
precision floor of bf16 itself.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

LN_EPS = 1e-5  # F.layer_norm default eps; reference transformer.py:43-44 passes none


# --------------------------------------------------------------------------------------
# rounding helper for the lowp emulation
# --------------------------------------------------------------------------------------
class _RoundBF16(torch.autograd.Function):
    """x -> bf16 -> fp32 in forward; the incoming gradient is rounded the same way
    (under autocast the gradient of a bf16 tensor is itself a bf16 tensor)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _r(x: torch.Tensor, lowp: bool) -> torch.Tensor:
    return _RoundBF16.apply(x) if lowp else x


# --------------------------------------------------------------------------------------
# configs (reference transformer.py:5-14, 56-59; train_vit.py:16-28)
# --------------------------------------------------------------------------------------
PRESETS = {"S": (6, 8, 512), "B": (12, 12, 768), "L": (24, 16, 1024)}  # transformer.py:56-58


@dataclass
class OracleViTConfig:
    image_size: int
    in_channels: int
    patch_size: int
    n_layers: int
    n_heads: int
    n_embd: int
    extra_tokens: int
    n_patches: int | None = None  # callers may override (train_titok.py:32)

    def __post_init__(self):
        if self.n_patches is None:
            self.n_patches = (self.image_size // self.patch_size) ** 2  # train_vit.py:26

    @property
    def seq_len(self):
        return self.n_patches + self.extra_tokens  # train_vit.py:28 block_size

    @classmethod
    def preset(cls, image_size, in_channels, patch_size, transformer, extra_tokens):
        L, H, D = PRESETS[transformer]
        return cls(image_size, in_channels, patch_size, L, H, D, extra_tokens)


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def layer_norm(x: torch.Tensor) -> torch.Tensor:
    """Non-affine LayerNorm over the last dim, biased variance, eps 1e-5.
    Reference: transformer.py:43-44 `F.layer_norm(x, (self.n_embd,))`."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + LN_EPS)


def linear(x, w, b, lowp=False):
    """y = x @ w.T + b with PyTorch [out,in] weights.  Reference call sites
    transformer.py:21,37,39 (nn.Linear).  lowp: operands and result rounded to bf16."""
    y = _r(x, lowp) @ _r(w, lowp).t()
    if b is not None:
        y = y + _r(b, lowp)
    return _r(y, lowp)


class _GeluStoredGradBF16(torch.autograd.Function):
    """lowp emulation of the MI355X flow: the forward keeps bf16(gelu'(x)) (the derivative is evaluated in the fc1
    epilogue, rounded once to bf16 for storage), the backward multiplies the bf16 upstream gradient by it."""

    @staticmethod
    def forward(ctx, x):
        cdf = 0.5 * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))
        pdf = torch.exp(-0.5 * x * x) * (1.0 / math.sqrt(2.0 * math.pi))
        ctx.save_for_backward((cdf + x * pdf).to(torch.bfloat16).to(torch.float32))
        return (x * cdf).to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (dg,) = ctx.saved_tensors
        return g.to(torch.bfloat16).to(torch.float32) * dg


def gelu_erf(x, lowp=False):
    """Exact (erf) GELU; reference transformer.py:38 `nn.GELU()` (approximate='none')."""
    if lowp:
        return _GeluStoredGradBF16.apply(x)
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def split_qkv(qkv: torch.Tensor, n_heads: int):
    """[B,N,3D] -> q,k,v each [B,H,N,dh].  Output-channel order is (qkv, head, dh):
    reference transformer.py:27 `rearrange(..., "b n (qkv h d) -> qkv b h n d")`."""
    B, N, D3 = qkv.shape
    dh = D3 // 3 // n_heads
    t = qkv.reshape(B, N, 3, n_heads, dh).permute(2, 0, 3, 1, 4)
    return t[0], t[1], t[2]


def sdpa(q, k, v, causal=False, lowp=False):
    """softmax(q k^T / sqrt(dh) + mask) v, mask = -inf strictly above the diagonal when
    causal.  Reference transformer.py:22-25,28 (F.scaled_dot_product_attention with the
    default scale and the additive mask buffer).  Dropout is 0 on every measured config."""
    dh = q.shape[-1]
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    if causal:
        n = q.shape[-2]
        mask = torch.triu(torch.ones(n, n, dtype=torch.bool), diagonal=1)
        s = s.masked_fill(mask, float("-inf"))
    s = s - s.amax(dim=-1, keepdim=True)
    e = torch.exp(s)
    denom = e.sum(dim=-1, keepdim=True)
    if lowp:
        # the kernels feed exp(s - max) to the matrix cores in bf16 and normalise afterwards
        # with the fp32 row sum
        return _r((_r(e, True) @ v) / denom, True)
    return (e / denom) @ v


def attention(x, sd, prefix, n_heads, causal=False, lowp=False):
    """Reference transformer.py:26-29 (Attention.forward): fused QKV linear, split,
    SDPA, merge heads `b h n d -> b n (h d)`.  There is NO output projection."""
    qkv = linear(x, sd[prefix + "qkv.weight"], sd[prefix + "qkv.bias"], lowp)
    q, k, v = split_qkv(qkv, n_heads)
    o = sdpa(q, k, v, causal, lowp)  # [B,H,N,dh]
    B, H, N, dh = o.shape
    return o.permute(0, 2, 1, 3).reshape(B, N, H * dh)


def mlp(x, sd, prefix, lowp=False):
    """Reference transformer.py:36-41: Linear(D,4D) -> GELU -> Linear(4D,D) -> Dropout(0)."""
    h = linear(x, sd[prefix + "0.weight"], sd[prefix + "0.bias"], lowp)
    h = gelu_erf(h, lowp)
    return linear(h, sd[prefix + "2.weight"], sd[prefix + "2.bias"], lowp)


def transformer_layer(x, sd, prefix, n_heads, causal=False, lowp=False):
    """Reference transformer.py:42-45: x = x + attn(LN(x)); x = x + mlp(LN(x)).
    The residual stream stays fp32 (fp32 + bf16 -> fp32 under autocast)."""
    x = x + attention(layer_norm(x), sd, prefix + "multi_attn.", n_heads, causal, lowp)
    x = x + mlp(layer_norm(x), sd, prefix + "mlp.", lowp)
    return x


def transformer(x, sd, prefix, n_layers, n_heads, causal=False, lowp=False):
    """Reference transformer.py:52-54: sequential layers, no final norm."""
    for i in range(n_layers):
        x = transformer_layer(x, sd, f"{prefix}layers.{i}.", n_heads, causal, lowp)
    return x


def patchify(images: torch.Tensor, p: int) -> torch.Tensor:
    """[B,C,Hh,Ww] -> [B, n_patches, C*p*p]; patches row-major over (h,w), patch vector in
    (c,kh,kw) order — the contraction order of Conv2d(kernel=stride=p), train_vit.py:34,39-40."""
    B, C, Hh, Ww = images.shape
    gh, gw = Hh // p, Ww // p
    t = images.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)
    return t.reshape(B, gh * gw, C * p * p)


def vit_embed(images, sd, prefix, cfg: OracleViTConfig, lowp=False):
    """Reference train_vit.py:38-44 (ViT.forward prologue): conv patchify, + pos_emb on the
    patch tokens only, learned extra tokens PREPENDED (indices 0..extra-1)."""
    D = cfg.n_embd
    w = sd[prefix + "patch_proj.weight"].reshape(D, -1)  # [D, C*p*p]
    tok = linear(patchify(images, cfg.patch_size), w, sd[prefix + "patch_proj.bias"], lowp)
    tok = tok + sd[prefix + "pos_emb.weight"][: cfg.n_patches]
    extra = sd[prefix + "extra_emb.weight"].unsqueeze(0).expand(images.shape[0], -1, -1)
    return torch.cat([extra, tok], dim=1)


def vit(images, sd, prefix, cfg: OracleViTConfig, lowp=False):
    """Reference train_vit.py:38-45 (ViT.forward)."""
    x = vit_embed(images, sd, prefix, cfg, lowp)
    return transformer(x, sd, prefix + "transformer.", cfg.n_layers, cfg.n_heads, False, lowp)


def vit_classifier(images, sd, cfg: OracleViTConfig, lowp=False):
    """Reference train_vit.py:53: head(vit(x)[:, 0])."""
    feat = vit(images, sd, "vit.", cfg, lowp)[:, 0]
    return linear(feat, sd["head.weight"], sd["head.bias"], lowp)


def cross_entropy(logits, labels):
    """Mean softmax cross-entropy; reference train_vit.py:81,102 (nn.CrossEntropyLoss)."""
    z = logits.float()
    lse = torch.logsumexp(z, dim=-1)
    return (lse - z.gather(1, labels[:, None]).squeeze(1)).mean()


# --------------------------------------------------------------------------------------
# 1-D / 2-D VQ tokenizers on top of the ViT
# --------------------------------------------------------------------------------------
def vq_quantize(x, codebook):
    """Reference train_titok.py:50-59 / train_vit_vqgan.py:49-58: L2-normalise the latents, nearest
    NORMALISED code by Euclidean distance, look up the RAW code, codebook + 0.25 commitment MSE,
    straight-through output."""
    xn = x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    en = codebook / codebook.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    d2 = (xn.detach().unsqueeze(-2) - en.detach()).pow(2).sum(-1)        # [..., K]
    idx = d2.argmin(dim=-1)
    q = codebook[idx]
    qloss = (q - xn.detach()).pow(2).mean() + 0.25 * (q.detach() - xn).pow(2).mean()
    return xn + (q - xn).detach(), idx, qloss


def pixel_shuffle(tokens, grid, p):
    """'b (h w) c -> b c h w' then 'b (p1 p2 c) h w -> b c (h p1) (w p2)' (train_titok.py:73-75)."""
    B, _, F = tokens.shape
    c = F // (p * p)
    return tokens.reshape(B, grid, grid, p, p, c).permute(0, 5, 1, 3, 2, 4).reshape(B, c, grid * p, grid * p)


def tokenizer_forward(images, sd, enc, quant, dec, enc_cfg: OracleViTConfig, dec_cfg: OracleViTConfig, keep_enc, keep_dec,
                      grid, patch, lowp=False, indices=None):
    """TiTok (train_titok.py:40-43,69-77,89-93) and ViT-VQGAN (train_vit_vqgan.py:40-43,67-74,86-90):
    ViT encoder -> Linear -> VQ -> Linear -> 'b h c -> b c h 1' -> ViT decoder (1x1 patches) ->
    1x1 conv -> pixel shuffle.  keep_* = how many leading tokens each ViT output keeps (None = all)."""
    tok = vit(images, sd, enc + "vit.", enc_cfg, lowp)
    if keep_enc is not None:
        tok = tok[:, :keep_enc]
    latents = linear(tok, sd[enc + "proj.weight"], sd[enc + "proj.bias"], lowp)
    quantized, idx, qloss = vq_quantize(latents, sd[quant + "codebook.weight"])
    if indices is not None:                                   # decode a prescribed code sequence instead
        quantized = sd[quant + "codebook.weight"][indices]
    z = linear(quantized, sd[dec + "quant_proj.weight"], sd[dec + "quant_proj.bias"], lowp)
    z = z.transpose(1, 2).unsqueeze(-1)                       # [b, c, h, 1]
    out = vit(z, sd, dec + "vit.", dec_cfg, lowp)
    if keep_dec is not None:
        out = out[:, :keep_dec]
    w = sd[dec + "embd_proj.weight"]
    y = linear(out, w.reshape(w.shape[0], -1), sd[dec + "embd_proj.bias"], lowp)
    return pixel_shuffle(y, grid, patch), idx, qloss, latents


# --------------------------------------------------------------------------------------
# blocks.py surface: affine LayerNorm, attention with output projection
# --------------------------------------------------------------------------------------
def layer_norm_affine(x, w, b):
    """nn.LayerNorm(d) with weight/bias, eps 1e-5 (reference blocks.py:43,48,179,184)."""
    return layer_norm(x) * w + b


def attention_proj(x, wqkv, bqkv, wo, bo, n_heads, lowp=False):
    """qkv Linear -> SDPA -> output projection, batch-first [B,N,D]: reference blocks.Attention
    (blocks.py:95-121) and nn.MultiheadAttention as used by blocks.py:56-60 (same (qkv, head, dh) packing)."""
    qkv = linear(x, wqkv, bqkv, lowp)
    q, k, v = split_qkv(qkv, n_heads)
    o = sdpa(q, k, v, False, lowp)
    B, H, N, dh = o.shape
    return linear(o.permute(0, 2, 1, 3).reshape(B, N, H * dh), wo, bo, lowp)


def mlp2(x, w1, b1, w2, b2, lowp=False):
    """fc1 -> erf-GELU -> fc2 (reference blocks.Mlp blocks.py:165-171; mlp c_fc/gelu/c_proj blocks.py:50-54)."""
    return linear(gelu_erf(linear(x, w1, b1, lowp), lowp), w2, b2, lowp)


def residual_attention_block(x_lnd, sd, n_heads, lowp=False):
    """Reference blocks.ResidualAttentionBlock.forward (blocks.py:62-70) on sequence-first [L,N,D] input."""
    x = x_lnd.transpose(0, 1)
    x = x + attention_proj(layer_norm_affine(x, sd["ln_1.weight"], sd["ln_1.bias"]), sd["attn.in_proj_weight"], sd["attn.in_proj_bias"],
                           sd["attn.out_proj.weight"], sd["attn.out_proj.bias"], n_heads, lowp)
    if "mlp.c_fc.weight" in sd:
        x = x + mlp2(layer_norm_affine(x, sd["ln_2.weight"], sd["ln_2.bias"]), sd["mlp.c_fc.weight"], sd["mlp.c_fc.bias"],
                     sd["mlp.c_proj.weight"], sd["mlp.c_proj.bias"], lowp)
    return x.transpose(0, 1)


def uvit_block(x, sd, n_heads, skip=None, lowp=False):
    """Reference blocks.UViTBlock._forward (blocks.py:196-201)."""
    if skip is not None:
        x = linear(torch.cat([x, skip], dim=-1), sd["skip_linear.weight"], sd["skip_linear.bias"], lowp).float()
    x = x + attention_proj(layer_norm_affine(x, sd["norm1.weight"], sd["norm1.bias"]), sd["attn.qkv.weight"], sd.get("attn.qkv.bias"),
                           sd["attn.proj.weight"], sd["attn.proj.bias"], n_heads, lowp)
    return x + mlp2(layer_norm_affine(x, sd["norm2.weight"], sd["norm2.bias"]), sd["mlp.fc1.weight"], sd["mlp.fc1.bias"],
                    sd["mlp.fc2.weight"], sd["mlp.fc2.bias"], lowp)


# --------------------------------------------------------------------------------------
# blocks.py tokenizer wrappers (blocks.py:209, 286, 365, 406)
# --------------------------------------------------------------------------------------
def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def conv3x3_same(x, w, b):
    """nn.Conv2d(Cin, Cout, 3, padding=1) as nine shifted channel mixes (reference blocks.py:333)."""
    B, C, H, W = x.shape
    xp = torch.zeros((B, C, H + 2, W + 2), dtype=x.dtype)
    xp[:, :, 1:H + 1, 1:W + 1] = x
    y = b.view(1, -1, 1, 1).expand(B, -1, H, W)
    for kh in range(3):
        for kw in range(3):
            y = y + torch.einsum("oi,bihw->bohw", w[:, :, kh, kw], xp[:, :, kh:kh + H, kw:kw + W])
    return y


def _blocks_stack(x, sd, n_layers, n_heads, lowp):
    """`for i: x = self.transformer[i](x)` between the two LND permutes (blocks.py:270-273); run batch-first."""
    x = x.transpose(0, 1)
    for i in range(n_layers):
        x = residual_attention_block(x, _sub(sd, f"transformer.{i}."), n_heads, lowp)
    return x.transpose(0, 1)


def titok_block_encoder(pixels, latent_tokens, sd, patch, grid, n_layers, n_heads, lowp=False):
    """Reference blocks.TiTokEncoder.forward (blocks.py:256-282) -> [B, token_size, 1, n_latents]."""
    B = pixels.shape[0]
    w = sd["patch_embed.weight"]
    x = linear(patchify(pixels, patch), w.reshape(w.shape[0], -1), sd["patch_embed.bias"], lowp).float()
    x = torch.cat([sd["class_embedding"].unsqueeze(0).expand(B, -1, -1), x], dim=1) + sd["positional_embedding"]
    lat = (latent_tokens + sd["latent_token_positional_embedding"]).unsqueeze(0).expand(B, -1, -1)
    x = torch.cat([x, lat], dim=1)
    x = layer_norm_affine(x, sd["ln_pre.weight"], sd["ln_pre.bias"])
    x = _blocks_stack(x, sd, n_layers, n_heads, lowp)
    lat = layer_norm_affine(x[:, 1 + grid * grid:], sd["ln_post.weight"], sd["ln_post.bias"])
    cw = sd["conv_out.weight"]
    z = linear(lat, cw.reshape(cw.shape[0], -1), sd["conv_out.bias"], lowp).float()      # 1x1 conv over the width axis
    return z.permute(0, 2, 1).reshape(B, cw.shape[0], 1, lat.shape[1])


def titok_block_decoder(z_quantized, sd, patch, grid, n_layers, n_heads, text_guidance=None, lowp=False):
    """Reference blocks.TiTokDecoder.forward (blocks.py:335-356); with `text_guidance` [B, T, E] it is
    blocks.TATiTokDecoder.forward (blocks.py:369-403)."""
    B, C, H, Wl = z_quantized.shape
    x = z_quantized.reshape(B, C * H, Wl).permute(0, 2, 1)
    x = linear(x, sd["decoder_embed.weight"], sd["decoder_embed.bias"], lowp).float()
    mask = sd["mask_token"].expand(B, grid * grid, -1)
    mask = torch.cat([sd["class_embedding"].unsqueeze(0).expand(B, -1, -1), mask], dim=1) + sd["positional_embedding"]
    x = torch.cat([mask, x + sd["latent_token_positional_embedding"][:Wl]], dim=1)
    if text_guidance is not None:
        t = linear(text_guidance, sd["text_guidance_proj.weight"], sd["text_guidance_proj.bias"], lowp).float()
        x = torch.cat([x, t + sd["text_guidance_positional_embedding"]], dim=1)
    x = layer_norm_affine(x, sd["ln_pre.weight"], sd["ln_pre.bias"])
    x = _blocks_stack(x, sd, n_layers, n_heads, lowp)
    x = layer_norm_affine(x[:, 1:1 + grid * grid], sd["ln_post.weight"], sd["ln_post.bias"])
    fw = sd["ffn.0.weight"]
    y = linear(x, fw.reshape(fw.shape[0], -1), sd["ffn.0.bias"], lowp).float()
    return conv3x3_same(pixel_shuffle(y, grid, patch), sd["conv_out.weight"], sd["conv_out.bias"])


def vector_quantizer(z, codebook, commitment_cost=0.25, use_l2_norm=False):
    """Reference blocks.VectorQuantizer.forward (blocks.py:430-495) without the clustering update:
    -> (z_quantized [b,c,h,w] straight-through, quantizer_loss, commitment_loss, codebook_loss, indices [b,h,w])."""
    zc = z.float().permute(0, 2, 3, 1)
    zf = zc.reshape(-1, zc.shape[-1])
    unit = (lambda t: t / t.norm(dim=-1, keepdim=True).clamp_min(1e-12)) if use_l2_norm else (lambda t: t)
    zn, en = unit(zf), unit(codebook)
    d = zn.pow(2).sum(1, keepdim=True) + en.pow(2).sum(1) - 2 * zn @ en.t()
    idx = d.detach().argmin(dim=1)
    zq = unit(codebook[idx]).view(zc.shape)
    zc = unit(zc)
    commitment = commitment_cost * (zq.detach() - zc).pow(2).mean()
    cb_loss = (zq - zc.detach()).pow(2).mean()
    out = (zc + (zq - zc).detach()).permute(0, 3, 1, 2)
    return out, commitment + cb_loss, commitment, cb_loss, idx.view(zc.shape[:3])


def vq_cluster_update(z, codebook, embed_prob, decay=0.99, use_l2_norm=False):
    """The `clustering_vq` training-time codebook refresh of blocks.py:454-479 for ONE process (gather = identity):
    -> (new codebook, new embed_prob)."""
    zc = z.float().permute(0, 2, 3, 1)
    zf = zc.reshape(-1, zc.shape[-1])
    unit = (lambda t: t / t.norm(dim=-1, keepdim=True).clamp_min(1e-12)) if use_l2_norm else (lambda t: t)
    zn, en = unit(zf), unit(codebook)
    d = zn.pow(2).sum(1, keepdim=True) + en.pow(2).sum(1) - 2 * zn @ en.t()
    K = codebook.shape[0]
    probs = torch.bincount(d.argmin(dim=1), minlength=K).float() / zf.shape[0]
    embed_prob = embed_prob * decay + probs * (1 - decay)
    feat = zf[d.argmin(dim=0)]
    w = torch.exp(-(embed_prob * K * 10) / (1 - decay) - 1e-3).unsqueeze(1)
    return codebook * (1 - w) + feat * w, embed_prob


# --------------------------------------------------------------------------------------
# LR schedule (reference utils.py:5-9), closed form
# --------------------------------------------------------------------------------------
def lr_at(step: int, base_lr: float, warmup_steps: int, train_steps: int, min_lr: float) -> float:
    """Learning rate in effect for optimiser step number `step` (0-based) under the
    reference's SequentialLR[warmup LambdaLR, CosineAnnealingLR(T_max=train_steps), const]
    with milestones [warmup_steps, train_steps].  Quirks reproduced on purpose (SURVEY.md
    section 5): the cosine starts counting at the warmup milestone, so only
    (train_steps - warmup_steps)/train_steps of it elapses, and at `train_steps` the rate
    jumps back to base_lr."""
    if step < warmup_steps:
        return base_lr * min(1.0, step / warmup_steps)
    if step < train_steps:
        t = step - warmup_steps
        return min_lr + (base_lr - min_lr) * (1 + math.cos(math.pi * t / train_steps)) / 2
    return base_lr


# --------------------------------------------------------------------------------------
# convenience: forward + backward returning gradients keyed like the state_dict
# --------------------------------------------------------------------------------------
def classifier_loss_and_grads(images, labels, sd, cfg: OracleViTConfig, lowp=False):
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    logits = vit_classifier(images, leaves, cfg, lowp)
    loss = cross_entropy(logits, labels)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return logits.detach(), loss.detach(), dict(zip(leaves.keys(), grads))


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    """||a-b|| / ||b|| in fp64 — the comparison metric used by every parity test."""
    a = a.detach().double().flatten()
    b = b.detach().double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
