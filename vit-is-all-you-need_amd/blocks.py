"""MI355X-native drop-in for the transformer blocks of the reference's `blocks` module
(reference blocks.py:32-201: ResidualAttentionBlock, Attention, DropPath, Mlp, UViTBlock — SURVEY.md
section 8f row 4).  Same class names, constructor signatures, attributes and state_dict keys
(`ln_1/ln_2`, `attn.in_proj_weight/in_proj_bias/out_proj.*`, `mlp.c_fc/c_proj`; `norm1/norm2`,
`attn.qkv/proj`, `mlp.fc1/fc2`, `skip_linear`), forward/backward on libvitamd kernels in the bf16 dtype
flow of vitamd/functions.py (affine LayerNorm kernel, fused-QKV attention, GEMMs with fused bias /
GELU / residual epilogues).

Differences from the reference, stated: head_dim must be 64; the SDPA runs in bf16 (blocks.Attention
upcasts q,k,v to fp32 first, blocks.py:100); non-zero drop / proj_drop / drop_path rates are accepted and are the
identity in eval mode, but TRAINING with them raises (blocks.Attention's 'flash' mode ignores attn_drop anyway,
blocks.py:98-103).

The tokenizer wrappers of blocks.py:208-505 (TiTokEncoder / TiTokDecoder / TATiTokDecoder / VectorQuantizer)
are here as well, over the same kernels: patch-embed GEMM with fused bias + positional embedding, fp32 affine
LayerNorm kernel for ln_pre / ln_post, BlockFn per layer (batch-major, so the reference's two LND permutes per
model disappear), padded-Linear GEMMs for the 1x1 convolutions / decoder_embed / text projection, the
3x3 `conv_out` kernel and the nearest-code search kernel.  Token concatenation, slicing, the pixel-shuffle
rearrange and the O(tokens x token_size) elementwise terms of the quantiser losses are torch device ops.
`config` is duck-typed (attribute access, plus `.model.vq_model.get(...)` for TATiTokDecoder) - no OmegaConf."""
from collections import OrderedDict

from typing import Mapping, Text, Tuple

import torch
import torch.nn as nn
from einops.layers.torch import Rearrange

from vitamd import ops
from vitamd.block_functions import AttnProjFn, BlockFn, Conv3x3Fn, DropPathFn, LayerNormAffineFn, MlpFn
from vitamd.functions import PatchEmbedFn, linear

ATTENTION_MODE = "hip"   # the reference picks 'flash' / 'xformers' / 'math' at import (blocks.py:72-81)
print(f"attention mode is {ATTENTION_MODE}")


def _need_gelu(act_layer):
    if act_layer is not nn.GELU:
        raise NotImplementedError("only the erf-GELU activation is fused into the GEMM epilogues")


def _heads(dim, num_heads):
    if dim % num_heads or dim // num_heads != 64:
        raise NotImplementedError("the attention kernels are built for head_dim 64")
    return num_heads


class ResidualAttentionBlock(nn.Module):
    """open_clip-style block on SEQUENCE-FIRST input [L, N, D] (reference blocks.py:32-70)."""

    def __init__(self, d_model, n_head, mlp_ratio=4.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("only nn.LayerNorm is supported")
        _need_gelu(act_layer)
        self.n_head = _heads(d_model, n_head)
        self.ln_1 = norm_layer(d_model)
        self.attn = nn.MultiheadAttention(d_model, n_head)      # parameter container: in_proj_weight/bias, out_proj
        self.mlp_ratio = mlp_ratio
        if mlp_ratio > 0:
            self.ln_2 = norm_layer(d_model)
            mlp_width = int(d_model * mlp_ratio)
            self.mlp = nn.Sequential(OrderedDict([
                ("c_fc", nn.Linear(d_model, mlp_width)),
                ("gelu", act_layer()),
                ("c_proj", nn.Linear(mlp_width, d_model)),
            ]))

    def attention(self, x: torch.Tensor):
        xt = x.transpose(0, 1).contiguous()
        y = AttnProjFn.apply(xt, self.attn.in_proj_weight, self.attn.in_proj_bias, self.attn.out_proj.weight,
                             self.attn.out_proj.bias, self.n_head)
        return y.transpose(0, 1)

    def forward(self, x: torch.Tensor):
        return self.forward_nld(x.transpose(0, 1).contiguous()).transpose(0, 1)   # LND -> NLD: one row per token, batch-major

    def forward_nld(self, xt: torch.Tensor):
        """Same block on batch-first [N, L, D] input (what the kernels want; used by the tokenizer wrappers below)."""
        has_mlp = self.mlp_ratio > 0
        y = BlockFn.apply(xt, self.ln_1.weight, self.ln_1.bias, self.attn.in_proj_weight, self.attn.in_proj_bias,
                          self.attn.out_proj.weight, self.attn.out_proj.bias,
                          self.ln_2.weight if has_mlp else None, self.ln_2.bias if has_mlp else None,
                          self.mlp.c_fc.weight if has_mlp else None, self.mlp.c_fc.bias if has_mlp else None,
                          self.mlp.c_proj.weight if has_mlp else None, self.mlp.c_proj.bias if has_mlp else None,
                          self.n_head, has_mlp)
        return y


class Attention(nn.Module):
    """U-ViT attention: qkv Linear (bias optional) -> SDPA -> proj Linear (reference blocks.py:84-121)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = _heads(dim, num_heads)
        head_dim = dim // num_heads
        if qk_scale is not None and abs(qk_scale - head_dim ** -0.5) > 1e-12:
            raise NotImplementedError("only the default qk scale head_dim**-0.5 is built into the kernels")
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)     # unused by the reference's 'flash' mode as well
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x, _drop_path=0.0):
        # attn_drop is unused by the reference's 'flash' mode as well; proj_drop is live in training (reference blocks.py:118)
        return AttnProjFn.apply(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, self.num_heads,
                                float(self.proj_drop.p) if self.training else 0.0, float(_drop_path))


def drop_path(x, drop_prob: float = 0., training: bool = False):
    """Stochastic depth per sample (reference blocks.py:124-139): identity in eval mode or at rate 0; in training a whole sample's
    branch is zeroed with probability drop_prob and the survivors are scaled by 1/(1 - drop_prob) - on the dropout kernel."""
    if drop_prob == 0. or not training:
        return x
    return DropPathFn.apply(x, float(drop_prob))


class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob or 0., self.training)


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2 (reference blocks.py:155-171)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        _need_gelu(act_layer)
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x, _drop_path=0.0):
        return MlpFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias,
                           float(self.drop.p) if self.training else 0.0, float(_drop_path))


class UViTBlock(nn.Module):
    """Pre-LN block with optional long-skip concat Linear (reference blocks.py:174-201)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, skip=False, use_checkpoint=False):
        super().__init__()
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("only nn.LayerNorm is supported")
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()     # as the reference (blocks.py:182)
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)
        self.skip_linear = nn.Linear(2 * dim, dim) if skip else None
        self.use_checkpoint = use_checkpoint

    def forward(self, x, skip=None):
        if self.use_checkpoint and torch.is_grad_enabled():
            # activation checkpointing (reference blocks.py:188-192): only the block input is kept, the forward is re-run in backward.
            # The dropout seeds come from torch's CPU generator, whose state checkpoint() restores for the re-run: same masks.
            from torch.utils.checkpoint import checkpoint
            return checkpoint(self._forward, x, skip, use_reentrant=False)
        return self._forward(x, skip)

    def _forward(self, x, skip=None):
        if self.skip_linear is not None:
            x = linear(torch.cat([x, skip], dim=-1), self.skip_linear.weight, self.skip_linear.bias)
        p_path = float(getattr(self.drop_path, "drop_prob", 0.) or 0.) if self.training else 0.0
        if self.training and (p_path or self.mlp.drop.p or self.attn.proj_drop.p):
            # training with drop rates (reference blocks.py:194-201): x + drop_path(attn(norm1(x))), then x + drop_path(mlp(norm2(x))),
            # the branches on the drop-aware Functions (masks from the stateless hash, regenerated in backward); the two residual adds
            # and the LayerNorms run un-fused here (fp32 stream)
            x = x + self.attn(_ln(self.norm1, x), _drop_path=p_path)
            return x + self.mlp(_ln(self.norm2, x), _drop_path=p_path)
        return BlockFn.apply(x, self.norm1.weight, self.norm1.bias, self.attn.qkv.weight, self.attn.qkv.bias,
                             self.attn.proj.weight, self.attn.proj.bias, self.norm2.weight, self.norm2.bias,
                             self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias,
                             self.attn.num_heads, True)


# ------------------------------------------------------------------------------------------------
# tokenizer wrappers (reference blocks.py:204-505)
# ------------------------------------------------------------------------------------------------
_WIDTH = {"small": 512, "base": 768, "large": 1024}
_LAYERS = {"small": 8, "base": 12, "large": 24}
_HEADS = {"small": 8, "base": 12, "large": 16}


def _expand_token(token, batch_size: int):
    return token.unsqueeze(0).expand(batch_size, -1, -1)


def _ln(mod, x):
    return LayerNormAffineFn.apply(x.contiguous(), mod.weight, mod.bias, mod.eps)


def _conv1x1_tokens(conv, tokens):
    """nn.Conv2d(kernel_size=1) parameters applied to token-major [B, T, C] input -> [B, T, out]."""
    return linear(tokens, conv.weight.view(conv.weight.shape[0], -1), conv.bias)


class TiTokEncoder(nn.Module):
    """Image patches + class token + learned latent tokens through `num_layers` ResidualAttentionBlocks; the
    latent tokens' outputs are projected to `token_size` (reference blocks.py:208-282).  Parameter names as there."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.image_size = config.image_size
        self.patch_size = config.patch_size
        self.grid_size = self.image_size // self.patch_size
        self.model_size = config.transformer
        self.num_latent_tokens = config.latent_tokens
        self.token_size = config.latent_dim
        self.width = _WIDTH[self.model_size]
        self.num_layers = _LAYERS[self.model_size]
        self.num_heads = _HEADS[self.model_size]

        self.patch_embed = nn.Conv2d(in_channels=3, out_channels=self.width, kernel_size=self.patch_size, stride=self.patch_size, bias=True)
        scale = self.width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(1, self.width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid_size ** 2 + 1, self.width))
        self.latent_token_positional_embedding = nn.Parameter(scale * torch.randn(self.num_latent_tokens, self.width))
        self.ln_pre = nn.LayerNorm(self.width)
        self.transformer = nn.ModuleList()
        for _ in range(self.num_layers):
            self.transformer.append(ResidualAttentionBlock(self.width, self.num_heads, mlp_ratio=4.0))
        self.ln_post = nn.LayerNorm(self.width)
        self.conv_out = nn.Conv2d(self.width, self.token_size, kernel_size=1, bias=True)

    def forward(self, pixel_values, latent_tokens):
        batch_size = pixel_values.shape[0]
        g2 = self.grid_size ** 2
        pos = self.positional_embedding
        # conv patchify + bias + positional embedding in one GEMM; the class token (+ its position) is row 0
        x = PatchEmbedFn.apply(pixel_values, self.patch_embed.weight, self.patch_embed.bias, pos[1:], self.class_embedding + pos[:1],
                               self.patch_size, g2)                                   # [B, 1 + grid**2, width] fp32
        latent_tokens = _expand_token(latent_tokens.to(x.dtype) + self.latent_token_positional_embedding, batch_size)
        x = torch.cat([x, latent_tokens], dim=1)
        x = _ln(self.ln_pre, x)
        for blk in self.transformer:
            x = blk.forward_nld(x)
        latent_tokens = _ln(self.ln_post, x[:, 1 + g2:])
        z = _conv1x1_tokens(self.conv_out, latent_tokens)                            # [B, latents, token_size]
        return z.permute(0, 2, 1).reshape(batch_size, self.token_size, 1, self.num_latent_tokens)


class TiTokDecoder(nn.Module):
    """Quantised latent tokens + one mask token per image patch -> pixels (reference blocks.py:285-356)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.image_size = config.image_size
        self.patch_size = config.patch_size
        self.grid_size = self.image_size // self.patch_size
        self.model_size = config.transformer
        self.num_latent_tokens = config.latent_tokens
        self.token_size = config.latent_dim
        self.width = _WIDTH[self.model_size]
        self.num_layers = _LAYERS[self.model_size]
        self.num_heads = _HEADS[self.model_size]

        self.decoder_embed = nn.Linear(self.token_size, self.width, bias=True)
        scale = self.width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(1, self.width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid_size ** 2 + 1, self.width))
        self.mask_token = nn.Parameter(scale * torch.randn(1, 1, self.width))
        self.latent_token_positional_embedding = nn.Parameter(scale * torch.randn(self.num_latent_tokens, self.width))
        self.ln_pre = nn.LayerNorm(self.width)
        self.transformer = nn.ModuleList()
        for _ in range(self.num_layers):
            self.transformer.append(ResidualAttentionBlock(self.width, self.num_heads, mlp_ratio=4.0))
        self.ln_post = nn.LayerNorm(self.width)
        # parameter containers with the reference's names (ffn.0.*, conv_out.*); applied through the kernels below
        self.ffn = nn.Sequential(
            nn.Conv2d(self.width, self.patch_size * self.patch_size * 3, 1, padding=0, bias=True),
            Rearrange('b (p1 p2 c) h w -> b c (h p1) (w p2)', p1=self.patch_size, p2=self.patch_size),)
        self.conv_out = nn.Conv2d(3, 3, 3, padding=1, bias=True)

    def _tokens(self, z_quantized):
        """decoder input sequence (reference blocks.py:336-348): [class token, grid^2 mask tokens] + positions, then the embedded
        latent tokens + their own positions"""
        batch, channels, one, n_latent = z_quantized.shape
        assert one == 1 and n_latent == self.num_latent_tokens, f"{one}, {n_latent}, {self.num_latent_tokens}"
        latents = linear(z_quantized.reshape(batch, channels, n_latent).transpose(1, 2), self.decoder_embed.weight, self.decoder_embed.bias)
        latents = latents + self.latent_token_positional_embedding[:n_latent]
        canvas = torch.cat([self.class_embedding.reshape(1, 1, -1), self.mask_token.expand(1, self.grid_size ** 2, -1)], dim=1)
        canvas = (canvas + self.positional_embedding).to(latents.dtype).expand(batch, -1, -1)
        return torch.cat([canvas, latents], dim=1)

    def _pixels(self, x):
        batchsize, g, p = x.shape[0], self.grid_size, self.patch_size
        x = _ln(self.ln_pre, x)
        for blk in self.transformer:
            x = blk.forward_nld(x)
        x = _ln(self.ln_post, x[:, 1:1 + g * g])                                   # remove cls embed
        y = _conv1x1_tokens(self.ffn[0], x)                                         # [B, grid**2, p*p*3], token t = (h, w)
        img = y.view(batchsize, g, g, p, p, 3).permute(0, 5, 1, 3, 2, 4).reshape(batchsize, 3, g * p, g * p)
        return Conv3x3Fn.apply(img, self.conv_out.weight, self.conv_out.bias)

    def forward(self, z_quantized):
        return self._pixels(self._tokens(z_quantized))


class TATiTokDecoder(TiTokDecoder):
    """TiTokDecoder with projected text-guidance tokens appended to the sequence (reference blocks.py:359-403)."""

    def __init__(self, config):
        super().__init__(config)
        scale = self.width ** -0.5
        self.text_context_length = config.model.vq_model.get("text_context_length", 77)
        self.text_embed_dim = config.model.vq_model.get("text_embed_dim", 768)
        self.text_guidance_proj = nn.Linear(self.text_embed_dim, self.width)
        self.text_guidance_positional_embedding = nn.Parameter(scale * torch.randn(self.text_context_length, self.width))

    def forward(self, z_quantized, text_guidance):
        x = self._tokens(z_quantized)
        text_guidance = linear(text_guidance, self.text_guidance_proj.weight, self.text_guidance_proj.bias)
        text_guidance = text_guidance + self.text_guidance_positional_embedding
        return self._pixels(torch.cat([x, text_guidance], dim=1))


def gather(t):
    """All-gather along dim 0 across data-parallel ranks (identity in a single process).  The reference calls
    a `gather` it never defines or imports (blocks.py:457,466,467 - a NameError as written); this is the
    accelerate-style meaning the surrounding comments describe."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t.contiguous())
    return torch.cat(parts, dim=0)


class VectorQuantizer(torch.nn.Module):
    """Nearest-code quantiser with commitment / codebook losses and straight-through gradients, fp32
    (reference blocks.py:405-505).  The argmin over the codebook runs on the nearest-code kernel."""

    def __init__(self, codebook_size: int = 1024, token_size: int = 256, commitment_cost: float = 0.25,
                 use_l2_norm: bool = False, clustering_vq: bool = False):
        super().__init__()
        self.codebook_size = codebook_size
        self.token_size = token_size
        self.commitment_cost = commitment_cost
        self.embedding = torch.nn.Embedding(codebook_size, token_size)
        self.embedding.weight.data.uniform_(-1.0 / codebook_size, 1.0 / codebook_size)
        self.use_l2_norm = use_l2_norm
        self.clustering_vq = clustering_vq
        if clustering_vq:
            self.decay = 0.99
            self.register_buffer("embed_prob", torch.zeros(self.codebook_size))

    # ---- pieces of the quantiser (behaviour of reference blocks.py:429-505) ----
    def _unit(self, t):
        """tokens / codes on the unit sphere when the quantiser is the cosine (l2-normalised) kind"""
        return torch.nn.functional.normalize(t, dim=-1) if self.use_l2_norm else t

    def _lookup(self, codes_or_weights):
        """hard lookup for integer code ids [T]; soft lookup (weights [T, K] times the codebook) for a float matrix"""
        if codes_or_weights.dim() == 1:
            return self.embedding(codes_or_weights)
        if codes_or_weights.dim() == 2:
            return codes_or_weights @ self.embedding.weight
        raise NotImplementedError

    def get_codebook_entry(self, indices):
        return self._unit(self._lookup(indices))

    @torch.no_grad()
    def _refresh_codebook(self, code_ids, tokens_unit, tokens_raw):
        """clustering_vq (reference blocks.py:455-475): exponential usage statistics over the gathered batch, then every code is
        pulled towards its nearest sample with a weight that is ~1 for codes nobody uses and ~0 for busy ones"""
        if code_ids.dim() != 1:
            raise ValueError(f"min_encoding_indices in a wrong shape, {code_ids.shape}")
        ids_all = gather(code_ids)
        usage = torch.bincount(ids_all, minlength=self.codebook_size).float() / ids_all.shape[0]
        self.embed_prob.mul_(self.decay).add_(usage, alpha=1 - self.decay)
        samples_unit, samples_raw = gather(tokens_unit), gather(tokens_raw).detach()
        nearest_sample = ops.vq_nearest(self._unit(self.embedding.weight).detach().float().contiguous(), samples_unit)   # roles swapped
        pull = torch.exp(-(self.embed_prob * self.codebook_size * 10) / (1 - self.decay) - 1e-3)[:, None].expand(-1, self.token_size)
        self.embedding.weight.data = self.embedding.weight.data * (1 - pull) + samples_raw[nearest_sample] * pull

    def forward(self, z: torch.Tensor) -> Tuple[torch.Tensor, Mapping[Text, torch.Tensor]]:
        batch, _, rows, cols = z.shape
        x = z.float().permute(0, 2, 3, 1)                                  # channels last: one token per (b, h, w)
        tokens_raw = x.reshape(-1, x.shape[-1])
        tokens = self._unit(tokens_raw)
        tokens_c = tokens.detach().contiguous()
        codes_c = self._unit(self.embedding.weight).detach().float().contiguous()
        ids = ops.vq_nearest(tokens_c, codes_c)                            # argmin_k |token - code_k|^2 on the nearest-code kernel
        target = tokens.view(x.shape)                                      # what the losses compare against (normalised if cosine)
        picked = self.get_codebook_entry(ids).view(x.shape)
        mse = lambda u, v: ((u - v) ** 2).mean()
        commit = self.commitment_cost * mse(picked.detach(), target)
        codebook = mse(picked, target.detach())
        if self.clustering_vq and self.training:
            self._refresh_codebook(ids, tokens_c, tokens_raw)
        out = (target + (picked - target).detach()).permute(0, 3, 1, 2).contiguous()     # straight-through, back to b c h w
        info = {"quantizer_loss": commit + codebook, "commitment_loss": commit, "codebook_loss": codebook,
                "min_encoding_indices": ids.view(batch, rows, cols)}
        return out, info
