"""MI355X-native drop-in for the transformer blocks of the reference's `blocks` module
(reference blocks.py:32-201: ResidualAttentionBlock, Attention, DropPath, Mlp, UViTBlock — SURVEY.md
section 8f row 4).  Same class names, constructor signatures, attributes and state_dict keys
(`ln_1/ln_2`, `attn.in_proj_weight/in_proj_bias/out_proj.*`, `mlp.c_fc/c_proj`; `norm1/norm2`,
`attn.qkv/proj`, `mlp.fc1/fc2`, `skip_linear`), forward/backward on libvitamd kernels in the bf16 dtype
flow of vitamd/functions.py (affine LayerNorm kernel, fused-QKV attention, GEMMs with fused bias /
GELU / residual epilogues).

Differences from the reference, stated: head_dim must be 64; the SDPA runs in bf16 (blocks.Attention
upcasts q,k,v to fp32 first, blocks.py:100); the nn.Dropout / DropPath layers only accept rate 0 on this
path (blocks.Attention's 'flash' mode ignores attn_drop anyway, blocks.py:98-103).  The tokenizer
wrappers of blocks.py:208-505 (TiTokEncoder/TiTokDecoder/TATiTokDecoder/VectorQuantizer, used only by
the out-of-scope train_tatitok.py; they need a 3x3 convolution and an OmegaConf config tree) are not
provided — the transformer.py-based TiTok lives in train_titok.py."""
from collections import OrderedDict

import torch
import torch.nn as nn

from vitamd.block_functions import AttnProjFn, BlockFn, MlpFn
from vitamd.functions import linear

ATTENTION_MODE = "hip"   # the reference picks 'flash' / 'xformers' / 'math' at import (blocks.py:72-81)
print(f"attention mode is {ATTENTION_MODE}")


def _need_gelu(act_layer):
    if act_layer is not nn.GELU:
        raise NotImplementedError("only the erf-GELU activation is fused into the GEMM epilogues")


def _need_zero(rate, what):
    if rate != 0.0:
        raise NotImplementedError(f"{what} > 0 is not implemented for the blocks.py surface")


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
        has_mlp = self.mlp_ratio > 0
        xt = x.transpose(0, 1).contiguous()                       # LND -> NLD: one row per token, batch-major
        y = BlockFn.apply(xt, self.ln_1.weight, self.ln_1.bias, self.attn.in_proj_weight, self.attn.in_proj_bias,
                          self.attn.out_proj.weight, self.attn.out_proj.bias,
                          self.ln_2.weight if has_mlp else None, self.ln_2.bias if has_mlp else None,
                          self.mlp.c_fc.weight if has_mlp else None, self.mlp.c_fc.bias if has_mlp else None,
                          self.mlp.c_proj.weight if has_mlp else None, self.mlp.c_proj.bias if has_mlp else None,
                          self.n_head, has_mlp)
        return y.transpose(0, 1)


class Attention(nn.Module):
    """U-ViT attention: qkv Linear (bias optional) -> SDPA -> proj Linear (reference blocks.py:84-121)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = _heads(dim, num_heads)
        head_dim = dim // num_heads
        if qk_scale is not None and abs(qk_scale - head_dim ** -0.5) > 1e-12:
            raise NotImplementedError("only the default qk scale head_dim**-0.5 is built into the kernels")
        self.scale = qk_scale or head_dim ** -0.5
        _need_zero(proj_drop, "proj_drop")
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)     # unused by the reference's 'flash' mode as well
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        return AttnProjFn.apply(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, self.num_heads)


def drop_path(x, drop_prob: float = 0., training: bool = False):
    if drop_prob == 0. or not training:
        return x
    raise NotImplementedError("drop_path > 0 is not implemented for the blocks.py surface")


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
        _need_zero(drop, "drop")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return MlpFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)


class UViTBlock(nn.Module):
    """Pre-LN block with optional long-skip concat Linear (reference blocks.py:174-201)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, skip=False, use_checkpoint=False):
        super().__init__()
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("only nn.LayerNorm is supported")
        _need_zero(drop_path, "drop_path")
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)
        self.skip_linear = nn.Linear(2 * dim, dim) if skip else None
        self.use_checkpoint = use_checkpoint       # accepted for signature parity; activations fit in HBM, nothing is recomputed

    def forward(self, x, skip=None):
        return self._forward(x, skip)

    def _forward(self, x, skip=None):
        if self.skip_linear is not None:
            x = linear(torch.cat([x, skip], dim=-1), self.skip_linear.weight, self.skip_linear.bias)
        return BlockFn.apply(x, self.norm1.weight, self.norm1.bias, self.attn.qkv.weight, self.attn.qkv.bias,
                             self.attn.proj.weight, self.attn.proj.bias, self.norm2.weight, self.norm2.bias,
                             self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias,
                             self.attn.num_heads, True)
