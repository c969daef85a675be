"""MI355X-native drop-in for the reference's `transformer` module (reference transformer.py:1-59).

Same public surface — TransformerConfig, Attention, TransformerLayer, Transformer, S/B/L,
transformer_configs — same constructor signatures, attribute names and state_dict keys
(`layers.{i}.multi_attn.qkv.{weight,bias}`, `layers.{i}.multi_attn.mask` when causal,
`layers.{i}.mlp.{0,2}.{weight,bias}`), so `from transformer import Transformer, transformer_configs`
in the reference's training scripts resolves here unchanged and reference checkpoints load.

What differs is everything underneath: forward and backward run as hand-written gfx950 kernels
from libvitamd.so (vitamd/functions.py strings them together) in the reference's autocast dtype
flow with bf16 as the low-precision type.  Inputs must live on a ROCm device; there is no CPU or
stock-PyTorch fallback.
"""
from dataclasses import dataclass

import torch
import torch.nn as nn

from vitamd.functions import AttentionFn, TransformerLayerFn, TransformerStackFn


@dataclass
class TransformerConfig:
    n_layers: int
    n_heads: int
    n_embd: int
    block_size: int
    causal: bool = False
    dropout: float = 0.0

    def __post_init__(self):
        self.head_dim = self.n_embd // self.n_heads


def _adopt(module: nn.Module, config: TransformerConfig):
    # the reference copies every config field onto the module (transformer.py:20,34,50)
    if "causal" not in config.__dict__:
        config.causal = False  # configs pickled before `causal` existed (transformer.py:19)
    for k, v in config.__dict__.items():
        setattr(module, k, v)


def _check_dropout(p: float):
    if not (0.0 <= p < 1.0):
        raise ValueError(f"dropout must be in [0, 1), got {p}")


class Attention(nn.Module):
    """Fused-QKV multi-head attention, NO output projection (reference transformer.py:16-29)."""

    def __init__(self, config: TransformerConfig):
        super().__init__()
        _adopt(self, config)
        self.qkv = nn.Linear(self.n_embd, self.n_embd * 3)
        if self.causal:
            mask = torch.triu(torch.ones(config.block_size, config.block_size), diagonal=1)
            self.register_buffer("mask", mask.masked_fill(mask == 1, float("-inf")))  # checkpoint-key parity

    def forward(self, x):
        _check_dropout(self.dropout)
        # like the reference, the SDPA dropout is applied in eval() as well (transformer.py:28 passes
        # dropout_p=self.dropout unconditionally)
        return AttentionFn.apply(x, self.qkv.weight, self.qkv.bias, self.n_heads, bool(self.causal), float(self.dropout))


class TransformerLayer(nn.Module):
    """Pre-LN block with non-affine LayerNorm and a 4x erf-GELU MLP (reference transformer.py:31-45)."""

    def __init__(self, config: TransformerConfig):
        super().__init__()
        _adopt(self, config)
        self.multi_attn = Attention(config)
        self.mlp = nn.Sequential(
            nn.Linear(self.n_embd, 4 * self.n_embd),
            nn.GELU(),
            nn.Linear(4 * self.n_embd, self.n_embd),
            nn.Dropout(self.dropout),
        )

    def _params(self):
        return (self.multi_attn.qkv.weight, self.multi_attn.qkv.bias, self.mlp[0].weight, self.mlp[0].bias,
                self.mlp[2].weight, self.mlp[2].bias)

    def forward(self, x):
        _check_dropout(self.dropout)
        p_mlp = float(self.dropout) if self.training else 0.0          # nn.Dropout in the MLP follows train/eval
        return TransformerLayerFn.apply(x, *self._params(), self.n_heads, bool(self.causal), float(self.dropout), p_mlp)


class Transformer(nn.Module):
    """Stack of layers, no final norm (reference transformer.py:47-54)."""

    def __init__(self, config: TransformerConfig):
        super().__init__()
        _adopt(self, config)
        self.layers = nn.ModuleList([TransformerLayer(config) for _ in range(config.n_layers)])

    def forward(self, x):
        _check_dropout(self.dropout)
        params = [p for layer in self.layers for p in layer._params()]
        p_mlp = float(self.dropout) if self.training else 0.0
        return TransformerStackFn.apply(x, self.n_heads, bool(self.causal), float(self.dropout), p_mlp, *params)


def S(**kwargs): return TransformerConfig(n_layers=6, n_heads=8, n_embd=512, **kwargs)
def B(**kwargs): return TransformerConfig(n_layers=12, n_heads=12, n_embd=768, **kwargs)
def L(**kwargs): return TransformerConfig(n_layers=24, n_heads=16, n_embd=1024, **kwargs)


transformer_configs = {"S": S, "B": B, "L": L}
