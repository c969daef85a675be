"""Deterministic, torch-RNG-free parameter / input generator.  TEST INFRASTRUCTURE ONLY.

Golden fixtures store seeds instead of weights: both the fixture generator (which feeds the
values to the real reference in the build container) and the tests on the GPU box (where the
reference does not exist) rebuild bit-identical tensors from numpy's PCG64, whose stream is
stable across platforms and numpy versions.

Distributions follow the reference's default initialisers:
Linear / Conv2d weight and bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)); Embedding ~ N(0,1).
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def uniform(seed, name, shape, bound) -> torch.Tensor:
    a = _rng(seed, name).uniform(-bound, bound, size=shape).astype(np.float32)
    return torch.from_numpy(a)


def normal(seed, name, shape, std=1.0) -> torch.Tensor:
    a = (_rng(seed, name).standard_normal(size=shape) * std).astype(np.float32)
    return torch.from_numpy(a)


def randint(seed, name, shape, high) -> torch.Tensor:
    return torch.from_numpy(_rng(seed, name).integers(0, high, size=shape, dtype=np.int64))


def transformer_state(seed, prefix, n_layers, n_embd, causal_block=None) -> dict:
    """Parameters of reference transformer.Transformer under `prefix` (keys as in
    transformer.py:21,36-41,51)."""
    D = n_embd
    sd = {}
    for i in range(n_layers):
        p = f"{prefix}layers.{i}."
        b = 1.0 / np.sqrt(D)
        sd[p + "multi_attn.qkv.weight"] = uniform(seed, p + "qkv.w", (3 * D, D), b)
        sd[p + "multi_attn.qkv.bias"] = uniform(seed, p + "qkv.b", (3 * D,), b)
        if causal_block is not None:  # persistent buffer, transformer.py:22-25
            m = torch.triu(torch.ones(causal_block, causal_block), diagonal=1)
            sd[p + "multi_attn.mask"] = m.masked_fill(m == 1, float("-inf"))
        sd[p + "mlp.0.weight"] = uniform(seed, p + "fc1.w", (4 * D, D), b)
        sd[p + "mlp.0.bias"] = uniform(seed, p + "fc1.b", (4 * D,), b)
        b2 = 1.0 / np.sqrt(4 * D)
        sd[p + "mlp.2.weight"] = uniform(seed, p + "fc2.w", (D, 4 * D), b2)
        sd[p + "mlp.2.bias"] = uniform(seed, p + "fc2.b", (D,), b2)
    return sd


def vit_state(seed, prefix, in_channels, patch, n_patches, extra, n_layers, n_embd) -> dict:
    """Parameters of reference train_vit.ViT under `prefix` (train_vit.py:34-37)."""
    D = n_embd
    fan_in = in_channels * patch * patch
    b = 1.0 / np.sqrt(fan_in)
    sd = {
        prefix + "patch_proj.weight": uniform(seed, prefix + "pp.w", (D, in_channels, patch, patch), b),
        prefix + "patch_proj.bias": uniform(seed, prefix + "pp.b", (D,), b),
        prefix + "pos_emb.weight": normal(seed, prefix + "pos", (n_patches, D)),
        prefix + "extra_emb.weight": normal(seed, prefix + "extra", (extra, D)),
    }
    sd.update(transformer_state(seed, prefix + "transformer.", n_layers, D))
    return sd


def classifier_state(seed, in_channels, patch, n_patches, extra, n_layers, n_embd, num_classes) -> dict:
    """Parameters of reference train_vit.ViTClassifier (train_vit.py:50-51)."""
    sd = vit_state(seed, "vit.", in_channels, patch, n_patches, extra, n_layers, n_embd)
    b = 1.0 / np.sqrt(n_embd)
    sd["head.weight"] = uniform(seed, "head.w", (num_classes, n_embd), b)
    sd["head.bias"] = uniform(seed, "head.b", (num_classes,), b)
    return sd


def linear_state(seed, prefix, out_f, in_f) -> dict:
    b = 1.0 / np.sqrt(in_f)
    return {prefix + "weight": uniform(seed, prefix + "w", (out_f, in_f), b), prefix + "bias": uniform(seed, prefix + "b", (out_f,), b)}


def tokenizer_state(seed, enc, quant, dec, image_patches, latent_tokens, enc_extra, dec_extra, patch, n_layers, n_embd,
                    codebook_size, latent_dim) -> dict:
    """Parameters of reference train_titok.TiTok (enc='enc.', quant='quant.', dec='dec.') or
    train_vit_vqgan.ViTVQGAN (enc='encoder.', dec='decoder.'): two ViTs, three projections, codebook."""
    D = n_embd
    sd = vit_state(seed, enc + "vit.", 3, patch, image_patches, enc_extra, n_layers, D)
    sd.update(linear_state(seed, enc + "proj.", latent_dim, D))
    sd[quant + "codebook.weight"] = uniform(seed, quant + "codebook", (codebook_size, latent_dim), 1.0 / codebook_size)  # train_titok.py:49
    sd.update(vit_state(seed, dec + "vit.", D, 1, latent_tokens, dec_extra, n_layers, D))
    sd.update(linear_state(seed, dec + "quant_proj.", D, latent_dim))
    ep = linear_state(seed, dec + "embd_proj.", 3 * patch * patch, D)
    ep[dec + "embd_proj.weight"] = ep[dec + "embd_proj.weight"].reshape(3 * patch * patch, D, 1, 1)  # Conv2d(k=1) weight
    sd.update(ep)
    return sd


def module_state(seed, shapes: dict) -> dict:
    """Generic deterministic parameters for a module given {key: shape}: LayerNorm-style vectors named
    '*ln*'/'*norm*' get weight = 1 + 0.2 N(0,1), bias = 0.1 N(0,1) (so the affine part is exercised);
    everything else N(0, 0.05)."""
    sd = {}
    for k, shape in shapes.items():
        is_norm = any(t in k for t in ("ln_", "norm"))
        if is_norm and k.endswith("weight"):
            sd[k] = 1.0 + normal(seed, k, tuple(shape), 0.2)
        elif is_norm:
            sd[k] = normal(seed, k, tuple(shape), 0.1)
        else:
            sd[k] = normal(seed, k, tuple(shape), 0.05)
    return sd
