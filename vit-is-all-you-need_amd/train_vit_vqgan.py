"""MI355X-native drop-in for the model half of the reference's `train_vit_vqgan` module
(reference train_vit_vqgan.py:18-91): ViTVQGANConfig, ViTVQGANEncoder, Quantizer, ViTVQGANDecoder,
ViTVQGAN — same signatures and state_dict keys (`encoder.*`, `quant.*`, `decoder.*`).  Zero extra
tokens (an empty `nn.Embedding(0, D)` is part of the checkpoint contract), one latent per patch."""
from dataclasses import dataclass

import torch.nn as nn

from train_titok import HipConv1x1, HipLinear, Quantizer, pixel_shuffle_tokens  # noqa: F401  (Quantizer is the same class)
from train_vit import ViT, ViTConfig


@dataclass
class ViTVQGANConfig:
    image_size: int
    patch_size: int
    codebook_size: int
    latent_dim: int
    transformer: str

    def __post_init__(self):
        self.patch_dim = self.image_size // self.patch_size
        self.n_patches = self.patch_dim ** 2
        self.latent_tokens = self.n_patches
        self.enc_vit_config = ViTConfig(self.image_size, 3, self.patch_size, self.transformer, 0, 0.0)
        self.n_embd = self.enc_vit_config.trans_config.n_embd
        self.dec_vit_config = ViTConfig(self.latent_tokens, self.n_embd, 1, self.transformer, 0, 0.0)
        self.dec_vit_config.n_patches = self.latent_tokens


class ViTVQGANEncoder(nn.Module):
    def __init__(self, config: ViTVQGANConfig):
        super().__init__()
        self.latent_tokens = config.latent_tokens
        self.vit = ViT(config.enc_vit_config)
        self.proj = HipLinear(config.n_embd, config.latent_dim)

    def forward(self, x):
        return self.proj(self.vit(x))


class ViTVQGANDecoder(nn.Module):
    def __init__(self, config: ViTVQGANConfig):
        super().__init__()
        self.config = config
        self.vit = ViT(config.dec_vit_config)
        self.quant_proj = HipLinear(config.latent_dim, config.n_embd)
        self.embd_proj = HipConv1x1(config.n_embd, 3 * config.patch_size ** 2, kernel_size=1)

    def forward(self, z):
        z = self.quant_proj(z).transpose(1, 2).unsqueeze(-1)     # 'b h c -> b c h 1'
        return pixel_shuffle_tokens(self.embd_proj(self.vit(z)), self.config.patch_dim, self.config.patch_size)


class ViTVQGAN(nn.Module):
    def __init__(self, config: ViTVQGANConfig):
        super().__init__()
        self.config = config
        self.encoder = ViTVQGANEncoder(config)
        self.quant = Quantizer(config)
        self.decoder = ViTVQGANDecoder(config)

    def encode(self, z):
        """image -> code ids [b, n_patches]"""
        _, ids, _ = self.quant(self.encoder(z))
        return ids

    def decode(self, z_quant):
        return self.decoder(z_quant)

    def decode_indices(self, indices):
        return self.decoder(self.quant.codebook(indices))

    def forward(self, x):
        """-> (reconstruction [b, 3, H, W], code ids, quantiser loss)"""
        tokens, ids, qloss = self.quant(self.encoder(x))
        return self.decoder(tokens), ids, qloss
