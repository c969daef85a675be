"""MI355X-native drop-in for the model half of the reference's `train_titok` module
(reference train_titok.py:18-93): TiTokConfig, TiTokEncoder, Quantizer, TiTokDecoder, TiTok with
the same constructor signatures, attributes and state_dict keys (`enc.vit.*`, `enc.proj.*`,
`quant.codebook.weight`, `dec.vit.*`, `dec.quant_proj.*`, `dec.embd_proj.*`).  Both ViTs, the three
projections and the nearest-code search run on libvitamd kernels; the O(tokens x latent_dim)
elementwise glue of the quantiser (L2 normalise, the two MSE terms, the straight-through add) and the
pure data movement (token slicing, pixel shuffle) are torch device ops."""
from dataclasses import dataclass

import torch
import torch.nn as nn

from train_vit import ViT, ViTConfig
from vitamd import ops
from vitamd.functions import linear


@dataclass
class TiTokConfig:
    image_size: int
    patch_size: int
    latent_tokens: int
    codebook_size: int
    latent_dim: int
    transformer: str

    def __post_init__(self):
        self.patch_dim = self.image_size // self.patch_size
        self.n_patches = self.patch_dim ** 2
        self.enc_vit_config = ViTConfig(self.image_size, 3, self.patch_size, self.transformer, self.latent_tokens, 0.0)
        self.n_embd = self.enc_vit_config.trans_config.n_embd
        # decoder: 1x1 "patches" over the latent tokens, one learned mask token per image patch as extra tokens
        self.dec_vit_config = ViTConfig(self.latent_tokens, self.n_embd, 1, self.transformer, self.n_patches, 0.0)
        self.dec_vit_config.n_patches = self.latent_tokens


class HipLinear(nn.Linear):
    """nn.Linear parameters (same state_dict keys), forward/backward on the MFMA GEMMs."""

    def forward(self, x):
        return linear(x, self.weight, self.bias)


class HipConv1x1(nn.Conv2d):
    """nn.Conv2d(kernel_size=1) parameters, applied to TOKEN-major input [B, T, C] -> [B, T, out]."""

    def forward(self, tokens):
        return linear(tokens, self.weight, self.bias)


def pixel_shuffle_tokens(y, grid, p):
    """[B, grid*grid, p*p*c] -> [B, c, grid*p, grid*p]; the two rearranges of reference
    train_titok.py:73-75 composed ('b (h w) (p1 p2 c) -> b c (h p1) (w p2)')."""
    B, _, F = y.shape
    c = F // (p * p)
    return y.view(B, grid, grid, p, p, c).permute(0, 5, 1, 3, 2, 4).reshape(B, c, grid * p, grid * p)


class TiTokEncoder(nn.Module):
    def __init__(self, titok_config: TiTokConfig):
        super().__init__()
        self.latent_tokens = titok_config.latent_tokens
        self.vit = ViT(titok_config.enc_vit_config)
        self.proj = HipLinear(titok_config.n_embd, titok_config.latent_dim)

    def forward(self, x):
        return self.proj(self.vit(x)[:, :self.latent_tokens])   # latent tokens are the PREPENDED extra tokens


class Quantizer(nn.Module):
    def __init__(self, titok_config):
        super().__init__()
        self.codebook = nn.Embedding(titok_config.codebook_size, titok_config.latent_dim)
        self.codebook.weight.data.uniform_(-1.0 / titok_config.codebook_size, 1.0 / titok_config.codebook_size)

    def forward(self, x):
        """cosine-similarity VQ (behaviour of reference train_titok.py:50-59): tokens and codes are compared on the unit sphere,
        the RAW code rows are what comes out; loss = |codes - sg(tokens)|^2 + 0.25 |sg(codes) - tokens|^2 (means); straight-through"""
        unit = torch.nn.functional.normalize(x, dim=-1)
        with torch.no_grad():
            codes_unit = torch.nn.functional.normalize(self.codebook.weight, dim=-1).float().contiguous()
            ids = ops.vq_nearest(unit.reshape(-1, unit.shape[-1]).float().contiguous(), codes_unit).view(unit.shape[:-1])
        picked = self.codebook(ids)
        sq = lambda t: t.pow(2).mean()
        loss = sq(picked - unit.detach()) + 0.25 * sq(picked.detach() - unit)
        return unit + (picked - unit).detach(), ids, loss


class TiTokDecoder(nn.Module):
    def __init__(self, titok_config: TiTokConfig):
        super().__init__()
        self.config = titok_config
        self.vit = ViT(titok_config.dec_vit_config)
        self.quant_proj = HipLinear(titok_config.latent_dim, titok_config.n_embd)
        self.embd_proj = HipConv1x1(titok_config.n_embd, 3 * titok_config.patch_size ** 2, kernel_size=1)

    def forward(self, z):
        z = self.quant_proj(z)                                   # [b, latents, n_embd]
        z = z.transpose(1, 2).unsqueeze(-1)                      # 'b h c -> b c h 1'
        out_embd = self.vit(z)[:, :self.config.n_patches]        # mask tokens come first (extra tokens)
        return pixel_shuffle_tokens(self.embd_proj(out_embd), self.config.patch_dim, self.config.patch_size)


class TiTok(nn.Module):
    def __init__(self, titok_config: TiTokConfig):
        super().__init__()
        self.config = titok_config
        self.enc = TiTokEncoder(titok_config)
        self.quant = Quantizer(titok_config)
        self.dec = TiTokDecoder(titok_config)

    def encode(self, z):
        """image -> code ids [b, latent_tokens]"""
        _, ids, _ = self.quant(self.enc(z))
        return ids

    def decode(self, z_quant):
        return self.dec(z_quant)

    def decode_indices(self, indices):
        return self.dec(self.quant.codebook(indices))

    def forward(self, x):
        """-> (reconstruction [b, 3, H, W], code ids, quantiser loss)"""
        tokens, ids, qloss = self.quant(self.enc(x))
        return self.dec(tokens), ids, qloss
