"""MI355X-native drop-in for the model half of the reference's `train_vit` module
(reference train_vit.py:16-53): ViTConfig, ViT, ViTClassifier with identical constructor
signatures, attributes (.config, .patch_proj, .pos_emb, .extra_emb, .transformer, .vit, .head) and
state_dict keys, so `from train_vit import ViTConfig, ViT` (train_titok.py:8, train_vit_vqgan.py:8)
resolves here.  The training loop of the reference (train_vit.py:55-129) is re-stated in
`train_step` / `main` without wandb / torchvision / fp16 GradScaler: bf16 needs no loss scaling.
"""
import argparse
import time
from dataclasses import dataclass

import torch
import torch.nn as nn

from transformer import Transformer, transformer_configs
from utils import get_lr_scheduler
from vitamd.functions import PatchEmbedFn, linear


@dataclass
class ViTConfig:
    image_size: int
    in_channels: int
    patch_size: int
    transformer: str
    extra_tokens: int
    dropout: float

    def __post_init__(self):
        self.n_patches = (self.image_size // self.patch_size) ** 2
        self.patch_dim = 3 * self.patch_size ** 2  # hard-coded 3 as in the reference (train_vit.py:27); unused
        self.trans_config = transformer_configs[self.transformer](block_size=self.n_patches + self.extra_tokens,
                                                                  dropout=self.dropout)


class ViT(nn.Module):
    def __init__(self, args: ViTConfig):
        super().__init__()
        self.config = args
        D = args.trans_config.n_embd
        self.patch_proj = nn.Conv2d(in_channels=args.in_channels, out_channels=D, kernel_size=args.patch_size,
                                    stride=args.patch_size)
        self.pos_emb = nn.Embedding(args.n_patches, D)
        self.extra_emb = nn.Embedding(args.extra_tokens, D)
        self.transformer = Transformer(args.trans_config)

    def forward(self, x):
        # conv patchify + (h w) flatten + pos_emb + prepended extra tokens in one GEMM epilogue
        emb = PatchEmbedFn.apply(x, self.patch_proj.weight, self.patch_proj.bias, self.pos_emb.weight,
                                 self.extra_emb.weight, self.config.patch_size, self.config.n_patches)
        return self.transformer(emb)


class ViTClassifier(nn.Module):
    def __init__(self, vit_config: ViTConfig, num_classes=1000):
        super().__init__()
        self.vit = ViT(vit_config)
        self.head = nn.Linear(vit_config.trans_config.n_embd, num_classes)

    def forward(self, x):
        return linear(self.vit(x)[:, 0], self.head.weight, self.head.bias)


def train_step(model, images, labels, optim, lr_sched=None, loss_fn=None):
    """One iteration of the reference hot loop (train_vit.py:99-107) without the fp16 scaler."""
    loss_fn = loss_fn or nn.functional.cross_entropy
    optim.zero_grad(set_to_none=True)
    loss = loss_fn(model(images), labels)
    loss.backward()
    optim.step()
    if lr_sched is not None:
        lr_sched.step()
    return loss


def main():
    p = argparse.ArgumentParser(description="ViT classifier training on synthetic data (MI355X-native path)")
    p.add_argument("--image_size", type=int, default=224)
    p.add_argument("--patch_size", type=int, default=16)
    p.add_argument("--extra_tokens", type=int, default=1)
    p.add_argument("--transformer", type=str, default="B")
    p.add_argument("--num_classes", type=int, default=1000)
    p.add_argument("--bs", type=int, default=256)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--weight_decay", type=float, default=1e-2)
    p.add_argument("--warmup_steps", type=int, default=10)
    p.add_argument("--train_steps", type=int, default=50)
    args = p.parse_args()
    dev = torch.device("cuda")
    cfg = ViTConfig(args.image_size, 3, args.patch_size, args.transformer, args.extra_tokens, 0.0)
    model = ViTClassifier(cfg, args.num_classes).to(dev)
    optim = torch.optim.AdamW(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    sched = get_lr_scheduler(optim, args.warmup_steps, args.train_steps, args.lr / 10)
    g = torch.Generator(device="cpu").manual_seed(0)
    images = torch.randn(args.bs, 3, args.image_size, args.image_size, generator=g).to(dev)
    labels = torch.randint(0, args.num_classes, (args.bs,), generator=g).to(dev)
    for step in range(args.train_steps):
        t0 = time.time()
        loss = train_step(model, images, labels, optim, sched)
        torch.cuda.synchronize()
        print(f"step {step} loss {loss.item():.4f} {args.bs / (time.time() - t0):.0f} img/s")


if __name__ == "__main__":
    main()
