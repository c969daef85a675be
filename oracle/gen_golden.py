"""Golden-vector generator.  TEST INFRASTRUCTURE ONLY; runs in the BUILD CONTAINER ONLY.

Imports the real reference from /root/reference (read-only, never copied), feeds it the
deterministic tensors of `oracle/weights.py`, runs its PyTorch-CPU fp32 forward/backward and
writes small fixtures under `tests/golden/`.  The reference cannot travel to the GPU box; the
fixtures (data only: inputs by seed, expected outputs) are what pins the oracle there.

  python oracle/gen_golden.py            # regenerates every fixture

`transformer.py` imports as-is (torch + einops).  `train_vit.py` / `train_titok.py` /
`train_vit_vqgan.py` import packages that are absent from this image (torchvision, wandb,
lpips, vector_quantize_pytorch) at module top level but never touch them from the model classes
(all loop code is under `__main__`); empty placeholder modules are registered for those names
so the `import` statements succeed.  Nothing is taken from them.

Each fixture also records the reference's OWN bf16-autocast deviation from its fp32 result
(CPU autocast, same weights and inputs) so tolerances are judged against a measured floor.
"""
from __future__ import annotations

import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import weights as W  # noqa: E402
from vit_oracle import rel_l2  # noqa: E402

SAMPLE_STRIDE = 997  # prime; flattened tensors are sampled every 997th element


def _placeholder(name, **attrs):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m


def import_reference():
    sys.path.insert(0, REF)
    import transformer as ref_transformer  # clean import, no placeholders needed

    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.models = types.ModuleType("torchvision.models")
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tv.transforms)
    sys.modules.setdefault("torchvision.models", tv.models)
    _placeholder("wandb")
    _placeholder("lpips")
    _placeholder("vector_quantize_pytorch", FSQ=None)
    import train_vit as ref_train_vit
    import utils as ref_utils
    global REF_TITOK, REF_VQGAN
    import train_titok as REF_TITOK
    import train_vit_vqgan as REF_VQGAN

    return ref_transformer, ref_train_vit, ref_utils


def sample(t: torch.Tensor) -> torch.Tensor:
    return t.detach().flatten()[::SAMPLE_STRIDE].clone().float()


SMALL = 1 << 16  # tensors up to this many elements are stored in full


def summarize(t: torch.Tensor) -> dict:
    t = t.detach().float()
    d = {"norm": float(t.double().norm()), "sample": sample(t), "shape": list(t.shape)}
    if t.numel() <= SMALL:
        d["full"] = t.clone()
    return d


def save(name, obj):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    torch.save(obj, path)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def run_fwd_bwd(model, x, dy, autocast_bf16=False):
    model.zero_grad(set_to_none=True)
    x = x.detach().clone().requires_grad_(True)
    if autocast_bf16:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y = model(x)
    else:
        y = model(x)
    (y.float() * dy).sum().backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    return y.detach().float(), x.grad.detach().clone(), grads


# ------------------------------------------------------------------------------------------
def gen_transformer_fixtures(RT):
    # G1/G2/G5: tiny stacks, everything stored in full
    for tag, causal in (("tiny", False), ("tiny_causal", True)):
        L, H, D, N, B, seed = 2, 2, 128, 37, 3, 11
        cfg = RT.TransformerConfig(n_layers=L, n_heads=H, n_embd=D, block_size=N, causal=causal)
        model = RT.Transformer(cfg)
        sd = W.transformer_state(seed, "", L, D, causal_block=N if causal else None)
        model.load_state_dict(sd, strict=True)
        x = W.normal(seed, "x", (B, N, D))
        dy = W.normal(seed, "dy", (B, N, D))
        y, dx, grads = run_fwd_bwd(model, x, dy)
        y16, dx16, g16 = run_fwd_bwd(model, x, dy, autocast_bf16=True)
        save(f"transformer_{tag}.pt", {
            "cfg": {"n_layers": L, "n_heads": H, "n_embd": D, "seq": N, "batch": B, "seed": seed, "causal": causal},
            "state_keys": sorted(model.state_dict().keys()),
            "y": y, "dx": dx, "grads": grads,
            "ref_bf16_floor": {"y": rel_l2(y16, y), "dx": rel_l2(dx16, dx),
                               "grads": {k: rel_l2(g16[k], grads[k]) for k in grads}},
        })

    # one ViT-B layer at the real sequence length, summarised
    L, H, D, N, B, seed = 1, 12, 768, 197, 2, 12
    cfg = RT.TransformerConfig(n_layers=L, n_heads=H, n_embd=D, block_size=N)
    model = RT.Transformer(cfg)
    model.load_state_dict(W.transformer_state(seed, "", L, D), strict=True)
    x = W.normal(seed, "x", (B, N, D))
    dy = W.normal(seed, "dy", (B, N, D))
    y, dx, grads = run_fwd_bwd(model, x, dy)
    y16, dx16, g16 = run_fwd_bwd(model, x, dy, autocast_bf16=True)
    save("transformer_layer_b.pt", {
        "cfg": {"n_layers": L, "n_heads": H, "n_embd": D, "seq": N, "batch": B, "seed": seed, "causal": False},
        "y": summarize(y), "dx": summarize(dx), "grads": {k: summarize(v) for k, v in grads.items()},
        "y_row0": y[0, 0].clone(), "dx_row0": dx[0, 0].clone(),
        "ref_bf16_floor": {"y": rel_l2(y16, y), "dx": rel_l2(dx16, dx),
                           "grads": {k: rel_l2(g16[k], grads[k]) for k in grads}},
    })


def gen_classifier_fixture(TV, name, image_size, preset, num_classes, batch, seed, full_grads=False):
    cfg = TV.ViTConfig(image_size, 3, 16, preset, 1, 0.0)
    model = TV.ViTClassifier(cfg, num_classes=num_classes)
    tc = cfg.trans_config
    sd = W.classifier_state(seed, 3, 16, cfg.n_patches, 1, tc.n_layers, tc.n_embd, num_classes)
    model.load_state_dict(sd, strict=True)
    images = W.normal(seed, "images", (batch, 3, image_size, image_size))
    labels = W.randint(seed, "labels", (batch,), num_classes)
    loss_fn = torch.nn.CrossEntropyLoss()

    def run(autocast_bf16):
        model.zero_grad(set_to_none=True)
        if autocast_bf16:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                logits = model(images)
                loss = loss_fn(logits, labels)
        else:
            logits = model(images)
            loss = loss_fn(logits, labels)
        loss.backward()
        return logits.detach().float(), float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    logits, loss, grads = run(False)
    l16, loss16, g16 = run(True)
    obj = {
        "cfg": {"image_size": image_size, "preset": preset, "num_classes": num_classes, "batch": batch,
                "seed": seed, "patch": 16, "extra_tokens": 1, "n_layers": tc.n_layers, "n_heads": tc.n_heads,
                "n_embd": tc.n_embd, "n_patches": cfg.n_patches},
        "n_params": sum(p.numel() for p in model.parameters()),
        "state_keys": sorted(model.state_dict().keys()),
        "state_shapes": {k: list(v.shape) for k, v in model.state_dict().items()},
        "logits": logits, "loss": loss,
        "grads": {k: summarize(v) for k, v in grads.items()},
        "ref_bf16_floor": {"logits": rel_l2(l16, logits), "loss_abs": abs(loss16 - loss),
                           "grads": {k: rel_l2(g16[k], grads[k]) for k in grads}},
    }
    if full_grads:
        obj["full_grads"] = {k: grads[k] for k in ("head.weight", "head.bias", "vit.extra_emb.weight",
                                                     "vit.pos_emb.weight", "vit.patch_proj.bias")}
    save(name, obj)


def gen_lr_fixture(RU):
    base, warm, train, min_lr, steps = 1e-3, 10, 50, 1e-4, 70
    p = torch.nn.Parameter(torch.zeros(1))
    optim = torch.optim.AdamW([p], lr=base)
    sched = RU.get_lr_scheduler(optim, warm, train, min_lr)
    trace = []
    for _ in range(steps):
        trace.append(optim.param_groups[0]["lr"])
        optim.step()
        sched.step()
    save("lr_schedule.pt", {"base_lr": base, "warmup_steps": warm, "train_steps": train, "min_lr": min_lr,
                            "lrs": torch.tensor(trace, dtype=torch.float64)})


def gen_train_steps_fixture(TV, RU):
    """Config 1 plumbing: k full optimiser steps of the reference's loop body
    (train_vit.py:99-107, fp32, no scaler) on a fixed batch; pins the training-step glue."""
    seed, batch, num_classes, steps = 21, 16, 10, 4
    cfg = TV.ViTConfig(32, 3, 16, "S", 1, 0.0)
    model = TV.ViTClassifier(cfg, num_classes=num_classes)
    tc = cfg.trans_config
    model.load_state_dict(W.classifier_state(seed, 3, 16, cfg.n_patches, 1, tc.n_layers, tc.n_embd, num_classes))
    images = W.normal(seed, "images", (batch, 3, 32, 32))
    labels = W.randint(seed, "labels", (batch,), num_classes)
    optim = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2)
    sched = RU.get_lr_scheduler(optim, 2, 100, 1e-4)
    loss_fn = torch.nn.CrossEntropyLoss()
    losses = []
    for _ in range(steps):
        optim.zero_grad()
        loss = loss_fn(model(images), labels)
        loss.backward()
        optim.step()
        sched.step()
        losses.append(float(loss))
    save("train_steps_s32.pt", {"cfg": {"seed": seed, "batch": batch, "num_classes": num_classes, "steps": steps,
                                        "lr": 1e-3, "weight_decay": 1e-2, "warmup": 2, "train_steps": 100, "min_lr": 1e-4},
                                "losses": torch.tensor(losses, dtype=torch.float64),
                                "final_head_bias": model.head.bias.detach().clone()})


def gen_tokenizer_fixture(name, model, sd, images, extra_cfg):
    """TiTok / ViT-VQGAN: reconstruction, indices, quantiser loss, encoder latents and the gradients of
    mse(recon, images) + quantize_loss (the perceptual term needs network weights: out of scope)."""
    model.load_state_dict(sd, strict=True)

    def run(autocast_bf16):
        model.zero_grad(set_to_none=True)
        if autocast_bf16:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                recon, idx, ql = model(images)
        else:
            recon, idx, ql = model(images)
        loss = torch.nn.functional.mse_loss(recon.float(), images) + ql.float()
        loss.backward()
        return recon.detach().float(), idx.detach(), float(ql), float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    enc = model.enc if hasattr(model, "enc") else model.encoder
    latents = enc(images).detach()
    recon, idx, ql, loss, grads = run(False)
    r16, i16, q16, l16, g16 = run(True)
    with torch.no_grad():
        recon_fixed = model.decode_indices(idx).float()
    save(name, {
        "cfg": extra_cfg, "n_params": sum(p.numel() for p in model.parameters()),
        "state_keys": sorted(model.state_dict().keys()),
        "state_shapes": {k: list(v.shape) for k, v in model.state_dict().items()},
        "latents": latents, "indices": idx, "quantize_loss": ql, "loss": loss,
        "recon": summarize(recon), "recon_from_indices": summarize(recon_fixed),
        "grads": {k: summarize(v) for k, v in grads.items()},
        "ref_bf16_floor": {"recon": rel_l2(r16, recon), "index_agreement": float((i16 == idx).float().mean()),
                           "loss_abs": abs(l16 - loss), "grads": {k: rel_l2(g16[k], grads[k]) for k in grads}},
    })


def gen_tokenizer_fixtures():
    # BASELINE configs[3]: TiTok-S 256x256, 32 latent tokens, codebook 2048 x 12
    seed, B = 31, 2
    cfg = REF_TITOK.TiTokConfig(256, 16, 32, 2048, 12, "S")
    sd = W.tokenizer_state(seed, "enc.", "quant.", "dec.", 256, 32, 32, 256, 16, 6, 512, 2048, 12)
    images = W.uniform(seed, "images", (B, 3, 256, 256), 0.5) + 0.5
    gen_tokenizer_fixture("titok_s256.pt", REF_TITOK.TiTok(cfg), sd, images,
                          {"seed": seed, "batch": B, "image_size": 256, "patch": 16, "latent_tokens": 32, "codebook_size": 2048,
                           "latent_dim": 12, "preset": "S", "n_layers": 6, "n_embd": 512})
    # BASELINE configs[4] model: ViT-VQGAN-B 256x256 (0 extra tokens, one latent per patch)
    seed, B = 32, 1
    cfg = REF_VQGAN.ViTVQGANConfig(256, 16, 2048, 12, "B")
    sd = W.tokenizer_state(seed, "encoder.", "quant.", "decoder.", 256, 256, 0, 0, 16, 12, 768, 2048, 12)
    images = W.uniform(seed, "images", (B, 3, 256, 256), 0.5) + 0.5
    gen_tokenizer_fixture("vitvqgan_b256.pt", REF_VQGAN.ViTVQGAN(cfg), sd, images,
                          {"seed": seed, "batch": B, "image_size": 256, "patch": 16, "latent_tokens": 256, "codebook_size": 2048,
                           "latent_dim": 12, "preset": "B", "n_layers": 12, "n_embd": 768})


BLOCK_CASES = {
    # name: (class, ctor kwargs, input kind)
    "rab": ("ResidualAttentionBlock", dict(d_model=128, n_head=2), "lnd"),
    "rab_nomlp": ("ResidualAttentionBlock", dict(d_model=128, n_head=2, mlp_ratio=0), "lnd"),
    "uvit_skip": ("UViTBlock", dict(dim=128, num_heads=2, skip=True), "nld_skip"),
    "uvit_bias": ("UViTBlock", dict(dim=128, num_heads=2, qkv_bias=True), "nld"),
    "attn": ("Attention", dict(dim=128, num_heads=2, qkv_bias=True), "nld"),
    "mlp": ("Mlp", dict(in_features=128, hidden_features=512), "nld"),
}


def gen_blocks_fixture():
    """the reference's blocks.py classes (fp32 CPU) on seeded inputs."""
    import blocks as RB   # the reference's module (torch + einops only)
    out = {}
    for name, (cls, kw, kind) in BLOCK_CASES.items():
        seed = 50 + len(out)
        m = getattr(RB, cls)(**kw)
        shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(W.module_state(seed, shapes), strict=True)
        L, N, D = 37, 3, 128
        x = W.normal(seed, "x", (L, N, D) if kind == "lnd" else (N, L, D)).requires_grad_(True)
        dy = W.normal(seed, "dy", tuple(x.shape))
        args = [x]
        if kind == "nld_skip":
            skip = W.normal(seed, "skip", (N, L, D)).requires_grad_(True)
            args.append(skip)

        def run(bf16):
            m.zero_grad(set_to_none=True)
            for a in args:
                a.grad = None
            if bf16:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    y = m(*args)
            else:
                y = m(*args)
            (y.float() * dy).sum().backward()
            return (y.detach().float(), [a.grad.clone() for a in args], {k: p.grad.detach().clone() for k, p in m.named_parameters()})

        y, dxs, grads = run(False)
        y16, dx16, g16 = run(True)
        out[name] = {"seed": seed, "shapes": shapes, "y": y, "dx": dxs[0], "dskip": dxs[1] if len(dxs) > 1 else None,
                     "grads": grads,
                     "ref_bf16_floor": {"y": rel_l2(y16, y), "dx": rel_l2(dx16[0], dxs[0]), "grads": {k: rel_l2(g16[k], grads[k]) for k in grads}}}
        out[name] = {k: v for k, v in out[name].items() if v is not None}
    save("blocks_tiny.pt", out)


DEPTH_STRIDE = 13    # parameter gradients: vectors in full, matrices as norm + every 13th element (the full set would be 9.6 MB)
DEPTH_CASE = dict(depth=12, dim=128, num_heads=2, qkv_bias=True, tokens=197, batch=2, seed=77)


def gen_blocks_depth_fixture():
    """VERDICT r1 'missing' item 4: blocks.Attention upcasts q, k, v to fp32 before SDPA (reference blocks.py:100) while the build's
    attention kernels work on bf16 operands.  The tiny per-block golden cannot show whether that matters at depth, so this fixture
    runs a STACK of the reference's UViTBlocks (fp32 CPU, and under bf16 autocast - where the reference's SDPA still runs in fp32)
    and records the output, the input gradient and every parameter gradient (norm + strided sample)."""
    import blocks as RB
    c = DEPTH_CASE
    blocks = [RB.UViTBlock(dim=c["dim"], num_heads=c["num_heads"], qkv_bias=c["qkv_bias"]) for _ in range(c["depth"])]
    shapes = {k: list(v.shape) for k, v in blocks[0].state_dict().items()}
    for i, m in enumerate(blocks):
        m.load_state_dict(W.module_state(c["seed"] + i, shapes), strict=True)
    x = W.normal(c["seed"], "x", (c["batch"], c["tokens"], c["dim"])).requires_grad_(True)
    dy = W.normal(c["seed"], "dy", tuple(x.shape))

    def run(bf16):
        x.grad = None
        for m in blocks:
            m.zero_grad(set_to_none=True)
        h = x
        if bf16:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                for m in blocks:
                    h = m(h)
        else:
            for m in blocks:
                h = m(h)
        (h.float() * dy).sum().backward()
        grads = {f"{i}.{k}": p.grad.detach().clone() for i, m in enumerate(blocks) for k, p in m.named_parameters()}
        return h.detach().float(), x.grad.clone(), grads

    y, dx, grads = run(False)
    y16, dx16, g16 = run(True)
    save("blocks_depth.pt", {"case": dict(c), "shapes": shapes, "y": y, "dx": dx, "grads": {k: {"norm": float(v.double().norm()), "sample": (v.flatten() if v.numel() <= 1024 else v.flatten()[::DEPTH_STRIDE]).clone()} for k, v in grads.items()},
                             "ref_bf16_floor": {"y": rel_l2(y16, y), "dx": rel_l2(dx16, dx), "grads": {k: rel_l2(g16[k], grads[k]) for k in grads}}})


TOK_CFG = dict(image_size=32, patch_size=8, transformer="small", latent_tokens=8, latent_dim=12, text_context_length=5, text_embed_dim=64)
VQ_CASES = {
    # name: (ctor kwargs, z shape)
    "plain": (dict(codebook_size=256, token_size=12), (4, 12, 4, 8)),
    "l2norm": (dict(codebook_size=256, token_size=12, use_l2_norm=True), (4, 12, 4, 8)),
    "wide": (dict(), (2, 256, 1, 16)),                                    # the class defaults: 1024 codes x 256
    "cluster": (dict(codebook_size=64, token_size=12, clustering_vq=True), (4, 12, 4, 8)),
}


def _grads_summary(named):
    return {k: summarize(g) for k, g in named.items()}


def gen_block_tokenizer_fixture():
    """blocks.TiTokEncoder / TiTokDecoder / TATiTokDecoder / VectorQuantizer (blocks.py:208-505),
    reference fp32 CPU forward + backward on seeded inputs.  `config` is a plain namespace (the classes only read
    attributes, and `.model.vq_model.get`)."""
    import blocks as RB
    from types import SimpleNamespace as NS
    c = TOK_CFG
    cfg = NS(image_size=c["image_size"], patch_size=c["patch_size"], transformer=c["transformer"], latent_tokens=c["latent_tokens"],
             latent_dim=c["latent_dim"], model=NS(vq_model={"text_context_length": c["text_context_length"], "text_embed_dim": c["text_embed_dim"]}))
    B = 3
    out = {"config": dict(c), "batch": B}
    width = 512

    def case(name, cls, seed, make_inputs):
        m = getattr(RB, cls)(cfg)
        shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(W.module_state(seed, shapes), strict=True)
        inputs = make_inputs(seed)

        def run(bf16):
            m.zero_grad(set_to_none=True)
            args = [a.detach().clone().requires_grad_(True) for a in inputs]
            if bf16:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    y = m(*args)
            else:
                y = m(*args)
            dy = W.normal(seed, "dy", tuple(y.shape))
            (y.float() * dy).sum().backward()
            return y.detach().float(), [a.grad.detach().clone() for a in args], {k: p.grad.detach().clone() for k, p in m.named_parameters()}

        y, dins, grads = run(False)
        y16, d16, g16 = run(True)
        out[name] = {"seed": seed, "shapes": shapes, "y": summarize(y), "dinputs": [summarize(d) for d in dins], "grads": _grads_summary(grads),
                     "ref_bf16_floor": {"y": rel_l2(y16, y), "dinputs": [rel_l2(a, b) for a, b in zip(d16, dins)],
                                        "grads_max": max(rel_l2(g16[k], grads[k]) for k in grads)}}

    lat, ldim = c["latent_tokens"], c["latent_dim"]
    case("encoder", "TiTokEncoder", 71, lambda s: [W.normal(s, "pixels", (B, 3, 32, 32)), W.normal(s, "latent_tokens", (lat, width), 0.05)])
    case("decoder", "TiTokDecoder", 72, lambda s: [W.normal(s, "zq", (B, ldim, 1, lat))])
    case("tatitok_decoder", "TATiTokDecoder", 73, lambda s: [W.normal(s, "zq", (B, ldim, 1, lat)),
                                                              W.normal(s, "text", (B, c["text_context_length"], c["text_embed_dim"]))])

    RB.gather = lambda t: t   # the reference never defines `gather` (blocks.py:457): single-process meaning
    for name, (kw, zshape) in VQ_CASES.items():
        seed = 80 + len([k for k in out if k.startswith("vq_")])
        m = RB.VectorQuantizer(**kw)
        shapes = {k: list(v.shape) for k, v in m.state_dict().items() if k != "embed_prob"}
        sd = W.module_state(seed, shapes)
        m.load_state_dict(sd, strict=False)
        m.train()
        z = (W.normal(seed, "z", zshape) * 0.05).requires_grad_(True)
        code0 = m.embedding.weight.detach().clone()
        zq, res = m(z)
        dy = W.normal(seed, "dy", zshape)
        ((zq * dy).sum() + res["quantizer_loss"]).backward()
        out["vq_" + name] = {"seed": seed, "kwargs": dict(kw), "zshape": list(zshape), "shapes": shapes,
                             "zq": summarize(zq), "indices": res["min_encoding_indices"].clone(),
                             "quantizer_loss": float(res["quantizer_loss"]), "commitment_loss": float(res["commitment_loss"]),
                             "codebook_loss": float(res["codebook_loss"]), "dz": summarize(z.grad), "dcodebook": summarize(m.embedding.weight.grad)}
        if kw.get("clustering_vq"):
            out["vq_" + name]["codebook_after"] = summarize(m.embedding.weight.detach())
            out["vq_" + name]["embed_prob_after"] = m.embed_prob.detach().clone()
            out["vq_" + name]["codebook_moved"] = rel_l2(m.embedding.weight.detach(), code0)
    save("blocks_tokenizers.pt", out)


def gen_autocast_fixture(RT, TV):
    """The reference's OWN bf16-autocast results (CPU autocast of the unmodified reference modules), stored as tensors: an
    independent yardstick for the HIP path that does not pass through oracle/vit_oracle.py's kernel-shaped `lowp` emulation
    (VERDICT r1 item 3).  Same seeds / shapes as transformer_tiny.pt, transformer_layer_b.pt and vit_s32.pt."""
    out = {}
    L, H, D, N, B, seed = 2, 2, 128, 37, 3, 11
    model = RT.Transformer(RT.TransformerConfig(n_layers=L, n_heads=H, n_embd=D, block_size=N))
    model.load_state_dict(W.transformer_state(seed, "", L, D), strict=True)
    x, dy = W.normal(seed, "x", (B, N, D)), W.normal(seed, "dy", (B, N, D))
    y16, dx16, g16 = run_fwd_bwd(model, x, dy, autocast_bf16=True)
    out["transformer_tiny"] = {"y": y16, "dx": dx16.float(), "grads": {k: v.float() for k, v in g16.items()}}
    L, H, D, N, B, seed = 1, 12, 768, 197, 2, 12
    model = RT.Transformer(RT.TransformerConfig(n_layers=L, n_heads=H, n_embd=D, block_size=N))
    model.load_state_dict(W.transformer_state(seed, "", L, D), strict=True)
    x, dy = W.normal(seed, "x", (B, N, D)), W.normal(seed, "dy", (B, N, D))
    y16, dx16, g16 = run_fwd_bwd(model, x, dy, autocast_bf16=True)
    out["transformer_layer_b"] = {"y": summarize(y16), "dx": summarize(dx16), "grads": {k: summarize(v) for k, v in g16.items()}}
    cfg = TV.ViTConfig(32, 3, 16, "S", 1, 0.0)
    tc = cfg.trans_config
    model = TV.ViTClassifier(cfg, num_classes=10)
    seed, batch = 13, 64
    model.load_state_dict(W.classifier_state(seed, 3, 16, cfg.n_patches, 1, tc.n_layers, tc.n_embd, 10), strict=True)
    images, labels = W.normal(seed, "images", (batch, 3, 32, 32)), W.randint(seed, "labels", (batch,), 10)
    model.zero_grad(set_to_none=True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        logits = model(images)
        loss = torch.nn.CrossEntropyLoss()(logits, labels)
    loss.backward()
    out["vit_s32"] = {"logits": logits.detach().float(), "loss": float(loss),
                      "grads": {k: summarize(p.grad) for k, p in model.named_parameters()}}
    save("ref_autocast_bf16.pt", out)


def gen_sdpa_fixture():
    """F.scaled_dot_product_attention exactly as the reference calls it (transformer.py:27-28: default scale, no mask, dropout 0) on the
    ViT-B head geometry (B 2, N 197, H 12, head_dim 64), forward and dq / dk / dv: once in fp32 and once in bf16 (what the reference's
    SDPA sees under autocast: the QKV Linear's bf16 output), both from the SAME bf16-representable inputs.  The distance between the
    two is the reference's own bf16 floor for this op; the HIP attention kernels are held to it instead of to an
    argument about where bf16 roundings sit.  Inputs are regenerated from seeds by the test (oracle/weights.py); outputs are stored as
    every-7th-element samples."""
    import torch.nn.functional as F
    from einops import rearrange
    B, N, H, dh, seed = 2, 197, 12, 64, 41
    qkv = (W.normal(seed, "qkv", (B, N, 3 * H * dh)) * 1.5).bfloat16().float()
    d_o = W.normal(seed, "d_o", (B, N, H * dh)).bfloat16().float()

    def run(dtype):
        t = qkv.to(dtype).detach().clone().requires_grad_(True)
        q, k, v = rearrange(t, "b n (qkv h d) -> qkv b h n d", qkv=3, h=H)          # reference transformer.py:27
        o = rearrange(F.scaled_dot_product_attention(q, k, v), "b h n d -> b n (h d)")   # :28-29
        (o.float() * d_o).sum().backward() if dtype == torch.float32 else o.backward(d_o.to(dtype))
        return o.detach().float(), t.grad.detach().float()

    o32, g32 = run(torch.float32)
    o16, g16 = run(torch.bfloat16)
    D = H * dh
    s7 = lambda t: t.flatten()[::7].clone()                                      # noqa: E731
    parts = {"dq": slice(0, D), "dk": slice(D, 2 * D), "dv": slice(2 * D, 3 * D)}
    out = {"config": {"B": B, "N": N, "H": H, "head_dim": dh, "seed": seed, "input_scale": 1.5, "sample_stride": 7},
           "fp32": {"o": s7(o32), **{k: s7(g32[..., sl].contiguous()) for k, sl in parts.items()}},
           "bf16": {"o": s7(o16), **{k: s7(g16[..., sl].contiguous()) for k, sl in parts.items()}},
           "ref_bf16_floor": {"o": rel_l2(o16, o32), **{k: rel_l2(g16[..., sl], g32[..., sl]) for k, sl in parts.items()}}}
    print("reference SDPA bf16-vs-fp32 floor:", {k: f"{v:.2e}" for k, v in out["ref_bf16_floor"].items()})
    save("sdpa_b197.pt", out)


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def gen_autocast_b224(TV):
    """Adds the headline shape (ViT-B/16, 224 px, 12 layers, batch 2: the inputs of vit_b224.pt) to ref_autocast_bf16.pt: logits, loss and
    every parameter gradient of the unmodified reference under torch.autocast(cpu, bf16).  The other entries of
    the file are kept as they are."""
    path = os.path.join(OUT, "ref_autocast_bf16.pt")
    out = torch.load(path, weights_only=True)
    cfg = TV.ViTConfig(224, 3, 16, "B", 1, 0.0)
    tc = cfg.trans_config
    model = TV.ViTClassifier(cfg, num_classes=1000)
    seed, batch = 14, 2
    model.load_state_dict(W.classifier_state(seed, 3, 16, cfg.n_patches, 1, tc.n_layers, tc.n_embd, 1000), strict=True)
    images, labels = W.normal(seed, "images", (batch, 3, 224, 224)), W.randint(seed, "labels", (batch,), 1000)
    model.zero_grad(set_to_none=True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        logits = model(images)
        loss = torch.nn.CrossEntropyLoss()(logits, labels)
    loss.backward()
    out["vit_b224"] = {"logits": logits.detach().float(), "loss": float(loss),
                       "grads": {k: summarize(p.grad) for k, p in model.named_parameters()}}       # full tensors up to 65 536 elements (vectors), samples + norms beyond
    save("ref_autocast_bf16.pt", out)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if "--sdpa-only" in sys.argv:
        gen_sdpa_fixture()
        return
    RT, TV, RU = import_reference()
    if "--autocast-b224-only" in sys.argv:
        gen_autocast_b224(TV)
        return
    if "--autocast-only" in sys.argv:          # add the round-2 fixture without touching the others
        gen_autocast_fixture(RT, TV)
        return
    if "--blocks-depth-only" in sys.argv:
        gen_blocks_depth_fixture()
        return
    gen_autocast_fixture(RT, TV)
    gen_autocast_b224(TV)
    gen_sdpa_fixture()
    gen_transformer_fixtures(RT)
    gen_classifier_fixture(TV, "vit_s32.pt", 32, "S", 10, 64, seed=13, full_grads=True)   # BASELINE config 1
    gen_classifier_fixture(TV, "vit_b224.pt", 224, "B", 1000, 2, seed=14)                 # BASELINE config 2 shape
    gen_lr_fixture(RU)
    gen_train_steps_fixture(TV, RU)
    gen_tokenizer_fixtures()
    gen_blocks_fixture()
    gen_blocks_depth_fixture()
    gen_block_tokenizer_fixture()


if __name__ == "__main__":
    main()
